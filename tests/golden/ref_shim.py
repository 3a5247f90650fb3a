"""Import shim that lets the *reference* PSALM (read-only at /root/reference) run on CPU
in the authoring container, where detectron2/timm/fvcore/addict/torchvision and the compiled
MultiScaleDeformableAttention op are absent.

TEST INFRASTRUCTURE ONLY.  Used by tests/golden/make_golden.py to generate the committed golden
vectors; never imported by psalm_amd/, bench.py or the -m gpu tests (the GPU box has no
/root/reference).  Nothing here is product code.
"""
import importlib.machinery
import sys
import types
from unittest import mock

import torch
import torch.nn.functional as F

REFERENCE_ROOT = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _magic(name):
    m = mock.MagicMock(name=name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    sys.modules[name] = m
    return m


class _ImageList:
    """detectron2.structures.ImageList.from_tensors restated: pad to size_divisibility, stack."""

    def __init__(self, tensor, image_sizes):
        self.tensor = tensor
        self.image_sizes = image_sizes

    @staticmethod
    def from_tensors(tensors, size_divisibility=0, pad_value=0.0):
        sizes = [(t.shape[-2], t.shape[-1]) for t in tensors]
        mh = max(s[0] for s in sizes)
        mw = max(s[1] for s in sizes)
        if size_divisibility > 1:
            mh = (mh + size_divisibility - 1) // size_divisibility * size_divisibility
            mw = (mw + size_divisibility - 1) // size_divisibility * size_divisibility
        out = tensors[0].new_full((len(tensors), tensors[0].shape[0], mh, mw), pad_value)
        for i, t in enumerate(tensors):
            out[i, :, : t.shape[-2], : t.shape[-1]] = t
        return _ImageList(out, sizes)


class _Instances:
    def __init__(self, image_size, **kw):
        self.__dict__["_image_size"] = image_size
        self.__dict__["_fields"] = {}
        for k, v in kw.items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        self._fields[k] = v

    def __getattr__(self, k):
        if k in ("_fields", "_image_size"):
            raise AttributeError(k)
        try:
            return self._fields[k]
        except KeyError:
            raise AttributeError(k)

    @property
    def image_size(self):
        return self._image_size

    def has(self, k):
        return k in self._fields


class _Boxes:
    def __init__(self, tensor):
        self.tensor = tensor


class _BitMasks:
    def __init__(self, tensor):
        self.tensor = tensor


def _sem_seg_postprocess(result, img_size, output_height, output_width):
    # detectron2.modeling.postprocessing.sem_seg_postprocess: crop then bilinear(align_corners=False)
    result = result[:, : img_size[0], : img_size[1]].expand(1, -1, -1, -1)
    return F.interpolate(result, size=(output_height, output_width), mode="bilinear", align_corners=False)[0]


def _retry_if_cuda_oom(fn):
    return fn


def install():
    """Register the stubs.  Must run after `import transformers` (its availability probes choke on stubs)."""
    import transformers  # noqa: F401
    from transformers import PhiModel, PhiForCausalLM, PhiConfig  # noqa: F401

    if "psalm" in sys.modules:
        return
    import torch.nn as nn

    class DropPath(nn.Module):
        def __init__(self, p=0.0):
            super().__init__()
            self.p = p

        def forward(self, x):
            assert not self.training
            return x

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    def trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0):
        return nn.init.trunc_normal_(t, mean, std, a, b)

    _mod("timm")
    _mod("timm.models")
    _mod("timm.models.layers", DropPath=DropPath, to_2tuple=to_2tuple, trunc_normal_=trunc_normal_)

    def c2_xavier_fill(module):
        nn.init.kaiming_uniform_(module.weight, a=1)
        if module.bias is not None:
            nn.init.constant_(module.bias, 0)

    def c2_msra_fill(module):
        nn.init.kaiming_normal_(module.weight, mode="fan_out", nonlinearity="relu")
        if module.bias is not None:
            nn.init.constant_(module.bias, 0)

    _mod("fvcore")
    _mod("fvcore.nn")
    wi = _mod("fvcore.nn.weight_init", c2_xavier_fill=c2_xavier_fill, c2_msra_fill=c2_msra_fill)
    sys.modules["fvcore.nn"].weight_init = wi
    _magic("fvcore.common")
    _magic("fvcore.common.config")

    class Dict(dict):
        """addict.Dict restated: attribute access, missing key -> empty Dict, nested dicts wrapped."""

        def __init__(self, *a, **kw):
            super().__init__()
            for k, v in dict(*a, **kw).items():
                self[k] = v

        def __setitem__(self, k, v):
            if isinstance(v, dict) and not isinstance(v, Dict):
                v = Dict(v)
            super().__setitem__(k, v)

        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            if k not in self:
                self[k] = Dict()
            return self[k]

        def __setattr__(self, k, v):
            self[k] = v

    _mod("addict", Dict=Dict)

    def _raise(*a, **k):
        raise RuntimeError("MultiScaleDeformableAttention: no CPU kernel (reference falls back to grid_sample)")

    _mod("MultiScaleDeformableAttention", ms_deform_attn_forward=_raise, ms_deform_attn_backward=_raise)

    _magic("detectron2")
    _mod("detectron2.structures", ImageList=_ImageList, Instances=_Instances, Boxes=_Boxes, BitMasks=_BitMasks,
         BoxMode=mock.MagicMock(), PolygonMasks=mock.MagicMock(), polygons_to_bitmask=mock.MagicMock())
    _magic("detectron2.modeling")
    _mod("detectron2.modeling.postprocessing", sem_seg_postprocess=_sem_seg_postprocess)
    _magic("detectron2.utils")
    _mod("detectron2.utils.memory", retry_if_cuda_oom=_retry_if_cuda_oom)
    for n in ["detectron2.config", "detectron2.data", "detectron2.data.detection_utils", "detectron2.data.transforms",
              "detectron2.projects", "detectron2.projects.point_rend", "detectron2.projects.point_rend.point_features",
              "detectron2.utils.comm", "detectron2.layers", "detectron2.evaluation", "detectron2.utils.file_io",
              "detectron2.utils.logger", "detectron2.data.datasets", "detectron2.data.datasets.builtin_meta",
              "pycocotools", "pycocotools.mask", "pycocotools.coco", "cv2", "panopticapi", "panopticapi.utils",
              "shortuuid", "deepspeed", "peft", "torchvision", "torchvision.ops", "torchvision.ops.boxes",
              "torchvision.transforms", "torchvision.models", "torchvision.models._utils", "bitsandbytes"]:
        _magic(n)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def load_reference_mask_cfg(seg_task="panoptic",
                            yaml_path=REFERENCE_ROOT + "/psalm/mask_config/maskformer2_swin_base_384_bs16_50ep.yaml"):
    """YAML + _BASE_ chain merge (what psalm/train/train_datasets.py:36-42 get_mask_config produces)."""
    import os
    import yaml
    install()
    from addict import Dict

    def load(p):
        with open(p) as f:
            d = yaml.unsafe_load(f)
        base = d.pop("_BASE_", None)
        if base is not None:
            b = load(os.path.join(os.path.dirname(p), base))
            merge(b, d)
            return b
        return d

    def merge(a, b):
        for k, v in b.items():
            if isinstance(v, dict) and isinstance(a.get(k), dict):
                merge(a[k], v)
            else:
                a[k] = v

    cfg = Dict(load(yaml_path))
    cfg.MODEL.MASK_FORMER.SEG_TASK = seg_task
    return cfg


def reference_classes():
    install()
    from psalm.model.language_model.llava_phi import PSALM, LlavaConfig  # type: ignore
    return PSALM, LlavaConfig
