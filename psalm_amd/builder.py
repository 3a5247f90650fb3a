"""`load_pretrained_model` — the reference's model-builder entry point (psalm/model/builder.py:27-72) for the gfx950 path.

Same call signature and the same 4-tuple result `(tokenizer, model, image_processor, context_len)`, so the reference's
evaluation scripts (psalm/eval/panoptic_segmentation.py:96 and siblings) can switch by changing one import.  What it does:

  * mask-decoder config: the reference YAML (with its `_BASE_` chain) through `psalm_amd.config.load_mask_config`, with
    `MODEL.MASK_FORMER.SEG_TASK` taken from `model_args.seg_task` (builder.py:50-51);
  * checkpoint: the Hugging Face directory layout `PSALM.save_pretrained` writes -- `config.json` +
    `model.safetensors` | `model-0000x-of-0000y.safetensors` (+ index) | `pytorch_model.bin` -- read straight into a flat
    state dict (the parameter names are the reference's, llava_phi.py:146-187; no `PSALM.from_pretrained` module tree is built);
  * tokenizer: `AutoTokenizer.from_pretrained(model_path, use_fast=True)` when tokenizer files are present (builder.py:53);
  * `image_processor`: `model.get_vision_tower().image_processor`, as the reference reads it (builder.py:57-65): dict with the three
    keys ('panoptic', 'instance', 'semantic') -> `ImagePreprocessor`, whose `.preprocess(dataset_dict, ...)` is the mappers' inference
    contract (coco_panoptic_mapper.py:134-163: read `file_name`, ResizeShortestEdge + FixedSizeCrop, normalise, `padding_mask`);
  * `context_len`: `config.max_sequence_length` or 2048 (builder.py:67-70).

`load_8bit` / `load_4bit` / `device_map` are accepted for signature compatibility; bitsandbytes quantisation is a CUDA
feature of the reference's loader and is rejected here rather than ignored.
"""
from __future__ import annotations

import glob
import json
import os
from typing import Dict, Optional

import torch

from . import hf as _hf
from .config import PsalmConfig, load_mask_config

# builder.py:45-49 model_map: 'psalm' -> PSALM, 'psalm_video' -> PSALMForDAVISEval (llava_phi.py:1477-1998).  One class serves both here:
# psalm_amd.model.PSALM carries eval_seg and eval_video.
MODEL_MAP_NAMES = ("psalm", "psalm_video")


def read_checkpoint(model_path: str) -> Dict[str, torch.Tensor]:
    """Flat state dict from a Hugging Face checkpoint directory (safetensors, sharded safetensors, or torch .bin)."""
    st = sorted(glob.glob(os.path.join(model_path, "*.safetensors")))
    if st:
        from safetensors.torch import load_file
        sd: Dict[str, torch.Tensor] = {}
        for f in st:
            sd.update(load_file(f))
        return sd
    bins = sorted(glob.glob(os.path.join(model_path, "pytorch_model*.bin")))
    if bins:
        sd = {}
        for f in bins:
            sd.update(torch.load(f, map_location="cpu", weights_only=True))
        return sd
    raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {model_path}")


def config_from_hf(model_path: str, mask_cfg, seg_task: str) -> PsalmConfig:
    """PsalmConfig from the checkpoint's config.json (LlavaConfig(PhiConfig) fields, llava_phi.py:34-35) + the mask YAML."""
    kw = {}
    cj = os.path.join(model_path, "config.json")
    if os.path.exists(cj):
        with open(cj) as f:
            hf = json.load(f)
        for src, dst in (("vocab_size", "vocab_size"), ("hidden_size", "hidden_size"), ("intermediate_size", "intermediate_size"),
                         ("num_hidden_layers", "num_layers"), ("num_attention_heads", "num_heads"),
                         ("partial_rotary_factor", "partial_rotary_factor"), ("rope_theta", "rope_theta"),
                         ("layer_norm_eps", "layer_norm_eps"), ("max_position_embeddings", "max_position_embeddings"),
                         ("projector_outdim", "proj_planes")):
            if src in hf and hf[src] is not None:
                kw[dst] = hf[src]
        if hf.get("swin_type", "base") != "base":
            raise NotImplementedError("only the Swin-B tower of the released PSALM is built (SURVEY.md §2 row 2)")
    return PsalmConfig.from_mask_config(mask_cfg, seg_task=seg_task, **kw)


def hf_config(model_path: str):
    """`model.config`: LlavaConfig (llava_phi.py:34) from the checkpoint's config.json (extra PSALM keys kept as attributes)."""
    if os.path.exists(os.path.join(model_path, "config.json")):
        return _hf.LlavaConfig.from_pretrained(model_path)
    return _hf.LlavaConfig()


def resize_shortest_edge_shape(h: int, w: int, size: int, max_size: int):
    """detectron2 ResizeShortestEdge.get_output_shape (the reference's only eval-time geometric transform,
    coco_panoptic_mapper.py:83-87 `T.ResizeShortestEdge(short_edge_length=image_size, max_size=image_size)`)."""
    scale = size * 1.0 / min(h, w)
    newh, neww = (size, scale * w) if h < w else (scale * h, size)
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * scale, neww * scale
    return int(newh + 0.5), int(neww + 0.5)


class ImagePreprocessor:
    """The inference contract of the reference's dataset mappers (COCOPanopticNewBaselineDatasetMapper.preprocess,
    coco_panoptic_mapper.py:134-163; the instance / semantic mappers share it) without detectron2:

        HWC uint8 RGB  ->  ResizeShortestEdge(size, max_size=size): PIL bilinear on uint8 (detectron2 ResizeTransform.apply_image,
                           i.e. Pillow's antialiasing two-pass resampler with 8-bit intermediate rounding)
                       ->  FixedSizeCrop((size, size)): pad bottom / right with 128, `padding_mask` True on the pad
                       ->  (x - PIXEL_MEAN) / PIXEL_STD, CHW fp32.

    `preprocess(dataset_dict, ...)` takes the detectron2 dataset dict the reference's datasets pass (reads `file_name`, or takes an
    in-memory `image` HWC uint8) and returns a copy with `image`, `padding_mask`, `transforms` (the resize / pad geometry) and
    `height` / `width` (original size, unless the dict already carries them).  Ground-truth annotation transforms (panoptic PNGs, polygons,
    region masks) are dataset-side work the evaluators do from their JSON; dicts that carry `annotations` / `pan_seg_file_name` keep those
    keys untouched.  With `device` set, resize + pad + normalise run on the GPU (psalm_image_preprocess, bit-identical to Pillow)."""

    def __init__(self, size: int, mean, std, device=None, ops=None):
        self.size = size
        self.mean = torch.tensor(mean, dtype=torch.float32).view(3, 1, 1)
        self.std = torch.tensor(std, dtype=torch.float32).view(3, 1, 1)
        self.device = device
        self.ops = ops

    def __call__(self, image_hwc_uint8) -> dict:
        import numpy as np
        img = np.ascontiguousarray(torch.as_tensor(image_hwc_uint8).cpu().numpy()).astype(np.uint8, copy=False)
        h, w = int(img.shape[0]), int(img.shape[1])
        nh, nw = resize_shortest_edge_shape(h, w, self.size, self.size)
        if self.device is not None:                                   # GPU path: one launch pair, result stays on the device
            ops = self.ops
            if ops is None:
                from .hip_ops import get_ops
                ops = self.ops = get_ops()
            image, pm = ops.image_preprocess(torch.from_numpy(img).to(ops.device), nh, nw, self.size, self.mean.view(3), self.std.view(3))
        else:
            from PIL import Image
            res = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR)) if (nh, nw) != (h, w) else img
            out = torch.full((3, self.size, self.size), 128.0)
            out[:, :nh, :nw] = torch.from_numpy(np.ascontiguousarray(res)).permute(2, 0, 1).float()
            pm = torch.ones(self.size, self.size, dtype=torch.bool)
            pm[:nh, :nw] = False
            image = (out - self.mean) / self.std
        return {"image": image, "padding_mask": pm, "height": h, "width": w,
                "transforms": {"resize": (h, w, nh, nw), "pad": (self.size - nh, self.size - nw)}}

    def preprocess(self, dataset_dict, region_mask_type=None, mask_format="polygon", ignore_label=None):
        d = dict(dataset_dict)
        if "image" in d and not torch.is_tensor(d["image"]) or ("image" in d and d["image"].dtype == torch.uint8):
            img = d["image"]
        else:
            import numpy as np
            from PIL import Image
            with Image.open(d["file_name"]) as im:                    # detection_utils.read_image(file_name, format="RGB")
                img = np.asarray(im.convert("RGB"))
        r = self(img)
        if "height" in d and "width" in d and (int(d["height"]), int(d["width"])) != (r["height"], r["width"]):
            raise ValueError(f"Mismatched image shape for {d.get('file_name')}: dict says {(d['height'], d['width'])}, "
                             f"image is {(r['height'], r['width'])}")                     # detection_utils.check_image_size
        d.update(image=r["image"], padding_mask=r["padding_mask"], transforms=r["transforms"])
        d.setdefault("height", r["height"])
        d.setdefault("width", r["width"])
        # Interactive task: the annotations' visual-prompt RLEs -> `instances.region_masks` (coco_instance_mapper.py:233-252: decode ->
        # enhance_with_circles(10 | 5) -> transforms.apply_segmentation) and the kept objects' ground-truth masks -> `instances.gt_masks`
        # (eval_seg's region branch reads both, llava_phi.py:792,1458).  Bitmask (RLE) ground truth only: polygon rasterisation is pycocotools'
        # frPyObjects, the dataset side this processor does not restate.
        annos = [a for a in d.get("annotations", []) if a.get("iscrowd", 0) == 0]
        has_prompts = bool(annos) and "point_visual_prompt_mask" in annos[0]
        # (a dict without annotations, or whose annotations carry no prompt keys, comes back without `instances` whatever `region_mask_type` says:
        #  the reference consults it only inside `if 'point_visual_prompt_mask' in annos[0]`, coco_instance_mapper.py:218,233)
        if has_prompts:
            from .preprocess import apply_segmentation, region_masks_from_annotations, rle_to_mask
            from .synthetic import RegionInstances
            rm, kept = region_masks_from_annotations(annos, r["transforms"], region_mask_type)
            if not kept:
                raise ValueError("preprocess: no annotation of this image has a non-empty visual prompt of the requested kinds")
            gts = []
            for i in kept:
                seg = annos[i].get("segmentation")
                if not isinstance(seg, dict):
                    raise NotImplementedError("preprocess: region prompts need bitmask (RLE dict) ground truth -- mask_format='bitmask', as the "
                                              "reference's interactive / DAVIS evaluation passes (eval_davis.py:317); polygons are not rasterised here")
                gts.append(apply_segmentation(rle_to_mask(seg), r["transforms"]))
            import numpy as np
            d["instances"] = RegionInstances(torch.from_numpy(rm.astype(bool)), torch.from_numpy(np.stack(gts).astype(np.float32)))
            d["region_annotation_indices"] = kept
        return d


def load_pretrained_model(model_path, model_base, model_name, model_args,
                          mask_config="./psalm/mask_config/maskformer2_swin_base_384_bs16_50ep.yaml", load_8bit=False,
                          load_4bit=False, device_map="auto", device="cuda", precision: Optional[str] = None, use_graphs: bool = True,
                          ops=None):
    """psalm/model/builder.py:27-72, same call sequence: mask config (+ SEG_TASK from model_args) -> tokenizer ->
    `model_map[model_map_name].from_pretrained(model_path, mask_decoder_cfg=mask_cfg, **kwargs)` -> `model.get_vision_tower()` ->
    `.image_processor` -> context length from `model.config`.  `precision` (default: PSALM.DEFAULT_PRECISION, the fp32-parity mode),
    `use_graphs`, `ops` are extensions.  `mask_config=None` selects the built-in defaults of the released model; a path that does
    not exist raises (the reference's default is relative to ITS checkout -- a silently substituted default would build the wrong
    geometry)."""
    from .model import PSALM
    if load_8bit or load_4bit:
        raise NotImplementedError("bitsandbytes 8/4-bit loading is a CUDA feature of the reference loader; not available on this path")
    name = getattr(model_args, "model_map_name", "psalm")
    if name not in MODEL_MAP_NAMES:
        raise ValueError(f"model_map_name must be one of {MODEL_MAP_NAMES} (got {name!r})")      # builder.py:45-49
    seg_task = getattr(model_args, "seg_task", "instance")                                        # builder.py:51
    if mask_config is not None and not os.path.exists(mask_config):
        raise FileNotFoundError(f"mask_config {mask_config!r} not found (pass mask_config=None for the built-in defaults of the released model)")
    mask_cfg = load_mask_config(mask_config, seg_task=seg_task)
    tokenizer = None
    if any(os.path.exists(os.path.join(model_path, f)) for f in ("tokenizer.json", "vocab.json", "tokenizer_config.json")):
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(model_path, use_fast=True)
    model = PSALM.from_pretrained(model_path, mask_decoder_cfg=mask_cfg, precision=precision, use_graphs=use_graphs, ops=ops,
                                  torch_dtype=torch.float16, device_map="cpu")               # (the reference's kwargs, builder.py:29-41)
    vision_tower = model.get_vision_tower()
    vision_tower.to(device=device)
    image_processor = vision_tower.image_processor
    context_len = getattr(model.config, "max_sequence_length", 2048)                              # builder.py:67-70
    return tokenizer, model, image_processor, context_len


_hf.register(__import__("psalm_amd.model", fromlist=["PSALM"]).PSALM)       # AutoConfig / AutoModelForCausalLM (llava_phi.py:2001-2002)
