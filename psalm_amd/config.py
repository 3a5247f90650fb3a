"""Architecture + mask-decoder configuration of the PSALM segmentation-inference path.

Values restate the reference's defaults (cited per field).  `PsalmConfig()` is the released PSALM
(Swin-B + Phi-1.5 + Mask2Former head); `PsalmConfig.tiny()` is a shrunken architecture of the same
shape used only to drive the kernels end-to-end at sizes the CPU-side checks finish in seconds.

The mask-decoder YAML surface of the reference (`psalm/mask_config/*.yaml`, loaded by
`psalm/train/train_datasets.py:36-42 get_mask_config`) is provided by `load_mask_config`, which
understands the same `_BASE_` chaining and returns an attribute-style nested dict, so
`load_pretrained_model(..., mask_config=<path to the reference yaml>)` keeps working.
"""
from __future__ import annotations

import dataclasses
import os
from dataclasses import dataclass, field
from typing import Tuple

# psalm/constants.py:7-12 -- sentinel ids of the `input_ids` wire format
IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200
SEG_TOKEN_INDEX = -201
CLS_TOKEN_INDEX = -202
REGION_TOKEN_INDEX = -203
REFER_TOKEN_INDEX = -204


class AttrDict(dict):
    """Attribute-style nested dict (the role addict.Dict / mmcv Config play in the reference,
    psalm/mask_config/config.py:8-26).  Missing keys raise, unlike addict."""

    def __init__(self, *a, **kw):
        super().__init__()
        for k, v in dict(*a, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def default_mask_config(seg_task: str = "panoptic") -> AttrDict:
    """The merged result of maskformer2_swin_base_384_bs16_50ep.yaml -> maskformer2_R50_bs16_50ep.yaml
    -> Base-COCO-InstanceSegmentation.yaml, restricted to the keys the inference path reads
    (psalm/model/language_model/llava_phi.py:174-185,453-531)."""
    return AttrDict({
        "MODEL": {
            "SEM_SEG_HEAD": {
                "NUM_CLASSES": 80, "CONVS_DIM": 256, "MASK_DIM": 256, "COMMON_STRIDE": 4,
                "TRANSFORMER_ENC_LAYERS": 6,
                "IN_FEATURES": ["res2", "res3", "res4", "res5"],
                "DEFORMABLE_TRANSFORMER_ENCODER_IN_FEATURES": ["res3", "res4", "res5"],
            },
            "MASK_FORMER": {
                "HIDDEN_DIM": 256, "NUM_OBJECT_QUERIES": 100, "NHEADS": 8, "DROPOUT": 0.0,
                "DIM_FEEDFORWARD": 2048, "DEC_LAYERS": 10, "PRE_NORM": False,
                "SEG_NORM": False, "SEG_PROJ": True, "FUSE_SCORE": False, "SEG_TASK": seg_task,
                "SIZE_DIVISIBILITY": 32,
                "TEST": {"OVERLAP_THRESHOLD": 0.8, "OBJECT_MASK_THRESHOLD": 0.8},
            },
            "SWIN": {
                "EMBED_DIM": 128, "DEPTHS": [2, 2, 18, 2], "NUM_HEADS": [4, 8, 16, 32], "WINDOW_SIZE": 12,
                "OUT_FEATURES": ["res2", "res3", "res4", "res5"],
            },
            "PIXEL_MEAN": [123.675, 116.280, 103.530],
            "PIXEL_STD": [58.395, 57.120, 57.375],
        },
        "INPUT": {"IMAGE_SIZE": 1024, "MIN_SCALE": 0.1, "MAX_SCALE": 2.0, "FORMAT": "RGB"},
    })


def load_mask_config(path: str | None = None, seg_task: str | None = None) -> AttrDict:
    """YAML loader with `_BASE_` chaining, same merge semantics as the reference's
    get_mask_config (psalm/train/train_datasets.py:36-42: child keys override the base's).
    With path=None returns `default_mask_config()`."""
    if path is None:
        cfg = default_mask_config()
    else:
        import yaml

        def load(p):
            with open(p) as f:
                d = yaml.unsafe_load(f) or {}
            base = d.pop("_BASE_", None)
            if base is None:
                return d
            b = load(os.path.join(os.path.dirname(p), base))
            _merge(b, d)
            return b

        def _merge(a, b):
            for k, v in b.items():
                if isinstance(v, dict) and isinstance(a.get(k), dict):
                    _merge(a[k], v)
                else:
                    a[k] = v

        cfg = default_mask_config()
        _merge(cfg, load(path))
        cfg = AttrDict(cfg)
    if seg_task is not None:
        cfg.MODEL.MASK_FORMER.SEG_TASK = seg_task
    return cfg


@dataclass
class PsalmConfig:
    # --- Swin-B vision tower: psalm/model/multimodal_encoder/swin_trans.py:660-678 (build_swin_b)
    swin_embed_dim: int = 128
    swin_depths: Tuple[int, ...] = (2, 2, 18, 2)
    swin_heads: Tuple[int, ...] = (4, 8, 16, 32)
    swin_window: int = 12
    swin_mlp_ratio: int = 4
    swin_patch: int = 4
    # --- swin_conv projector: psalm/model/multimodal_projector/builder.py:326-375 (ResNetSwin)
    proj_planes: int = 2048
    # --- Phi-1.5 decoder: transformers PhiConfig as used by llava_phi.py:34,52 (susnato/phi-1_5_dev)
    vocab_size: int = 51200
    hidden_size: int = 2048
    intermediate_size: int = 8192
    num_layers: int = 24
    num_heads: int = 32
    partial_rotary_factor: float = 0.5
    rope_theta: float = 10000.0
    layer_norm_eps: float = 1e-5
    max_position_embeddings: int = 2048
    # --- Mask2Former head: psalm/mask_config/maskformer2_R50_bs16_50ep.yaml:4-55, llava_phi.py:453-531
    md_hidden: int = 256
    md_queries: int = 100
    md_heads: int = 8
    md_dim_ff: int = 2048
    md_dec_layers: int = 9          # DEC_LAYERS(10) - 1, llava_phi.py:459
    md_enc_layers: int = 6
    md_enc_ffn: int = 1024          # hard-coded, llava_phi.py:516
    md_levels: int = 3
    md_points: int = 4
    md_gn_groups: int = 32
    md_mask_dim: int = 256
    region_points: int = 256        # llava_phi.py:162 region_pooling(num_sample_point=256)
    size_divisibility: int = 32     # llava_phi.py:267
    seg_task: str = "panoptic"
    object_mask_threshold: float = 0.8   # llava_phi.py:331
    overlap_threshold: float = 0.8       # llava_phi.py:332

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_heads

    @property
    def rotary_dim(self) -> int:
        return int(self.head_dim * self.partial_rotary_factor)

    @property
    def swin_dims(self) -> Tuple[int, ...]:
        return tuple(self.swin_embed_dim * 2 ** i for i in range(len(self.swin_depths)))

    def replace(self, **kw) -> "PsalmConfig":
        return dataclasses.replace(self, **kw)

    @staticmethod
    def tiny(seg_task: str = "panoptic") -> "PsalmConfig":
        """Same graph, ~1e-4 of the FLOPs.  Head dims stay 32 (Swin / mask decoder) and 64 (LLM)
        because the kernels are specialised on them, as the reference architecture fixes them."""
        return PsalmConfig(
            swin_embed_dim=32, swin_depths=(2, 2, 2, 2), swin_heads=(1, 2, 4, 8), swin_window=12,
            proj_planes=64, vocab_size=512, hidden_size=128, intermediate_size=256, num_layers=2, num_heads=2,
            md_hidden=64, md_queries=12, md_heads=2, md_dim_ff=128, md_dec_layers=3, md_enc_layers=2,
            md_enc_ffn=96, md_gn_groups=8, md_mask_dim=64, region_points=16, seg_task=seg_task)

    @staticmethod
    def from_mask_config(mask_cfg, seg_task=None, **kw) -> "PsalmConfig":
        mf = mask_cfg.MODEL.MASK_FORMER
        sh = mask_cfg.MODEL.SEM_SEG_HEAD
        sw = mask_cfg.MODEL.SWIN
        return PsalmConfig(
            swin_embed_dim=sw.EMBED_DIM, swin_depths=tuple(sw.DEPTHS), swin_heads=tuple(sw.NUM_HEADS),
            swin_window=sw.WINDOW_SIZE, md_hidden=mf.HIDDEN_DIM, md_queries=mf.NUM_OBJECT_QUERIES,
            md_heads=mf.NHEADS, md_dim_ff=mf.DIM_FEEDFORWARD, md_dec_layers=mf.DEC_LAYERS - 1,
            md_enc_layers=sh.TRANSFORMER_ENC_LAYERS, md_mask_dim=sh.MASK_DIM,
            seg_task=seg_task or mf.SEG_TASK, **kw)
