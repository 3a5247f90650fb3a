"""`load_pretrained_model` — the reference's model-builder entry point (psalm/model/builder.py:27-72) for the gfx950 path.

Same call signature and the same 4-tuple result `(tokenizer, model, image_processor, context_len)`, so the reference's
evaluation scripts (psalm/eval/panoptic_segmentation.py:96 and siblings) can switch by changing one import.  What it does:

  * mask-decoder config: the reference YAML (with its `_BASE_` chain) through `psalm_amd.config.load_mask_config`, with
    `MODEL.MASK_FORMER.SEG_TASK` taken from `model_args.seg_task` (builder.py:50-51);
  * checkpoint: the Hugging Face directory layout `PSALM.save_pretrained` writes -- `config.json` +
    `model.safetensors` | `model-0000x-of-0000y.safetensors` (+ index) | `pytorch_model.bin` -- read straight into a flat
    state dict (the parameter names are the reference's, llava_phi.py:146-187; no `PSALM.from_pretrained` module tree is built);
  * tokenizer: `AutoTokenizer.from_pretrained(model_path, use_fast=True)` when tokenizer files are present (builder.py:53);
  * `image_processor`: dict with the reference's three keys ('panoptic', 'instance', 'semantic') -> `ImagePreprocessor`
    (resize-shortest-edge / pad-to-square / normalise of coco_panoptic_mapper.py:60-91,134-163 as a plain callable);
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

from .config import PsalmConfig, load_mask_config

MODEL_MAP_NAMES = ("psalm",)            # 'psalm_video' (PSALMForDAVISEval, llava_phi.py:1477-1998) is a "next" row


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


class ImagePreprocessor:
    """coco_panoptic_mapper.py:60-91,134-163 without detectron2: HWC uint8 RGB -> resize shortest edge to `size` (max `size`),
    pad bottom/right to (size, size) with 128, normalise by the reference's pixel mean / std -> dict with `image`
    (3,size,size) fp32, `padding_mask` (size,size) bool, `height`, `width` (the un-padded extent the evaluators resize to)."""

    def __init__(self, size: int, mean, std):
        self.size = size
        self.mean = torch.tensor(mean, dtype=torch.float32).view(3, 1, 1)
        self.std = torch.tensor(std, dtype=torch.float32).view(3, 1, 1)

    def __call__(self, image_hwc_uint8) -> dict:
        img = torch.as_tensor(image_hwc_uint8)
        h, w = int(img.shape[0]), int(img.shape[1])
        s = self.size / max(h, w) if max(h, w) * (self.size / min(h, w)) > self.size else self.size / min(h, w)
        nh, nw = min(self.size, int(h * s + 0.5)), min(self.size, int(w * s + 0.5))
        x = img.permute(2, 0, 1)[None].float()
        x = torch.nn.functional.interpolate(x, size=(nh, nw), mode="bilinear", align_corners=False)[0]
        out = torch.full((3, self.size, self.size), 128.0)
        out[:, :nh, :nw] = x
        pm = torch.ones(self.size, self.size, dtype=torch.bool)
        pm[:nh, :nw] = False
        return {"image": (out - self.mean) / self.std, "padding_mask": pm, "height": h, "width": w}

    preprocess = __call__


def load_pretrained_model(model_path, model_base, model_name, model_args,
                          mask_config="./psalm/mask_config/maskformer2_swin_base_384_bs16_50ep.yaml", load_8bit=False,
                          load_4bit=False, device_map="auto", device="cuda", precision: str = "bf16", use_graphs: bool = True,
                          ops=None):
    """psalm/model/builder.py:27-72.  `precision`, `use_graphs`, `ops` are extensions (defaults = the fast path)."""
    from .model import PSALM
    if load_8bit or load_4bit:
        raise NotImplementedError("bitsandbytes 8/4-bit loading is a CUDA feature of the reference loader; not available on this path")
    name = getattr(model_args, "model_map_name", "psalm")
    if name not in MODEL_MAP_NAMES:
        raise ValueError(f"model_map_name must be one of {MODEL_MAP_NAMES} (got {name!r})")      # builder.py:45-49
    seg_task = getattr(model_args, "seg_task", "instance")                                        # builder.py:51
    mask_cfg = load_mask_config(mask_config if (mask_config and os.path.exists(mask_config)) else None, seg_task=seg_task)
    tokenizer = None
    if any(os.path.exists(os.path.join(model_path, f)) for f in ("tokenizer.json", "vocab.json", "tokenizer_config.json")):
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(model_path, use_fast=True)
    cfg = config_from_hf(model_path, mask_cfg, seg_task)
    sd = read_checkpoint(model_path)
    model = PSALM(cfg, sd, ops=ops, precision=precision, use_graphs=use_graphs)
    m = mask_cfg.MODEL
    size = mask_cfg.INPUT.IMAGE_SIZE
    proc = ImagePreprocessor(size, m.PIXEL_MEAN, m.PIXEL_STD)
    image_processor = {"panoptic": proc, "instance": proc, "semantic": proc}                      # llava_phi.py:66-69
    context_len = 2048
    cj = os.path.join(model_path, "config.json")
    if os.path.exists(cj):
        with open(cj) as f:
            context_len = json.load(f).get("max_sequence_length", 2048)                          # builder.py:67-70
    return tokenizer, model, image_processor, context_len
