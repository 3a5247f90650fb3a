"""Seeded synthetic checkpoint + inputs for the PSALM inference path.

There is no network and no released checkpoint on disk, so parity and throughput are measured on
random-init weights of the reference architecture and synthetic COCO-shaped inputs (BASELINE.json,
SURVEY.md §8(d)).  The state-dict *layout* (names, shapes, dtypes) is exactly the reference's HF
checkpoint layout (`PSALM.state_dict()`, psalm/model/language_model/llava_phi.py:146-187) so that
`tests/golden/make_golden.py` can `load_state_dict(strict=True)` it into the reference model.

The value distributions are chosen (not copied from the reference's init functions) so that every
parameter is non-trivial -- non-zero biases and relative-position tables, non-unit norm scales,
non-trivial BatchNorm running statistics, non-zero sampling-offset weights -- and so that
activations stay O(1) through all stages: otherwise a broken kernel can hide behind zeros.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from .config import (CLS_TOKEN_INDEX, IMAGE_TOKEN_INDEX, REFER_TOKEN_INDEX, REGION_TOKEN_INDEX,
                     SEG_TOKEN_INDEX, PsalmConfig)


# Magnitudes that keep the 100 seg queries / class-name tokens distinguishable after 24 random decoder
# layers (a deep random transformer otherwise collapses all tokens onto one direction and every
# query predicts the same mask and class, which would leave the post-processing branches untested).
SEG_QUERY_STD = 3.0
EMBED_STD = 2.0
LLM_OUT_GAIN = 0.25
PRED_OUT_GAIN = 0.2
MASK_LOGIT_OFFSET = 24.0


class _Gen:
    def __init__(self, seed: int, shapes_only: bool = False):
        self.g = torch.Generator(device="cpu")
        self.g.manual_seed(seed)
        self.shapes_only = shapes_only          # placeholders of the right shapes (means / mid-range values), no random numbers drawn

    def normal(self, *shape, std=1.0, mean=0.0):
        if self.shapes_only:
            return torch.full(shape, float(mean), dtype=torch.float32)
        return torch.randn(*shape, generator=self.g, dtype=torch.float32) * std + mean

    def uniform(self, *shape, lo=0.0, hi=1.0):
        if self.shapes_only:
            return torch.full(shape, 0.5 * (lo + hi), dtype=torch.float32)
        return torch.rand(*shape, generator=self.g, dtype=torch.float32) * (hi - lo) + lo

    def linear(self, sd, name, out_f, in_f, gain=1.0, bias=True, bias_std=0.05):
        sd[name + ".weight"] = self.normal(out_f, in_f, std=gain / math.sqrt(in_f))
        if bias:
            sd[name + ".bias"] = self.normal(out_f, std=bias_std)

    def norm(self, sd, name, c):
        sd[name + ".weight"] = 1.0 + self.normal(c, std=0.1)
        sd[name + ".bias"] = self.normal(c, std=0.05)

    def conv(self, sd, name, out_c, in_c, k, gain=1.0, bias=True):
        sd[name + ".weight"] = self.normal(out_c, in_c, k, k, std=gain / math.sqrt(in_c * k * k))
        if bias:
            sd[name + ".bias"] = self.normal(out_c, std=0.05)

    def bn(self, sd, name, c):
        sd[name + ".weight"] = 1.0 + self.normal(c, std=0.1)
        sd[name + ".bias"] = self.normal(c, std=0.05)
        sd[name + ".running_mean"] = self.normal(c, std=0.1)
        sd[name + ".running_var"] = self.uniform(c, lo=0.5, hi=1.5)
        sd[name + ".num_batches_tracked"] = torch.tensor(100, dtype=torch.int64)


def relative_position_index(ws: int) -> torch.Tensor:
    """(ws*ws, ws*ws) int64 index into the (2ws-1)^2 bias table:
    (dy + ws-1)*(2ws-1) + (dx + ws-1)  -- swin_trans.py:96-107."""
    c = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = c[:, :, None] - c[:, None, :]
    return (rel[0] + ws - 1) * (2 * ws - 1) + (rel[1] + ws - 1)


def make_state_dict(cfg: PsalmConfig, seed: int = 0, include_lm_head: bool = False,
                    extra_vocab: int = 2, shapes_only: bool = False) -> Dict[str, torch.Tensor]:
    """fp32 CPU state dict in the reference checkpoint layout (SURVEY.md §5 'checkpoint / resume').
    shapes_only: constant placeholders of the same names / shapes (what the ranks > 0 of a multi-GPU job build their weight
    arena from before rank 0's weights arrive by broadcast -- no 1.6 G random numbers drawn per rank)."""
    g = _Gen(seed, shapes_only)
    sd: Dict[str, torch.Tensor] = {}
    H = cfg.hidden_size
    V = cfg.vocab_size + extra_vocab            # tokenizer adds [SEG] etc. (train.py); 51202 in practice

    sd["seg_query"] = g.normal(cfg.md_queries, H, std=SEG_QUERY_STD)
    sd["model.embed_tokens.weight"] = g.normal(V, H, std=EMBED_STD)
    for i in range(cfg.num_layers):
        p = f"model.layers.{i}."
        g.linear(sd, p + "self_attn.q_proj", H, H)
        g.linear(sd, p + "self_attn.k_proj", H, H)
        g.linear(sd, p + "self_attn.v_proj", H, H)
        g.linear(sd, p + "self_attn.dense", H, H, gain=LLM_OUT_GAIN)
        g.linear(sd, p + "mlp.fc1", cfg.intermediate_size, H)
        g.linear(sd, p + "mlp.fc2", H, cfg.intermediate_size, gain=LLM_OUT_GAIN)
        g.norm(sd, p + "input_layernorm", H)
    g.norm(sd, "model.final_layernorm", H)

    # ---- Swin (swin_trans.py:446-552)
    vt = "model.vision_tower."
    E = cfg.swin_embed_dim
    g.conv(sd, vt + "patch_embed.proj", E, 3, cfg.swin_patch)
    g.norm(sd, vt + "patch_embed.norm", E)
    ws = cfg.swin_window
    rpi = relative_position_index(ws)
    for s, (depth, heads) in enumerate(zip(cfg.swin_depths, cfg.swin_heads)):
        C = E * 2 ** s
        for b in range(depth):
            p = f"{vt}layers.{s}.blocks.{b}."
            g.norm(sd, p + "norm1", C)
            sd[p + "attn.relative_position_bias_table"] = g.normal((2 * ws - 1) ** 2, heads, std=0.5)
            sd[p + "attn.relative_position_index"] = rpi.clone()
            g.linear(sd, p + "attn.qkv", 3 * C, C, gain=1.5)
            g.linear(sd, p + "attn.proj", C, C, gain=0.5)
            g.norm(sd, p + "norm2", C)
            g.linear(sd, p + "mlp.fc1", cfg.swin_mlp_ratio * C, C)
            g.linear(sd, p + "mlp.fc2", C, cfg.swin_mlp_ratio * C, gain=0.5)
        if s < len(cfg.swin_depths) - 1:
            p = f"{vt}layers.{s}.downsample."
            g.linear(sd, p + "reduction", 2 * C, 4 * C, bias=False)
            g.norm(sd, p + "norm", 4 * C)
    for s in range(len(cfg.swin_depths)):
        g.norm(sd, f"{vt}norm{s}", E * 2 ** s)

    # ---- projector (multimodal_projector/builder.py:326-375)
    pj = "model.mm_projector."
    Cin, P = cfg.swin_dims[-1], cfg.proj_planes
    g.conv(sd, pj + "layer1.0.conv1", P, Cin, 3, bias=False)
    g.bn(sd, pj + "layer1.0.bn1", P)
    g.conv(sd, pj + "layer1.0.conv2", P, P, 3, bias=False)
    g.bn(sd, pj + "layer1.0.bn2", P)
    g.conv(sd, pj + "layer1.0.downsample.0", P, Cin, 1, bias=False)
    g.bn(sd, pj + "layer1.0.downsample.1", P)
    g.linear(sd, pj + "fc", H, P)

    if include_lm_head:       # unused by eval_seg (LP:1365); drawn from its own stream so it never shifts the others
        sd["lm_head.weight"] = _Gen(seed + 7919).normal(V, H, std=0.02)
    D = cfg.md_hidden
    g.linear(sd, "region_projector", D, H)

    # ---- pixel decoder (msdeformattn.py:166-265)
    pd = "pixel_decoder."
    dims = cfg.swin_dims
    for i, cin in enumerate([dims[3], dims[2], dims[1]]):      # res5, res4, res3
        g.conv(sd, f"{pd}input_proj.{i}.0", D, cin, 1)
        g.norm(sd, f"{pd}input_proj.{i}.1", D)
    sd[pd + "transformer.level_embed"] = g.normal(cfg.md_levels, D, std=0.5)
    M, L, Pn = cfg.md_heads, cfg.md_levels, cfg.md_points
    for i in range(cfg.md_enc_layers):
        p = f"{pd}transformer.encoder.layers.{i}."
        # offsets in level pixels: directional bias (as the reference's grid init, ms_deform_attn.py:64-72) + learned part
        sd[p + "self_attn.sampling_offsets.weight"] = g.normal(M * L * Pn * 2, D, std=1.0 / math.sqrt(D))
        thetas = torch.arange(M, dtype=torch.float32) * (2.0 * math.pi / M)
        grid = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(M, 1, 1, 2).repeat(1, L, Pn, 1)
        for k in range(Pn):
            grid[:, :, k, :] *= k + 1
        sd[p + "self_attn.sampling_offsets.bias"] = grid.reshape(-1) + g.normal(M * L * Pn * 2, std=0.1)
        g.linear(sd, p + "self_attn.attention_weights", M * L * Pn, D)
        g.linear(sd, p + "self_attn.value_proj", D, D)
        g.linear(sd, p + "self_attn.output_proj", D, D, gain=0.5)
        g.norm(sd, p + "norm1", D)
        g.linear(sd, p + "linear1", cfg.md_enc_ffn, D)
        g.linear(sd, p + "linear2", D, cfg.md_enc_ffn, gain=0.5)
        g.norm(sd, p + "norm2", D)
    g.conv(sd, pd + "mask_features", cfg.md_mask_dim, D, 1)
    g.conv(sd, pd + "adapter_1.0", D, dims[0], 1)
    g.norm(sd, pd + "adapter_1.1", D)
    g.conv(sd, pd + "layer_1.0", D, D, 3)
    g.norm(sd, pd + "layer_1.1", D)

    # ---- predictor (mask2former_transformer_decoder.py:394-486)
    pr = "predictor."
    for i in range(cfg.md_dec_layers):
        for kind, attn in (("self", "self_attn"), ("cross", "multihead_attn")):
            p = f"{pr}transformer_{kind}_attention_layers.{i}."
            sd[p + attn + ".in_proj_weight"] = g.normal(3 * D, D, std=1.0 / math.sqrt(D))
            sd[p + attn + ".in_proj_bias"] = g.normal(3 * D, std=0.05)
            g.linear(sd, p + attn + ".out_proj", D, D, gain=PRED_OUT_GAIN)
            g.norm(sd, p + "norm", D)
        p = f"{pr}transformer_ffn_layers.{i}."
        g.linear(sd, p + "linear1", cfg.md_dim_ff, D)
        g.linear(sd, p + "linear2", D, cfg.md_dim_ff, gain=PRED_OUT_GAIN)
        g.norm(sd, p + "norm", D)
    g.norm(sd, pr + "decoder_norm", D)
    sd[pr + "query_feat.weight"] = g.normal(cfg.md_queries, D)
    sd[pr + "query_embed.weight"] = g.normal(cfg.md_queries, D)
    sd[pr + "SEG_query_embed.weight"] = g.normal(cfg.md_queries + 1, D)
    sd[pr + "level_embed.weight"] = g.normal(cfg.md_levels, D, std=0.5)
    for name, n_layers, out in (("mask_embed", 3, cfg.md_mask_dim), ("SEG_proj", 2, D), ("CLASS_proj", 2, D),
                                ("REGION_proj", 2, D)):
        for j in range(n_layers):
            g.linear(sd, f"{pr}{name}.layers.{j}", out if j == n_layers - 1 else D, D, gain=1.2)
    # constant mask-feature channel x negative mask-embed bias = a global negative offset on the mask logits, so
    # that masks are sparse and some survive the panoptic overlap test (LP:356-365) instead of all being dropped
    sd[pd + "mask_features.weight"][0] *= 0.05
    sd[pd + "mask_features.bias"][0] = 4.0
    sd[pr + "mask_embed.layers.2.bias"][0] = -MASK_LOGIT_OFFSET / 4.0
    g.linear(sd, "seg_query_projector", D, H)
    g.linear(sd, "SEG_token_projector", D, H)
    g.linear(sd, "class_name_projector", D, H)
    return sd


def scale_residual_branches(sd: Dict[str, torch.Tensor], s: float) -> Dict[str, torch.Tensor]:
    """A copy of `sd` whose residual-branch OUTPUT projections are scaled by `s`: Phi dense / fc2, Swin attn.proj / mlp.fc2, the deformable
    encoder's output_proj / linear2, the mask decoder's out_proj / linear2 (weights and biases).  s < 1 makes every residual block more
    contractive -- the weight set of the experiment "is the reduced-precision modes' IoU a property of the arithmetic or of the unit-gain
    random network" (VERDICT r04 "Next" #7; tools/bench_configs.py --contractive)."""
    out = dict(sd)
    pats = (".self_attn.dense.", ".mlp.fc2.", ".attn.proj.", ".self_attn.output_proj.", ".linear2.", ".out_proj.")
    for k, v in sd.items():
        if any(p in k for p in pats) and (k.endswith(".weight") or k.endswith(".bias")) and torch.is_floating_point(v):
            out[k] = v * s
    return out


# ----------------------------------------------------------------------------------------------
# synthetic inputs (SURVEY.md §8(d) "Config 1..5"): sentinel-id prompts + side index tensors, the
# exact `eval_seg` keyword contract of llava_phi.py:1317-1336.
# ----------------------------------------------------------------------------------------------

def _disc(size, cy, cx, r):
    yy, xx = torch.meshgrid(torch.arange(size), torch.arange(size), indexing="ij")
    return ((yy - cy) ** 2 + (xx - cx) ** 2) <= r * r


class RegionInstances:
    """Attribute bag standing in for detectron2 `Instances` on the *input* side of the region task:
    `seg_info[i]['instances'].region_masks.tensor` (k,S,S) and `.gt_masks` (llava_phi.py:792,1458)."""

    class _T:
        def __init__(self, t):
            self.tensor = t

    def __init__(self, region_masks: torch.Tensor, gt_masks: torch.Tensor, vp_region_masks: Optional[torch.Tensor] = None):
        self.region_masks = RegionInstances._T(region_masks)
        self.gt_masks = gt_masks
        if vp_region_masks is not None:           # DAVIS / eval_video: prompt masks drawn on the PREVIOUS frame (llava_phi.py:1664)
            self.vp_region_masks = RegionInstances._T(vp_region_masks)


# (height, width) of COCO val2017-like originals: the common 4:3 / 3:2 landscape and portrait sizes, a square, a panorama, a small one
COCO_LIKE_SIZES = [(480, 640), (640, 480), (427, 640), (640, 427), (426, 640), (375, 500), (500, 375), (480, 640), (640, 640),
                   (333, 500), (612, 612), (360, 640), (640, 360), (428, 640), (500, 333), (240, 320)]


def resized_box(height: int, width: int, size: int):
    """Extent (rows, cols) of an original (height, width) image after the reference's eval transform T.ResizeShortestEdge(short_edge_length=size,
    max_size=size) (coco_panoptic_mapper.py:81-86; detectron2's rounding: scale the short edge to `size`, re-scale if the long edge then
    exceeds max_size, int(x + 0.5)) -- the un-padded box that T.FixedSizeCrop then pads to (size, size) at the bottom / right."""
    scale = size / min(height, width)
    nh, nw = (size, scale * width) if height < width else (scale * height, size)
    if max(nh, nw) > size:
        sc = size / max(nh, nw)
        nh, nw = nh * sc, nw * sc
    return int(nh + 0.5), int(nw + 0.5)


def make_inputs(cfg: PsalmConfig, task: str = "panoptic", size: int = 1024, batch: int = 1, seed: int = 0,
                num_classes: int = 133, pad: Optional[int] = None, video: bool = False, geometry=None, refer_lens=None) -> dict:
    """Keyword dict for `eval_seg(**inputs)`.

    panoptic : [txt.. <image> txt..] + C x [<cls> ,] + [txt.. <seg> txt]   (train_datasets.py:208-217)
    referring: [.. <image> .. <refer> .. <seg> ..]                          (train_datasets.py:644-695)
    region   : [.. <image> .. k x <region> .. <seg> ..]                     (train_datasets.py:307-354)
    `pad`: number of bottom/right padded pixels flagged in `padding_mask` (None -> 0).
    `geometry`: per image (crop_h, crop_w, height, width) -- the un-padded box inside the (size, size) canvas (`padding_mask` False there,
    pixels outside it zero as T.FixedSizeCrop leaves them after normalisation of its pad value) and the ORIGINAL image size the results
    are resized to (`seg_info[i]["height" / "width"]`, LP:1416-1429); overrides `pad`.
    `refer_lens`: referring task, tokens per sentence for each image (default: a fixed cycle of lengths).
    """
    g = torch.Generator(device="cpu")
    g.manual_seed(1000 + seed)
    V = cfg.vocab_size
    pad = pad or 0
    images = torch.randn(batch, 3, size, size, generator=g, dtype=torch.float32)
    seg_info = []
    for b in range(batch):
        pm = torch.zeros(size, size, dtype=torch.bool)
        if pad:
            pm[size - pad:, :] = True
            pm[:, size - pad:] = True
        if geometry is not None:
            oh, ow, gh, gw = geometry[b % len(geometry)]
            pm[oh:, :] = True
            pm[:, ow:] = True
            images[b, :, oh:, :] = 0
            images[b, :, :, ow:] = 0
            seg_info.append({"padding_mask": pm, "height": int(gh), "width": int(gw)})
            continue
        seg_info.append({"padding_mask": pm, "height": size - pad, "width": size - pad})

    def txt(n):
        return torch.randint(5, V, (n,), generator=g).tolist()

    out = {"images": images, "seg_info": seg_info}
    ids_list: List[List[int]] = []
    if task in ("panoptic", "semantic", "instance"):          # the three class-prompt tasks share one prompt format
        C = num_classes + 1                                   # + "background" (train_datasets.py:67)
        name_ids, cls_idx = [], []
        for c in range(C):
            n = 1 + (c % 3)                                    # 1-3 BPE tokens per name + trailing [SEG]
            name_ids += txt(n) + [V]                           # id V == first added token ([SEG])
            cls_idx += [c] * (n + 1)
        for b in range(batch):
            ids = txt(3) + [IMAGE_TOKEN_INDEX] + txt(2)
            for c in range(C):
                ids += [CLS_TOKEN_INDEX] + txt(1)
            ids += txt(2) + [SEG_TOKEN_INDEX] + txt(1)
            ids_list.append(ids)
        out["class_name_ids"] = torch.tensor([name_ids] * batch, dtype=torch.int64)
        out["cls_indices"] = torch.tensor([cls_idx] * batch, dtype=torch.int64)
        out["is_thing_list"] = [1] * 80 + [0] * (num_classes - 80) if num_classes > 80 else [1] * num_classes
    elif task == "referring":
        lens = [6, 9, 13, 21, 5, 17, 8, 11]
        refer = []
        for b in range(batch):
            ids = txt(4 + b) + [IMAGE_TOKEN_INDEX] + txt(3) + [REFER_TOKEN_INDEX] + txt(2) + [SEG_TOKEN_INDEX] + txt(1)
            ids_list.append(ids)
            refer.append(torch.tensor(txt(refer_lens[b % len(refer_lens)] if refer_lens else lens[b % len(lens)]) + [V], dtype=torch.int64))
        out["token_refer_id"] = refer
    elif task == "region":
        for b in range(batch):
            k = 1 + 2 * (b % 2)
            ids = txt(3) + [IMAGE_TOKEN_INDEX] + txt(2) + [REGION_TOKEN_INDEX] * k + txt(2) + [SEG_TOKEN_INDEX] + txt(1)
            ids_list.append(ids)
            masks = []
            for j in range(k):
                cy = int(torch.randint(size // 8, size - size // 8, (1,), generator=g))
                cx = int(torch.randint(size // 8, size - size // 8, (1,), generator=g))
                masks.append(_disc(size, cy, cx, max(2, size // 100)))
            rm = torch.stack(masks)
            vp = None
            if video:                                          # visual prompts live on the previous frame: different discs
                vp = torch.stack([_disc(size, int(torch.randint(size // 8, size - size // 8, (1,), generator=g)),
                                        int(torch.randint(size // 8, size - size // 8, (1,), generator=g)), max(2, size // 80))
                                  for _ in range(k)])
            seg_info[b]["instances"] = RegionInstances(rm, rm.clone().float(), vp)
    else:
        raise ValueError(task)

    T = max(len(x) for x in ids_list)
    input_ids = torch.full((batch, T), 0, dtype=torch.int64)
    attn = torch.zeros(batch, T, dtype=torch.bool)
    for b, ids in enumerate(ids_list):
        input_ids[b, : len(ids)] = torch.tensor(ids)
        attn[b, : len(ids)] = True
    if video:
        out["vp_images"] = torch.randn(batch, 3, size, size, generator=g, dtype=torch.float32)
    out["input_ids"] = input_ids
    out["attention_mask"] = attn
    out["labels"] = input_ids.clone()
    if task in ("panoptic", "semantic", "instance"):
        out["class_name_embedding_indices"] = (input_ids == CLS_TOKEN_INDEX).to(torch.int64)
    if task == "referring":
        out["refer_embedding_indices"] = (input_ids == REFER_TOKEN_INDEX).to(torch.int64)
    return out
