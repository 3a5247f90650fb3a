"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain torch-CPU fp32, functional restatement of the reference's `PSALM.eval_seg` inference path
(psalm/model/language_model/llava_phi.py:1317-1472 and everything it calls), written from the
reference's semantics with each function citing the file:line it follows.  Paths are relative to
/root/reference/.  Shorthand:
    LP  psalm/model/language_model/llava_phi.py
    SW  psalm/model/multimodal_encoder/swin_trans.py
    PJ  psalm/model/multimodal_projector/builder.py
    PD  psalm/model/mask_decoder/Mask2Former_Simplify/modeling/pixel_decoder/msdeformattn.py
    OPS psalm/model/mask_decoder/Mask2Former_Simplify/modeling/pixel_decoder/ops/
    TD  psalm/model/mask_decoder/Mask2Former_Simplify/modeling/transformer_decoder/mask2former_transformer_decoder.py
    PE  .../transformer_decoder/position_encoding.py
    CC  psalm/model/visual_prompt_module/context_cluster.py
    PHI transformers/models/phi/modeling_phi.py (third-party; reference pins transformers==4.36.2,
        pyproject.toml:27; restated from the installed 5.15.0 copy, lines cited below)

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this module,
and only as the checker.  The product (`psalm_amd/`) never imports it.

PINNING: this oracle is checked against golden vectors produced by running the *reference code
itself* in the authoring container (tests/golden/make_golden.py -> tests/golden/*.npz; see
tests/test_4_oracle_golden.py).  The reference's only own known-answer test on this path
(OPS/test.py:24-63, MSDA forward vs the grid_sample formula) is reproduced in
tests/test_3_msda.py.

Differences from the reference kept on purpose (each is value-preserving):
  * the Swin tower is evaluated once and its features reused (the reference evaluates it twice on
    the same input, LP:787 and LP:1369);
  * all images of a batch are post-processed (the reference returns after image 0, LP:1472);
  * region pooling takes the sampled point indices from `region_point_sampler` so that the RNG
    stream can be shared with the candidate (CC:31-40 draws from the global torch RNG).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

IMAGE_TOKEN_INDEX, SEG_TOKEN_INDEX, CLS_TOKEN_INDEX, REGION_TOKEN_INDEX, REFER_TOKEN_INDEX = -200, -201, -202, -203, -204


def _lin(sd, name, x, bias=True):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"] if bias and (name + ".bias") in sd else None)


def _ln(sd, name, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


# Control switch for the noise-floor experiments (tools/exp_referring_controls.py): True = the three attention forms below (Swin window, Phi causal,
# mask-decoder multi-head) form their scores, softmax and value products in float64 and round ONCE to fp32 -- an evaluation at least as exact as
# the reference's, in another summation order.  False (always, outside that tool): the reference's fp32 arithmetic, bit for bit.
ATTN_FLOAT64 = False


def _att_up(*ts):
    return tuple(t.double() for t in ts) if ATTN_FLOAT64 else ts


# ------------------------------------------------------------------------------------------ Swin
def _window_partition(x, ws):                                   # SW:37-49
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)


def _window_reverse(w, ws, H, W):                                # SW:52-66
    B = int(w.shape[0] / (H * W / ws / ws))
    x = w.view(B, H // ws, W // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


def swin_shift_mask(Hp, Wp, ws, shift):                          # SW:369-387
    img_mask = torch.zeros((1, Hp, Wp, 1))
    cnt = 0
    for h in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for w in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img_mask[:, h, w, :] = cnt
            cnt += 1
    mw = _window_partition(img_mask, ws).view(-1, ws * ws)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return am.masked_fill(am != 0, -100.0).masked_fill(am == 0, 0.0)


def _swin_block(sd, p, x, H, W, ws, shift, heads, attn_mask):    # SW:194-253 + SW:117-149
    B, L, C = x.shape
    shortcut = x
    x = _ln(sd, p + "norm1", x).view(B, H, W, C)
    pad_r = (ws - W % ws) % ws
    pad_b = (ws - H % ws) % ws
    x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))                     # zeros AFTER the norm (SW:207-214)
    Hp, Wp = x.shape[1], x.shape[2]
    if shift > 0:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    xw = _window_partition(x, ws).view(-1, ws * ws, C)
    B_, N, _ = xw.shape
    hd = C // heads
    qkv = _lin(sd, p + "attn.qkv", xw).reshape(B_, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = _att_up(qkv[0] * hd ** -0.5, qkv[1], qkv[2])
    attn = q @ k.transpose(-2, -1)
    table = sd[p + "attn.relative_position_bias_table"]
    idx = sd[p + "attn.relative_position_index"].view(-1)
    attn = attn + table[idx].view(N, N, -1).permute(2, 0, 1).unsqueeze(0)
    if shift > 0:
        nW = attn_mask.shape[0]
        attn = attn.view(B_ // nW, nW, heads, N, N) + attn_mask.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, heads, N, N)
    attn = attn.softmax(-1)
    xw = (attn @ v).float().transpose(1, 2).reshape(B_, N, C)
    xw = _lin(sd, p + "attn.proj", xw)
    x = _window_reverse(xw.view(-1, ws, ws, C), ws, Hp, Wp)
    if shift > 0:
        x = torch.roll(x, shifts=(shift, shift), dims=(1, 2))
    x = x[:, :H, :W, :].contiguous().view(B, H * W, C)
    x = shortcut + x
    h = _lin(sd, p + "mlp.fc2", F.gelu(_lin(sd, p + "mlp.fc1", _ln(sd, p + "norm2", x))))   # exact erf GELU, SW:16-34
    return x + h


def swin_forward(sd, cfg, images, prefix="model.vision_tower."):
    """SW:608-633.  images (B,3,H,W) -> [res2,res3,res4,res5] NCHW."""
    ps, ws = cfg.swin_patch, cfg.swin_window
    x = images
    if x.shape[3] % ps:
        x = F.pad(x, (0, ps - x.shape[3] % ps))                  # SW:431-434
    if x.shape[2] % ps:
        x = F.pad(x, (0, 0, 0, ps - x.shape[2] % ps))
    x = F.conv2d(x, sd[prefix + "patch_embed.proj.weight"], sd[prefix + "patch_embed.proj.bias"], stride=ps)
    Wh, Ww = x.shape[2], x.shape[3]
    x = _ln(sd, prefix + "patch_embed.norm", x.flatten(2).transpose(1, 2))       # SW:437-441
    outs = []
    for s, (depth, heads) in enumerate(zip(cfg.swin_depths, cfg.swin_heads)):
        H, W = Wh, Ww
        Hp, Wp = int(np.ceil(H / ws)) * ws, int(np.ceil(W / ws)) * ws
        am = swin_shift_mask(Hp, Wp, ws, ws // 2)
        for b in range(depth):
            x = _swin_block(sd, f"{prefix}layers.{s}.blocks.{b}.", x, H, W, ws, 0 if b % 2 == 0 else ws // 2, heads, am)
        C = x.shape[-1]
        xo = _ln(sd, f"{prefix}norm{s}", x)                       # SW:626-631
        outs.append(xo.view(-1, H, W, C).permute(0, 3, 1, 2).contiguous())
        if s < len(cfg.swin_depths) - 1:                          # PatchMerging SW:269-296
            p = f"{prefix}layers.{s}.downsample."
            xx = x.view(-1, H, W, C)
            if H % 2 or W % 2:
                xx = F.pad(xx, (0, 0, 0, W % 2, 0, H % 2))
            xx = torch.cat([xx[:, 0::2, 0::2], xx[:, 1::2, 0::2], xx[:, 0::2, 1::2], xx[:, 1::2, 1::2]], -1)
            xx = xx.view(xx.shape[0], -1, 4 * C)
            x = F.linear(_ln(sd, p + "norm", xx), sd[p + "reduction.weight"])
            Wh, Ww = (H + 1) // 2, (W + 1) // 2
    return outs


# ------------------------------------------------------------------------------------- projector
def _bn(sd, name, x, eps=1e-5):                                   # eval-mode BatchNorm2d
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"],
                        sd[name + ".bias"], False, 0.0, eps)


def projector_forward(sd, res5, prefix="model.mm_projector."):
    """PJ:365-375 + BasicBlock PJ:85-111 -- NOTE conv2 is applied twice (PJ:92-94)."""
    p = prefix + "layer1.0."
    out = F.relu(_bn(sd, p + "bn1", F.conv2d(res5, sd[p + "conv1.weight"], stride=2, padding=1)))
    out = F.conv2d(out, sd[p + "conv2.weight"], padding=1)
    out = F.conv2d(out, sd[p + "conv2.weight"], padding=1)
    out = _bn(sd, p + "bn2", out)
    res = _bn(sd, p + "downsample.1", F.conv2d(res5, sd[p + "downsample.0.weight"], stride=2))
    out = F.relu(out + res)
    out = out.reshape(out.shape[0], out.shape[1], -1).permute(0, 2, 1)
    return _lin(sd, prefix + "fc", out)


# --------------------------------------------------------------------------------- region pooling
def default_region_point_sampler(nonzero: torch.Tensor, n: int) -> torch.Tensor:
    """CC:31-40 rand_sample_repeat -- returns ROW INDICES into `nonzero` (len n), global torch RNG."""
    m = nonzero.shape[0]
    if m < n:
        return torch.cat((torch.arange(m), torch.randint(0, m, (n - m,))))
    if m == n:
        return torch.arange(m)
    return torch.randperm(m)[:n]


def region_pooling(image_tokens, region_masks_list, n_points, sampler: Callable = default_region_point_sampler):
    """CC:333-400.  image_tokens (B, h*w, C); region_masks_list[b] (k,S,S) bool -> list[(k,1,C)]."""
    feats = []
    for tok, masks in zip(image_tokens, region_masks_list):
        if len(masks) == 0:
            feats.append(None)
            continue
        S0, S1 = masks[0].shape
        wh = torch.tensor([S0, S1])[None]
        pts = []
        for m in masks:
            nz = m.nonzero()
            pts.append((nz / wh)[sampler(nz, n_points)])          # normalised (y,x) in [0,1)
        pts = torch.stack(pts)                                     # (k, n, 2)
        h = w = int(math.sqrt(tok.shape[0]))
        fmap = tok.reshape(h, w, -1).permute(2, 0, 1).unsqueeze(0).repeat(pts.shape[0], 1, 1, 1)
        grid = (2.0 * pts.flip(dims=(2,)) - 1.0).unsqueeze(2).float()   # (x,y), CC:371
        samp = F.grid_sample(fmap.float(), grid, align_corners=True).squeeze(3)   # (k,C,n)
        feats.append(samp.mean(-1).unsqueeze(1))                   # AdaptiveAvgPool1d(1), CC:391
    return feats


# --------------------------------------------------------------------------------- token splicing
def splice_inputs(sd, cfg, input_ids, attention_mask, image_tokens, class_name_ids=None, cls_indices=None,
                  token_refer_id=None, region_features=None, has_cls_indices=False, has_refer_indices=False):
    """LP:767-971 (+ LP:581-766, LP:566-580): replace sentinel ids by embeddings, build the index
    tensors, right-pad the batch, extend the attention mask.  Returns a dict."""
    E = sd["model.embed_tokens.weight"]
    seg_q = sd["seg_query"]
    B = input_ids.shape[0]
    embeds, seg_masks, cls_idx_out, refer_idx_out, region_masks = [], [], [], [], []
    for b in range(B):
        ids = input_ids[b].tolist()
        class_embed = None
        if class_name_ids is not None:                             # LP:566-574 embed_class_ids
            ci = cls_indices[b]
            uniq = torch.unique_consecutive(ci)
            uniq = uniq[uniq >= 0]
            class_embed = [E[class_name_ids[b][ci == u]] for u in uniq]
        refer_embed = E[token_refer_id[b]] if token_refer_id is not None else None
        parts, sm, cidx, ridx, rmask = [], [], [], [], []
        cls_i = reg_i = 0
        for t in ids:                                              # LP:614-746
            if t >= 0:
                parts.append(E[t][None]); sm.append(0); cidx.append(0); ridx.append(0); rmask.append(0)
            elif t == IMAGE_TOKEN_INDEX:
                n = image_tokens[b].shape[0]
                parts.append(image_tokens[b]); sm += [0] * n; cidx += [0] * n; ridx += [0] * n; rmask += [0] * n
            elif t == SEG_TOKEN_INDEX:
                n = seg_q.shape[0]
                parts.append(seg_q); sm += [1] * n; cidx += [0] * n; ridx += [0] * n; rmask += [0] * n
            elif t == CLS_TOKEN_INDEX:
                ce = class_embed[cls_i]
                cls_i += 1
                n = ce.shape[0]
                parts.append(ce); sm += [0] * n; cidx += [cls_i] * n; ridx += [0] * n; rmask += [0] * n   # LP:671-674 value i+1
            elif t == REGION_TOKEN_INDEX:
                rf = region_features[b][reg_i]
                reg_i += 1
                n = rf.shape[0]
                parts.append(rf); sm += [0] * n; cidx += [0] * n; ridx += [0] * n; rmask += [1] * n
            elif t == REFER_TOKEN_INDEX:
                n = refer_embed.shape[0]
                parts.append(refer_embed); sm += [0] * n; cidx += [0] * n; ridx += [1] * n; rmask += [0] * n
            else:
                raise ValueError(f"unknown sentinel {t}")
        embeds.append(torch.cat(parts, 0))
        seg_masks.append(torch.tensor(sm)); cls_idx_out.append(torch.tensor(cidx))
        refer_idx_out.append(torch.tensor(ridx)); region_masks.append(torch.tensor(rmask))
    T = input_ids.shape[1]
    Lmax = max(e.shape[0] for e in embeds)
    C = embeds[0].shape[1]
    X = torch.zeros(B, Lmax, C)
    am = torch.zeros(B, Lmax, dtype=torch.bool)

    def padto(lst):
        out = torch.zeros(B, Lmax, dtype=torch.int64)
        for b, t in enumerate(lst):
            out[b, : t.shape[0]] = t
        return out
    for b, e in enumerate(embeds):
        Lb = e.shape[0]
        X[b, :Lb] = e
        # LP:935-947 / LP:964-969: [True]*(Lb-T) ++ original mask ++ [False]*(Lmax-Lb)
        am[b, : Lb - T] = True
        am[b, Lb - T: Lb] = attention_mask[b].bool()
    return {"inputs_embeds": X, "attention_mask": am, "seg_query_mask": padto(seg_masks),
            "class_name_embedding_indices": padto(cls_idx_out) if has_cls_indices else None,
            "refer_embedding_indices": padto(refer_idx_out) if has_refer_indices else None,
            "region_embedding_masks": padto(region_masks) if region_features is not None else None,
            "lengths": [e.shape[0] for e in embeds]}


# ------------------------------------------------------------------------------------------- Phi
def gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def phi_forward(sd, cfg, inputs_embeds, attention_mask, prefix="model."):
    """PHI:343-396 (model), :263-300 (parallel-residual layer), :189-245 (attention, partial RoPE),
    :137-160 (eager attention, fp32 softmax), :53-90 (rotary cos/sin), :248-260 (MLP gelu_new).
    attention_mask (B,L) bool: True = attend.  Causal + key-padding additive mask (finfo.min)."""
    B, L, H = inputs_embeds.shape
    nh, hd, rd = cfg.num_heads, cfg.head_dim, cfg.rotary_dim
    pos = torch.arange(L, dtype=torch.float32)
    inv_freq = 1.0 / (cfg.rope_theta ** (torch.arange(0, rd, 2, dtype=torch.float32) / rd))
    freqs = pos[:, None] * inv_freq[None]
    emb = torch.cat((freqs, freqs), -1)
    cos, sin = emb.cos()[None, None], emb.sin()[None, None]                  # (1,1,L,rd)
    allow = torch.tril(torch.ones(L, L, dtype=torch.bool))[None, None] & attention_mask[:, None, None, :].bool()
    bias = torch.zeros(B, 1, L, L).masked_fill(~allow, torch.finfo(torch.float32).min)

    def rot_half(x):
        return torch.cat((-x[..., rd // 2:], x[..., : rd // 2]), -1)

    h = inputs_embeds
    for i in range(cfg.num_layers):
        p = f"{prefix}layers.{i}."
        x = _ln(sd, p + "input_layernorm", h, cfg.layer_norm_eps)
        q = _lin(sd, p + "self_attn.q_proj", x).view(B, L, nh, hd).transpose(1, 2)
        k = _lin(sd, p + "self_attn.k_proj", x).view(B, L, nh, hd).transpose(1, 2)
        v = _lin(sd, p + "self_attn.v_proj", x).view(B, L, nh, hd).transpose(1, 2)
        qr, kr = q[..., :rd], k[..., :rd]
        q = torch.cat((qr * cos + rot_half(qr) * sin, q[..., rd:]), -1)
        k = torch.cat((kr * cos + rot_half(kr) * sin, k[..., rd:]), -1)
        q, k, v = _att_up(q, k, v)
        w = torch.matmul(q, k.transpose(2, 3)) * hd ** -0.5 + bias
        w = F.softmax(w, dim=-1, dtype=w.dtype)                                    # (fp32: PHI:150 `softmax(..., dtype=torch.float32)`)
        a = torch.matmul(w, v).float().transpose(1, 2).reshape(B, L, H)
        a = _lin(sd, p + "self_attn.dense", a)
        m = _lin(sd, p + "mlp.fc2", gelu_new(_lin(sd, p + "mlp.fc1", x)))
        h = a + m + h
    return _ln(sd, prefix + "final_layernorm", h, cfg.layer_norm_eps)


# ---------------------------------------------------------------------- LLM -> decoder embeddings
def gather_llm_embeddings(sd, hidden, sp):
    """LP:1366-1390: seg queries (LP:1299-1316), class-name mean pool (LP:552-565), SEG mean pool
    (LP:972-978), region rows (LP:302-307); each followed by its Linear(2048->256)."""
    out = {}
    B = hidden.shape[0]
    sq = torch.stack([hidden[b][sp["seg_query_mask"][b] == 1] for b in range(B)])
    out["seg_query"] = _lin(sd, "seg_query_projector", sq)
    if sp["refer_embedding_indices"] is not None:
        se = torch.stack([hidden[b][sp["refer_embedding_indices"][b].bool()].mean(0, keepdim=True) for b in range(B)])
        out["SEG_embedding"] = _lin(sd, "SEG_token_projector", se)
    if sp["class_name_embedding_indices"] is not None:
        ce = []
        for b in range(B):
            idx = sp["class_name_embedding_indices"][b]
            ids = torch.unique(idx)
            ids = ids[ids != 0]
            ce.append(torch.cat([hidden[b][idx == i].mean(0, keepdim=True) for i in ids], 0))
        out["class_name_embedding"] = _lin(sd, "class_name_projector", torch.stack(ce))
    if sp["region_embedding_masks"] is not None:
        out["region_embedding_list"] = [_lin(sd, "region_projector", hidden[b][sp["region_embedding_masks"][b].bool()])
                                        for b in range(B)]
    return out


# --------------------------------------------------------------------------------- pixel decoder
def position_embedding_sine(B, H, W, num_pos_feats, temperature=10000.0):
    """PE:29-52 with normalize=True, scale=2*pi, mask=None -> (B, 2*num_pos_feats, H, W)."""
    y = torch.arange(1, H + 1, dtype=torch.float32)[:, None].expand(H, W)
    x = torch.arange(1, W + 1, dtype=torch.float32)[None, :].expand(H, W)
    eps, scale = 1e-6, 2 * math.pi
    y = y / (y[-1:, :] + eps) * scale
    x = x / (x[:, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_pos_feats)
    px = x[:, :, None] / dim_t
    py = y[:, :, None] / dim_t
    px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
    py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
    return torch.cat((py, px), dim=2).permute(2, 0, 1)[None].expand(B, -1, -1, -1)


def msda_core(value, spatial_shapes, level_start_index, loc, w):
    """The CUDA forward kernel's arithmetic (OPS/src/cuda/ms_deform_im2col_cuda.cuh:242-304 and the
    bilinear helper :38-89), vectorised:  out[b,q,m,:] = sum_{l,p} w * bilinear(value_l, loc),
    h_im = loc_y*H - 0.5, w_im = loc_x*W - 0.5, sample used only if -1 < h_im < H and -1 < w_im < W,
    out-of-range corners contribute 0.
    value (B,S,M,D); loc (B,Lq,M,L,P,2) as (x,y); w (B,Lq,M,L,P) -> (B,Lq,M*D)."""
    B, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    out = torch.zeros(B, Lq, M, D, dtype=value.dtype)
    bi = torch.arange(B)[:, None, None].expand(B, Lq, M)
    mi = torch.arange(M)[None, None, :].expand(B, Lq, M)
    for l in range(L):
        Hl, Wl = int(spatial_shapes[l][0]), int(spatial_shapes[l][1])
        start = int(level_start_index[l])
        for p in range(P):
            h_im = loc[:, :, :, l, p, 1] * Hl - 0.5
            w_im = loc[:, :, :, l, p, 0] * Wl - 0.5
            valid = (h_im > -1) & (w_im > -1) & (h_im < Hl) & (w_im < Wl)
            h_low, w_low = torch.floor(h_im), torch.floor(w_im)
            lh, lw = h_im - h_low, w_im - w_low
            h_low, w_low = h_low.long(), w_low.long()
            acc = torch.zeros(B, Lq, M, D, dtype=value.dtype)
            for dh, dw, cw in ((0, 0, (1 - lh) * (1 - lw)), (0, 1, (1 - lh) * lw), (1, 0, lh * (1 - lw)), (1, 1, lh * lw)):
                hh, ww = h_low + dh, w_low + dw
                ok = valid & (hh >= 0) & (hh <= Hl - 1) & (ww >= 0) & (ww <= Wl - 1)
                idx = start + hh.clamp(0, Hl - 1) * Wl + ww.clamp(0, Wl - 1)
                v = value[bi, idx, mi]                               # (B,Lq,M,D)
                acc = acc + (cw * ok)[..., None] * v
            out = out + w[:, :, :, l, p][..., None] * acc
    return out.reshape(B, Lq, M * D)


def msda_core_grid_sample(value, spatial_shapes, loc, w):
    """The reference's own CPU formula (OPS/functions/ms_deform_attn_func.py:52-78), restated."""
    N_, S_, M_, Dim = value.shape
    _, Lq_, _, L_, P_, _ = loc.shape
    vl = value.split([int(h) * int(w_) for h, w_ in spatial_shapes], dim=1)
    grids = 2 * loc - 1
    samp = []
    for lid, (H_, W_) in enumerate(spatial_shapes):
        v_ = vl[lid].flatten(2).transpose(1, 2).reshape(N_ * M_, Dim, int(H_), int(W_))
        g_ = grids[:, :, :, lid].transpose(1, 2).flatten(0, 1)
        samp.append(F.grid_sample(v_, g_, mode="bilinear", padding_mode="zeros", align_corners=False))
    aw = w.transpose(1, 2).reshape(N_ * M_, 1, Lq_, L_ * P_)
    out = (torch.stack(samp, dim=-2).flatten(-2) * aw).sum(-1).view(N_, M_ * Dim, Lq_)
    return out.transpose(1, 2).contiguous()


def _msda_layer(sd, p, cfg, src, pos, ref, shapes, starts, msda_fn):      # OPS/modules/ms_deform_attn.py:82-124
    B, Lq, D = src.shape
    M, L, P = cfg.md_heads, cfg.md_levels, cfg.md_points
    q = src + pos
    value = _lin(sd, p + "value_proj", src).view(B, Lq, M, D // M)
    off = _lin(sd, p + "sampling_offsets", q).view(B, Lq, M, L, P, 2)
    aw = F.softmax(_lin(sd, p + "attention_weights", q).view(B, Lq, M, L * P), -1).view(B, Lq, M, L, P)
    normalizer = torch.tensor([[w_, h_] for h_, w_ in shapes], dtype=torch.float32)
    loc = ref[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
    out = msda_fn(value, shapes, starts, loc, aw)
    return _lin(sd, p + "output_proj", out)


def pixel_decoder_forward(sd, cfg, feats, prefix="pixel_decoder.", msda_fn=msda_core):
    """PD:268-315 forward_features.  feats = [res2,res3,res4,res5] NCHW."""
    D, G = cfg.md_hidden, cfg.md_gn_groups
    srcs, poss = [], []
    for i, f in enumerate([feats[3], feats[2], feats[1]]):        # res5 -> res3, PD:272-276
        x = F.conv2d(f, sd[f"{prefix}input_proj.{i}.0.weight"], sd[f"{prefix}input_proj.{i}.0.bias"])
        srcs.append(F.group_norm(x, G, sd[f"{prefix}input_proj.{i}.1.weight"], sd[f"{prefix}input_proj.{i}.1.bias"]))
        poss.append(position_embedding_sine(f.shape[0], f.shape[2], f.shape[3], D // 2))
    # PD:136-164
    shapes = [(s.shape[2], s.shape[3]) for s in srcs]
    src = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
    pos = torch.cat([p.flatten(2).transpose(1, 2) + sd[prefix + "transformer.level_embed"][l].view(1, 1, -1)
                     for l, p in enumerate(poss)], 1)
    starts = [0]
    for h, w in shapes[:-1]:
        starts.append(starts[-1] + h * w)
    refs = []                                                     # PD:76-87 with valid_ratios == 1
    for (H_, W_) in shapes:
        ry, rx = torch.meshgrid(torch.linspace(0.5, H_ - 0.5, H_), torch.linspace(0.5, W_ - 0.5, W_), indexing="ij")
        refs.append(torch.stack((rx.reshape(-1) / W_, ry.reshape(-1) / H_), -1))
    ref = torch.cat(refs, 0)[None, :, None, :].expand(src.shape[0], -1, len(shapes), -1)
    for i in range(cfg.md_enc_layers):                            # PD:57-66 post-norm layer
        p = f"{prefix}transformer.encoder.layers.{i}."
        src = _ln(sd, p + "norm1", src + _msda_layer(sd, p + "self_attn.", cfg, src, pos, ref, shapes, starts, msda_fn))
        src = _ln(sd, p + "norm2", src + _lin(sd, p + "linear2", F.relu(_lin(sd, p + "linear1", src))))
    B = src.shape[0]
    out = []
    for l, (h, w) in enumerate(shapes):
        out.append(src[:, starts[l]: starts[l] + h * w].transpose(1, 2).reshape(B, D, h, w))
    # FPN on res2, PD:300-308
    x = feats[0]
    lat = F.relu(F.group_norm(F.conv2d(x, sd[prefix + "adapter_1.0.weight"], sd[prefix + "adapter_1.0.bias"]), G,
                              sd[prefix + "adapter_1.1.weight"], sd[prefix + "adapter_1.1.bias"]))
    y = lat + F.interpolate(out[-1].float(), size=lat.shape[-2:], mode="bilinear", align_corners=False)
    y = F.relu(F.group_norm(F.conv2d(y, sd[prefix + "layer_1.0.weight"], sd[prefix + "layer_1.0.bias"], padding=1), G,
                            sd[prefix + "layer_1.1.weight"], sd[prefix + "layer_1.1.bias"]))
    out.append(y)
    mask_features = F.conv2d(out[-1], sd[prefix + "mask_features.weight"], sd[prefix + "mask_features.bias"])
    return mask_features, out[:3], {"encoder_memory": src}


# -------------------------------------------------------------------------------------- predictor
def _mlp(sd, name, x, n):                                         # TD:187-199
    for j in range(n):
        x = _lin(sd, f"{name}.layers.{j}", x)
        if j < n - 1:
            x = F.relu(x)
    return x


def _mha(sd, name, q, k, v, heads, mask=None):
    """nn.MultiheadAttention(embed, heads) forward in eval, batch-first restatement.
    q (B,Lq,D), k/v (B,Lk,D); mask (B,heads,Lq,Lk) bool, True = not allowed (-inf)."""
    D = q.shape[-1]
    Wi, bi = sd[name + ".in_proj_weight"], sd[name + ".in_proj_bias"]
    qq = F.linear(q, Wi[:D], bi[:D])
    kk = F.linear(k, Wi[D:2 * D], bi[D:2 * D])
    vv = F.linear(v, Wi[2 * D:], bi[2 * D:])
    B, Lq, _ = qq.shape
    hd = D // heads
    qq = qq.view(B, Lq, heads, hd).transpose(1, 2) * hd ** -0.5
    kk = kk.view(B, -1, heads, hd).transpose(1, 2)
    vv = vv.view(B, -1, heads, hd).transpose(1, 2)
    qq, kk, vv = _att_up(qq, kk, vv)
    a = qq @ kk.transpose(-2, -1)
    if mask is not None:
        a = a.masked_fill(mask, float("-inf"))
    a = a.softmax(-1)
    o = (a @ vv).float().transpose(1, 2).reshape(B, Lq, D)
    return _lin(sd, name + ".out_proj", o)


def predictor_forward(sd, cfg, multi_scale, mask_features, seg_query, SEG_embedding=None, class_name_embedding=None,
                      region_embedding_list=None, prefix="predictor."):
    """TD:596-693 forward_woconcat + TD:695-762 forward_prediction_heads (batch-first layout)."""
    D, nh = cfg.md_hidden, cfg.md_heads
    B = seg_query.shape[0]
    src, pos, sizes = [], [], []
    for i in range(cfg.md_levels):
        x = multi_scale[i]
        sizes.append(x.shape[-2:])
        pos.append(position_embedding_sine(B, x.shape[2], x.shape[3], D // 2).flatten(2).transpose(1, 2))
        src.append(x.flatten(2).transpose(1, 2) + sd[prefix + "level_embed.weight"][i][None, None, :])
    query_embed = sd[prefix + "query_embed.weight"][None].expand(B, -1, -1)
    out = seg_query

    def heads(out, size):
        dec = _ln(sd, prefix + "decoder_norm", out)
        r = {"SEG": None, "cls": None, "region": None}
        if SEG_embedding is not None:
            r["SEG"] = torch.einsum("bld,bcd->blc", _mlp(sd, prefix + "SEG_proj", dec, 2), SEG_embedding)
        if class_name_embedding is not None:
            r["cls"] = torch.einsum("bld,bcd->blc", _mlp(sd, prefix + "CLASS_proj", dec, 2), class_name_embedding)
        if region_embedding_list is not None:
            dr = _mlp(sd, prefix + "REGION_proj", dec, 2)
            r["region"] = [torch.einsum("kd,ld->kl", re, d) for d, re in zip(dr, region_embedding_list)]
        me = _mlp(sd, prefix + "mask_embed", dec, 3)
        masks = torch.einsum("bqc,bchw->bqhw", me, mask_features)
        am = F.interpolate(masks, size=tuple(size), mode="bilinear", align_corners=False)
        am = (am.sigmoid().flatten(2) < 0.5)                       # (B,Q,HW) True = masked, TD:754-759
        return r, masks, am

    r, masks, am = heads(out, sizes[0])
    trace = []
    for i in range(cfg.md_dec_layers):
        l = i % cfg.md_levels
        am = am.clone()
        am[am.sum(-1) == am.shape[-1]] = False                     # TD:647
        m4 = am[:, None].expand(-1, nh, -1, -1)
        p = f"{prefix}transformer_cross_attention_layers.{i}."
        out = _ln(sd, p + "norm", out + _mha(sd, p + "multihead_attn", out + query_embed, src[l] + pos[l], src[l], nh, m4))
        p = f"{prefix}transformer_self_attention_layers.{i}."
        out = _ln(sd, p + "norm", out + _mha(sd, p + "self_attn", out + query_embed, out + query_embed, out, nh))
        p = f"{prefix}transformer_ffn_layers.{i}."
        out = _ln(sd, p + "norm", out + _lin(sd, p + "linear2", F.relu(_lin(sd, p + "linear1", out))))
        r, masks, am = heads(out, sizes[(i + 1) % cfg.md_levels])
        trace.append(out)
    return {"pred_SEG_logits": r["SEG"], "pred_class_name_logits": r["cls"], "pred_region_logits": r["region"],
            "pred_masks": masks, "decoder_states": trace}


# -------------------------------------------------------------------------------- post-processing
class Instances:
    """Minimal stand-in for detectron2.structures.Instances (attribute bag), LP:317-323."""

    def __init__(self, image_size, **kw):
        self.image_size = image_size
        for k, v in kw.items():
            setattr(self, k, v)


def sem_seg_postprocess(result, img_size, out_h, out_w):           # detectron2 postprocessing (LP:1427)
    result = result[:, : img_size[0], : img_size[1]].expand(1, -1, -1, -1)
    return F.interpolate(result, size=(out_h, out_w), mode="bilinear", align_corners=False)[0]


def _mask_scores(mask_pred):
    pm = (mask_pred > 0).float()
    return (mask_pred.sigmoid().flatten(1) * pm.flatten(1)).sum(1) / (pm.flatten(1).sum(1) + 1e-6), pm


def semantic_inference(cls, mask_pred):                             # LP:402-406
    return torch.einsum("qc,qhw->chw", F.softmax(cls, -1)[:, :-1], mask_pred.sigmoid())


def instance_inference(cls, mask_pred, is_thing_list, topk, panoptic_on=True):      # LP:407-447
    scores = F.softmax(cls, -1)[:, :-1]
    nc = scores.shape[-1]
    labels = torch.arange(nc).unsqueeze(0).repeat(scores.shape[0], 1).flatten(0, 1)
    s, idx = scores.flatten(0, 1).topk(topk, sorted=False)
    lab = labels[idx]
    qidx = idx // nc
    mp = mask_pred[qidx]
    if panoptic_on:
        keep = torch.tensor([bool(is_thing_list[int(l)]) for l in lab], dtype=torch.bool)
        s, lab, mp, qidx = s[keep], lab[keep], mp[keep], qidx[keep]
    ms, pm = _mask_scores(mp)
    return Instances(mask_pred.shape[-2:], pred_masks=pm, scores=s * ms, pred_classes=lab, query_index=qidx,
                     pred_boxes=torch.zeros(mp.shape[0], 4))


def panoptic_inference(cls, mask_pred, is_thing_list, obj_thr=0.8, overlap_thr=0.8):    # LP:325-386
    scores, labels = F.softmax(cls, -1).max(-1)
    nc = cls.shape[-1] - 1
    mp = mask_pred.sigmoid()
    keep = labels.ne(nc) & (scores > obj_thr)
    cur_scores, cur_classes, cur_masks = scores[keep], labels[keep], mp[keep]
    h, w = mp.shape[-2:]
    pan = torch.zeros((h, w), dtype=torch.int32)
    info = []
    if cur_masks.shape[0] == 0:
        return pan, info
    ids = (cur_scores.view(-1, 1, 1) * cur_masks).argmax(0)
    cur_id = 0
    stuff = {}
    for k in range(cur_classes.shape[0]):
        pc = int(cur_classes[k])
        isthing = is_thing_list[pc]
        area = int((ids == k).sum())
        orig = int((cur_masks[k] >= 0.5).sum())
        m = (ids == k) & (cur_masks[k] >= 0.5)
        if area > 0 and orig > 0 and int(m.sum()) > 0:
            if area / orig < overlap_thr:
                continue
            if not isthing:
                if pc in stuff:
                    pan[m] = stuff[pc]
                    continue
                stuff[pc] = cur_id + 1
            cur_id += 1
            pan[m] = cur_id
            info.append({"id": cur_id, "isthing": bool(isthing), "category_id": pc})
    return pan, info


def referring_inference(SEG_cls, mask_pred, topk):                   # LP:308-324
    s, idx = F.sigmoid(SEG_cls).flatten(0, 1).topk(topk, sorted=False)
    mp = mask_pred[idx]
    ms, pm = _mask_scores(mp)
    return Instances(mask_pred.shape[-2:], pred_masks=pm, scores=s * ms, query_index=idx,
                     pred_boxes=torch.zeros(mp.shape[0], 4))


def region_inference(region_cls, mask_pred):                         # LP:387-400
    scores = F.sigmoid(region_cls)
    ms, pm = _mask_scores(mask_pred)
    return Instances(mask_pred.shape[-2:], pred_masks=pm, scores=(scores * ms[None]).transpose(1, 0),
                     pred_boxes=torch.zeros(mask_pred.shape[0], 4))


# ----------------------------------------------------------------------------------------- driver
@torch.no_grad()
def eval_seg(sd: Dict[str, torch.Tensor], cfg, input_ids, attention_mask, images, seg_info, class_name_ids=None,
             class_name_embedding_indices=None, cls_indices=None, token_refer_id=None, refer_embedding_indices=None,
             labels=None, is_thing_list=None, region_point_sampler: Callable = default_region_point_sampler,
             return_stages: bool = False, postprocess: bool = True, msda_fn=msda_core, vp_images=None):
    """LP:1317-1472 (and, with vp_images, PSALMForDAVISEval.eval_video LP:1845-1998).  Returns list[dict] for ALL images
    (and the stage tensors if asked)."""
    task = cfg.seg_task
    st = {}
    feats = swin_forward(sd, cfg, images)                            # LP:787 / LP:1369 (evaluated once)
    image_tokens = projector_forward(sd, feats[-1])                  # LP:448-451
    st.update(res2=feats[0], res3=feats[1], res4=feats[2], res5=feats[3], image_tokens=image_tokens)
    region_features = None
    if (input_ids == REGION_TOKEN_INDEX).sum() != 0:                 # LP:1346-1349, LP:791-797
        if vp_images is not None:
            # PSALMForDAVISEval.eval_video (LP:1845-1998): identical to eval_seg except that the region features are pooled
            # from the PREVIOUS frame `vp_images` (its own Swin + projector pass) at `vp_region_masks` (LP:1663-1670)
            vp_tokens = projector_forward(sd, swin_forward(sd, cfg, vp_images)[-1])
            st["vp_image_tokens"] = vp_tokens
            region_features = region_pooling(vp_tokens, [s["instances"].vp_region_masks.tensor for s in seg_info],
                                             cfg.region_points, region_point_sampler)
        else:
            region_features = region_pooling(image_tokens, [s["instances"].region_masks.tensor for s in seg_info],
                                             cfg.region_points, region_point_sampler)
        st["region_features"] = region_features
    sp = splice_inputs(sd, cfg, input_ids, attention_mask, image_tokens, class_name_ids, cls_indices, token_refer_id,
                       region_features, class_name_embedding_indices is not None, refer_embedding_indices is not None)
    hidden = phi_forward(sd, cfg, sp["inputs_embeds"], sp["attention_mask"])       # LP:1354-1365
    st.update(inputs_embeds=sp["inputs_embeds"], hidden_states=hidden, lengths=sp["lengths"])
    emb = gather_llm_embeddings(sd, hidden, sp)
    st.update({k: v for k, v in emb.items()})
    mask_features, multi_scale, pdx = pixel_decoder_forward(sd, cfg, feats, msda_fn=msda_fn)   # LP:1370
    st.update(mask_features=mask_features, multi_scale_features=multi_scale, encoder_memory=pdx["encoder_memory"])
    po = predictor_forward(sd, cfg, multi_scale, mask_features, emb["seg_query"], emb.get("SEG_embedding"),
                           emb.get("class_name_embedding"), emb.get("region_embedding_list"))     # LP:1392
    st.update(pred_masks=po["pred_masks"], pred_class_name_logits=po["pred_class_name_logits"],
              pred_SEG_logits=po["pred_SEG_logits"], pred_region_logits=po["pred_region_logits"])
    if not postprocess:
        return ([], st) if return_stages else []
    div = cfg.size_divisibility                                      # ImageList.from_tensors(images, 32), LP:1400
    Hpad = (images.shape[-2] + div - 1) // div * div
    Wpad = (images.shape[-1] + div - 1) // div * div
    mask_up = F.interpolate(po["pred_masks"], size=(Hpad, Wpad), mode="bilinear", align_corners=False)   # LP:1401-1406
    results = []
    for b in range(images.shape[0]):
        info = seg_info[b]
        height = info.get("height", images.shape[-2])
        width = info.get("width", images.shape[-1])
        pm = info["padding_mask"]
        nz = np.where(~(pm.cpu().numpy() if torch.is_tensor(pm) else np.asarray(pm)).astype(bool))      # LP:1418-1423
        oh = int(nz[0].max() - nz[0].min() + 1)
        ow = int(nz[1].max() - nz[1].min() + 1)
        mp = sem_seg_postprocess(mask_up[b], [oh, ow], height, width)            # LP:1426-1429
        r = {}
        if task == "panoptic":
            cls = po["pred_class_name_logits"][b].float()
            r["sem_seg"] = semantic_inference(cls, mp)
            r["instances"] = instance_inference(cls, mp, is_thing_list, cfg.md_queries, True)
            r["panoptic_seg"] = panoptic_inference(cls, mp, is_thing_list, cfg.object_mask_threshold, cfg.overlap_threshold)
        elif task == "semantic":
            # sem_seg_postprocess_before_inference is False for this task (LP:301): the semantic map is computed on the padded
            # full-size masks and cropped / resized afterwards (LP:1437-1440)
            cls = po["pred_class_name_logits"][b].float()
            r["sem_seg"] = sem_seg_postprocess(semantic_inference(cls, mask_up[b]), [oh, ow], height, width)
            mp = mask_up[b]
        elif task == "instance":
            cls = po["pred_class_name_logits"][b].float()
            r["instances"] = instance_inference(cls, mp, None, cfg.md_queries, False)           # no thing filter (LP:428)
        elif task == "referring":
            r["instances"] = referring_inference(po["pred_SEG_logits"][b].float(), mp, cfg.md_queries)
        elif task == "region":
            gt = info["instances"].gt_masks
            r["gt"] = sem_seg_postprocess(gt, [oh, ow], height, width)
            # NOTE the reference uses sample 0's region logits for every image (LP:1462) but also
            # returns after image 0 (LP:1472); per-image logits are the only consistent reading.
            r["instances"] = region_inference(po["pred_region_logits"][b].float(), mp)
        else:
            raise NotImplementedError(task)
        r["mask_pred"] = mp
        results.append(r)
    return (results, st) if return_stages else results
