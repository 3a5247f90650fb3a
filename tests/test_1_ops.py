"""Op-level parity: every HIP kernel vs the same arithmetic written with plain torch fp32/fp64 on CPU
(the formulas are the reference's, taken from oracle/psalm_oracle.py where one exists).
Runs on the host emulation of the kernels here and on the real GPU under `-m gpu`."""
import math

import pytest
import torch
import torch.nn.functional as F

from ops_backend import ops  # noqa: F401
from oracle import psalm_oracle as O

DT = [torch.float32, torch.bfloat16]


def tol(dtype, scale=1.0):
    return (3e-5 if dtype == torch.float32 else 2 ** -7) * scale


def dev(ops, *ts):
    return [t.to(ops.device) if t is not None else None for t in ts]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("rows,C", [(5, 128), (9, 200), (3, 2048)])
def test_layernorm(ops, dtype, rows, C):
    g = torch.Generator().manual_seed(rows + C)
    x = (torch.randn(rows, C, generator=g) * 2 + 0.5).to(dtype)
    ga, be = torch.randn(C, generator=g), torch.randn(C, generator=g)
    want = F.layer_norm(x.float(), (C,), ga, be, 1e-5)
    got = ops.layernorm(*dev(ops, x, ga, be)).cpu().float()
    assert (got - want).abs().max() <= tol(dtype, want.abs().max())
    # strided output into a wider buffer
    big = torch.zeros(rows, C + 16, dtype=dtype, device=ops.device)
    ops.layernorm(*dev(ops, x, ga, be), out=big[:, 8:8 + C])
    assert (big[:, 8:8 + C].cpu().float() - want).abs().max() <= tol(dtype, want.abs().max())
    assert big[:, :8].abs().max() == 0


def test_layernorm_three_outputs(ops):
    """fp32 stream + bf16 copy + bf16 (result + broadcast table) from one LayerNorm pass."""
    g = torch.Generator().manual_seed(2)
    rows, C, r = 10, 64, 5
    x = torch.randn(rows, C, generator=g) * 2 + 0.3
    ga, be, add = torch.randn(C, generator=g), torch.randn(C, generator=g), torch.randn(r, C, generator=g)
    want = F.layer_norm(x, (C,), ga, be, 1e-5)
    d = ops.device
    o2 = torch.zeros(rows, C, dtype=torch.bfloat16, device=d)
    o3 = torch.zeros(rows, C, dtype=torch.bfloat16, device=d)
    o1 = ops.layernorm(x.to(d), ga.to(d), be.to(d), out2=o2, add=add.to(d), out3=o3)
    assert (o1.cpu() - want).abs().max() < 3e-5 * want.abs().max()
    assert (o2.cpu().float() - want).abs().max() <= 2 ** -8 * want.abs().max()
    want3 = want + add.repeat(2, 1)
    assert (o3.cpu().float() - want3).abs().max() <= 2 ** -8 * want3.abs().max()


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,H,W,C,ws,shift", [(1, 12, 12, 32, 12, 0), (2, 17, 14, 32, 12, 6), (1, 24, 30, 64, 12, 6)])
def test_swin_window_gather_and_merge(ops, dtype, B, H, W, C, ws, shift):
    g = torch.Generator().manual_seed(B * 100 + H)
    x = torch.randn(B, H * W, C, generator=g).to(dtype)
    ga, be = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    xn = F.layer_norm(x.float(), (C,), ga, be, 1e-5).view(B, H, W, C)
    pr, pb = (ws - W % ws) % ws, (ws - H % ws) % ws
    xp = F.pad(xn, (0, 0, 0, pr, 0, pb))
    Hp, Wp = xp.shape[1], xp.shape[2]
    if shift:
        xp = torch.roll(xp, (-shift, -shift), (1, 2))
    want = O._window_partition(xp, ws).view(-1, C)
    got = ops.swin_window_gather(*dev(ops, x.view(-1, C), ga, be), B, H, W, ws, shift).cpu().float()
    assert got.shape == want.shape
    assert (got - want).abs().max() <= tol(dtype, want.abs().max())
    # merge: inverse data movement + residual
    win = torch.randn(want.shape, generator=g).to(dtype)
    sc = torch.randn(B * H * W, C, generator=g)
    y = O._window_reverse(win.float().view(-1, ws, ws, C), ws, Hp, Wp)
    if shift:
        y = torch.roll(y, (shift, shift), (1, 2))
    wantm = sc + y[:, :H, :W].reshape(B * H * W, C)
    gotm = ops.swin_window_merge(*dev(ops, win, sc), B, H, W, ws, shift).cpu()
    assert (gotm - wantm).abs().max() <= 1e-5


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,H,W,C", [(1, 8, 8, 32), (2, 7, 5, 64)])
def test_patch_merge_ln(ops, dtype, B, H, W, C):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, H, W, C, generator=g).to(dtype)
    ga, be = 1 + 0.1 * torch.randn(4 * C, generator=g), 0.1 * torch.randn(4 * C, generator=g)
    xx = x.float()
    if H % 2 or W % 2:
        xx = F.pad(xx, (0, 0, 0, W % 2, 0, H % 2))
    cat = torch.cat([xx[:, 0::2, 0::2], xx[:, 1::2, 0::2], xx[:, 0::2, 1::2], xx[:, 1::2, 1::2]], -1)
    want = F.layer_norm(cat.reshape(-1, 4 * C), (4 * C,), ga, be, 1e-5)
    got = ops.patch_merge_ln(*dev(ops, x.view(-1, C), ga, be), B, H, W).cpu().float()
    assert (got - want).abs().max() <= tol(dtype, want.abs().max())


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,HW,C,G,relu", [(1, 100, 64, 8, False), (2, 150, 256, 32, True)])
def test_groupnorm(ops, dtype, B, HW, C, G, relu):
    g = torch.Generator().manual_seed(4)
    x = (torch.randn(B, HW, C, generator=g) * 1.5 + 0.3).to(dtype)
    ga, be = torch.randn(C, generator=g), torch.randn(C, generator=g)
    want = F.group_norm(x.float().transpose(1, 2).reshape(B, C, HW, 1), G, ga, be, 1e-5).reshape(B, C, HW).transpose(1, 2)
    if relu:
        want = F.relu(want)
    got = ops.groupnorm_nhwc(*dev(ops, x.reshape(-1, C), ga, be), B, HW, G, relu=relu).cpu().float().view(B, HW, C)
    assert (got - want).abs().max() <= tol(dtype, want.abs().max())


def test_add_gather_segment(ops):
    g = torch.Generator().manual_seed(5)
    a = torch.randn(12, 40, generator=g)
    b = torch.randn(4, 40, generator=g)
    got = ops.add_bcast(*dev(ops, a, b)).cpu()
    assert torch.equal(got, a + b.repeat(3, 1))
    t0, t1 = torch.randn(7, 40, generator=g), torch.randn(5, 40, generator=g).bfloat16()
    sid = torch.tensor([0, 1, -1, 1, 0, 0], dtype=torch.int32)
    srow = torch.tensor([6, 4, 0, 0, 1, 6], dtype=torch.int32)
    got = ops.gather_rows(dev(ops, t0, t1), *dev(ops, sid, srow), 40).cpu()
    want = torch.stack([t0[6], t1[4].float(), torch.zeros(40), t1[0].float(), t0[1], t0[6]])
    assert torch.equal(got, want)
    x = torch.randn(20, 40, generator=g)
    off = torch.tensor([0, 3, 4, 9], dtype=torch.int32)
    rows = torch.tensor([1, 5, 7, 2, 10, 11, 12, 19, 0], dtype=torch.int32)
    got = ops.segment_mean(*dev(ops, x, off, rows)).cpu()
    want = torch.stack([x[[1, 5, 7]].mean(0), x[[2]].mean(0), x[[10, 11, 12, 19, 0]].mean(0)])
    assert (got - want).abs().max() < 1e-6


# (1, 6, 7, 16, *): 672 (window, head) pairs -- above the 640 up to which the fp32 kernel runs its three-wavefront K-through-LDS flavour: the
# one-wavefront flavour with K resident in registers (a 1024^2 image's stages 1 - 2); fp32 only (the large grid is there for that flavour)
@pytest.mark.parametrize("dtype,B,nWh,nWw,heads,shift",
                         [(dt, *c) for dt in DT for c in [(1, 1, 1, 2, 0), (1, 2, 3, 1, 6), (2, 2, 2, 2, 6)]] +
                         [(torch.float32, 1, 6, 7, 16, 6), (torch.float32, 1, 6, 7, 16, 0)])
def test_window_attention(ops, dtype, B, nWh, nWw, heads, shift):
    ws, hd = 12, 32
    C = heads * hd
    N = ws * ws
    g = torch.Generator().manual_seed(6 + heads)
    nW = nWh * nWw
    qkv = (torch.randn(B * nW * N, 3 * C, generator=g) * 0.7).to(dtype)
    table = torch.randn((2 * ws - 1) ** 2, heads, generator=g)
    from psalm_amd.synthetic import relative_position_index
    idx = relative_position_index(ws).view(-1)
    q_, k_, v_ = qkv.float().view(B * nW, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    attn = (q_ * hd ** -0.5) @ k_.transpose(-2, -1) + table[idx].view(N, N, heads).permute(2, 0, 1)[None]
    if shift:
        am = O.swin_shift_mask(nWh * ws, nWw * ws, ws, shift)
        attn = (attn.view(B, nW, heads, N, N) + am[None, :, None]).view(-1, heads, N, N)
    want = (attn.softmax(-1) @ v_).transpose(1, 2).reshape(B * nW * N, C)
    got = ops.window_attention(*dev(ops, qkv, table), B, nWh, nWw, heads, ws, shift).cpu().float()
    assert (got - want).abs().max() <= tol(dtype, want.abs().max())


@pytest.mark.parametrize("heads,shift,nWh,nWw", [(2, 0, 2, 2), (4, 6, 2, 2), (16, 6, 6, 7)])       # (the last: 672 pairs, the one-wavefront flavour)
def test_window_attention_split_output(ops, heads, shift, nWh, nWw):
    """psalm_window_attention_split == psalm_window_attention (fp32) followed by a split: one power-of-two scale per window from the bound
    max_j (a_inv[j] * par[0] + par[1]) over the window's rows (>= every |v| of the window), hi + lo reproduces the fp32 output to 22 bits."""
    B, ws, hd = 1, 12, 32
    C = heads * hd
    N, nW = ws * ws, nWh * nWw
    rows = B * nW * N
    g = torch.Generator().manual_seed(5 + heads)
    qkv = torch.randn(rows, 3 * C, generator=g) * 0.7
    table = torch.randn((2 * ws - 1) ** 2, heads, generator=g)
    a_inv = torch.exp2(torch.randint(-14, -8, (rows,), generator=g).float())
    vmax_row = qkv[:, 2 * C:].abs().amax(1)
    par = torch.tensor([float((vmax_row / a_inv).max()) * 1.01, 0.25])                 # a_inv[j] * par[0] + par[1] >= |v_j|
    d = ops.device
    ref = ops.window_attention(qkv.to(d), table.to(d), B, nWh, nWw, heads, ws, shift).cpu()
    got = ops.window_attention_split(qkv.to(d), table.to(d), a_inv.to(d), par.to(d), B, nWh, nWw, heads, ws, shift)
    Kp = got.Kp
    t, inv = got.t.cpu(), got.inv_scale.cpu().double()
    hi, lo = t[:, :C].double(), t[:, Kp:Kp + C].double()
    assert ((hi + lo) * inv[:, None] - ref.double()).abs().max() <= 2.0 ** -21 * ref.abs().max() and hi.abs().max() < 2.0 ** 13
    if Kp > C:
        assert (t[:, C:Kp] == 0).all() and (t[:, Kp + C:] == 0).all()
    bound = (a_inv * par[0] + par[1]).view(nW * B, N).amax(1).double()
    sb = bound / inv.view(nW * B, N)[:, 0]
    assert (inv.view(nW * B, N) == inv.view(nW * B, N)[:, :1]).all()                       # one scale per window
    assert (sb >= 2.0 ** 12 * (1 - 1e-6)).all() and (sb < 2.0 ** 13 * (1 + 1e-6)).all()


def _rope_tables(L, rot, theta=10000.0):
    inv = 1.0 / (theta ** (torch.arange(0, rot, 2, dtype=torch.float32) / rot))
    fr = torch.arange(L, dtype=torch.float32)[:, None] * inv[None]
    emb = torch.cat((fr, fr), -1)
    return emb.cos().contiguous(), emb.sin().contiguous()


@pytest.mark.parametrize("dtype", DT)
# 300: three 128-query blocks, ragged last;  (1, 100, 8) / (2, 70, 4): more (batch, head) pairs than the 1- and 2-head cases
@pytest.mark.parametrize("B,L,heads", [(1, 70, 2), (2, 150, 1), (2, 300, 2), (1, 129, 1), (1, 100, 8), (2, 70, 4)])
def test_causal_attention(ops, dtype, B, L, heads):
    hd, rot = 64, 32
    H = heads * hd
    g = torch.Generator().manual_seed(7)
    ld = 3 * H + 16
    buf = (torch.randn(B * L, ld, generator=g) * 0.8).to(dtype)
    key_mask = torch.ones(B, L, dtype=torch.uint8)
    if B > 1:
        key_mask[1, L - 20:] = 0                                  # right padding of the 2nd sample
    cos, sin = _rope_tables(L, rot)
    q = buf[:, 0:H].float().view(B, L, heads, hd).transpose(1, 2)
    k = buf[:, H + 8:2 * H + 8].float().view(B, L, heads, hd).transpose(1, 2)
    v = buf[:, 2 * H + 16:3 * H + 16].float().view(B, L, heads, hd).transpose(1, 2)

    def rope(x):
        xr = x[..., :rot]
        rh = torch.cat((-xr[..., rot // 2:], xr[..., : rot // 2]), -1)
        return torch.cat((xr * cos + rh * sin, x[..., rot:]), -1)
    w = rope(q) @ rope(k).transpose(2, 3) * hd ** -0.5
    allow = torch.tril(torch.ones(L, L, dtype=torch.bool))[None, None] & key_mask[:, None, None, :].bool()
    w = w.masked_fill(~allow, torch.finfo(torch.float32).min).softmax(-1)
    want = (w @ v).transpose(1, 2).reshape(B * L, H)
    out = torch.zeros(B * L, H + 32, dtype=dtype, device=ops.device)
    ops.causal_attention(buf.to(ops.device), 0, H + 8, 2 * H + 16, out, 32, *dev(ops, cos, sin, key_mask), B, L, heads, hd, rot)
    got = out[:, 32:].cpu().float()
    assert (got - want).abs().max() <= tol(dtype, want.abs().max())
    assert out[:, :32].abs().max() == 0


@pytest.mark.parametrize("B,L,heads", [(1, 70, 2), (2, 150, 1), (2, 70, 4)])
def test_causal_attention_split_output(ops, B, L, heads):
    """psalm_causal_attention_f32_split == psalm_causal_attention_f32 followed by a split under the given per-row scales: hi + lo reproduces
    the fp32 output to 22 bits, in the requested columns of the operand buffer, nothing else is written."""
    hd, rot = 64, 32
    Hh = heads * hd
    g = torch.Generator().manual_seed(17)
    ld = 3 * Hh + 16
    buf = (torch.randn(B * L, ld, generator=g) * 0.8)
    key_mask = torch.ones(B, L, dtype=torch.uint8)
    if B > 1:
        key_mask[1, L - 20:] = 0
    cos, sin = _rope_tables(L, rot)
    d = ops.device
    ref = torch.zeros(B * L, Hh, device=d)
    ops.causal_attention(buf.to(d), 0, Hh + 8, 2 * Hh + 16, ref, 0, *dev(ops, cos, sin, key_mask), B, L, heads, hd, rot)
    ref = ref.cpu()
    vmax = buf[:, 2 * Hh + 16:3 * Hh + 16].abs().max()
    inv = torch.exp2(torch.ceil(torch.log2(vmax)) - 12 + torch.randint(0, 4, (B * L,), generator=g).float())    # |v| / inv < 2^13
    off = 64
    Kp = (off + Hh + 63) // 64 * 64 + 64
    so = torch.zeros(B * L, 2 * Kp, dtype=torch.float16, device=d)
    ops.causal_attention_split(buf.to(d), 0, Hh + 8, 2 * Hh + 16, so, inv.to(d), off, *dev(ops, cos, sin, key_mask), B, L, heads, hd, rot)
    so = so.cpu()
    hi, lo = so[:, off:off + Hh].double(), so[:, Kp + off:Kp + off + Hh].double()
    rec = (hi + lo) * inv.double()[:, None]
    assert ((rec - ref.double()).abs() <= 2.0 ** -21 * ref.abs().double() + 2.0 ** -24 * inv.double()[:, None]).all()
    assert hi.abs().max() < 2.0 ** 13
    mask = torch.ones(2 * Kp, dtype=torch.bool)
    mask[off:off + Hh] = False
    mask[Kp + off:Kp + off + Hh] = False
    assert (so[:, mask] == 0).all()


@pytest.mark.parametrize("dtype", DT)
def test_mha_attention_and_mask(ops, dtype):
    B, Lq, heads, hd = 2, 10, 2, 32
    D = heads * hd
    h = w = 16
    Ht = Wt = 6
    Lk = Ht * Wt
    g = torch.Generator().manual_seed(8)
    masks = torch.randn(B, Lq, h, w, generator=g)
    masks[0, 3] = -1.0                                             # an all-masked row -> must attend everywhere (TD:647)
    am = F.interpolate(masks, size=(Ht, Wt), mode="bilinear", align_corners=False)
    want_mask = (am.sigmoid().flatten(2) < 0.5)
    got_mask, flags = ops.attn_mask(masks.to(ops.device), Ht, Wt)
    assert torch.equal(got_mask.cpu().bool(), want_mask)
    assert flags.cpu()[0, 3] == 1 and flags.cpu().sum() == want_mask.all(-1).sum()
    q = torch.randn(B * Lq, D + 8, generator=g).to(dtype)
    kv = torch.randn(B * Lk, 2 * D, generator=g).to(dtype)
    qq = q[:, 4:4 + D]
    kk, vv = kv[:, :D], kv[:, D:]
    wm = want_mask.clone()
    wm[wm.all(-1)] = False
    a = (qq.float().view(B, Lq, heads, hd).transpose(1, 2) * hd ** -0.5) @ kk.float().view(B, Lk, heads, hd).transpose(1, 2).transpose(-2, -1)
    a = a.masked_fill(wm[:, None], float("-inf")).softmax(-1)
    want = (a @ vv.float().view(B, Lk, heads, hd).transpose(1, 2)).transpose(1, 2).reshape(B * Lq, D)
    qd, kvd = q.to(ops.device), kv.to(ops.device)
    got = ops.mha_attention(qd[:, 4:4 + D], kvd[:, :D], kvd[:, D:], B, Lq, Lk, heads, got_mask, flags).cpu().float()
    assert (got - want).abs().max() <= tol(dtype, want.abs().max())
    # unmasked self-attention form
    a2 = ((qq.float().view(B, Lq, heads, hd).transpose(1, 2) * hd ** -0.5) @ qq.float().view(B, Lq, heads, hd).transpose(1, 2).transpose(-2, -1)).softmax(-1)
    want2 = (a2 @ qq.float().view(B, Lq, heads, hd).transpose(1, 2)).transpose(1, 2).reshape(B * Lq, D)
    got2 = ops.mha_attention(qd[:, 4:4 + D], qd[:, 4:4 + D], qd[:, 4:4 + D], B, Lq, Lq, heads).cpu().float()
    assert (got2 - want2).abs().max() <= tol(dtype, want2.abs().max())


@pytest.mark.parametrize("Lq,Lk,masked", [(100, 1024, True), (100, 100, False), (37, 200, True), (128, 4096, True)])
def test_mha_attention_mfma_split_kv(ops, Lq, Lk, masked):
    """Matrix-core split-KV MHA (transposed V operand, partial softmax states merged by the combine kernel) vs torch."""
    B, heads, hd = 1, 8, 32
    D = heads * hd
    g = torch.Generator().manual_seed(Lq + Lk)
    q = (torch.randn(B * Lq, D + 8, generator=g)).bfloat16()
    kbuf = (torch.randn(B * Lk, 3 * D, generator=g)).bfloat16()
    ldvt = (Lk + 7) // 8 * 8 + 8
    vt = torch.zeros(B * D, ldvt, dtype=torch.bfloat16)
    vt[:, :Lk] = (torch.randn(B * D, Lk, generator=g)).bfloat16()
    qq, kk = q[:, 8:8 + D], kbuf[:, D:2 * D]
    mask = flags = None
    a = (qq.float().view(B, Lq, heads, hd).transpose(1, 2) * hd ** -0.5) @ kk.float().view(B, Lk, heads, hd).transpose(1, 2).transpose(-2, -1)
    if masked:
        mask = (torch.rand(B, Lq, Lk, generator=g) < 0.6)
        mask[0, 1, :] = True                                       # an all-masked row: flagged -> attends everywhere (TD:647)
        mask[0, 2, : Lk - 3] = True                                # a row whose only visible keys sit in the last split
        flags = mask.all(-1)
        wm = mask.clone()
        wm[flags] = False
        a = a.masked_fill(wm[:, None], float("-inf"))
    want = (a.softmax(-1) @ vt[:, :Lk].float().view(B, heads, hd, Lk).transpose(-1, -2)).transpose(1, 2).reshape(B * Lq, D)
    d = ops.device
    qd, kd = q.to(d), kbuf.to(d)
    got = ops.mha_attention_t(qd[:, 8:8 + D], kd[:, D:2 * D], vt.to(d), B, Lq, Lk, heads,
                              mask.to(torch.uint8).to(d) if masked else None, flags.to(torch.uint8).to(d) if masked else None).cpu().float()
    assert (got - want).abs().max() <= tol(torch.bfloat16, want.abs().max())


def test_gemm_row_bias_transposed_projection(ops):
    """V^T = W_v . X^T with the bias broadcast along rows (how the predictor produces the transposed value operand)."""
    from psalm_amd import hip_ops as H
    g = torch.Generator().manual_seed(3)
    X = torch.randn(200, 64, generator=g).bfloat16()
    Wv = (torch.randn(96, 64, generator=g) * 0.3).bfloat16()
    bv = torch.randn(96, generator=g)
    want = (X.double() @ Wv.double().t() + bv.double()).t()
    d = ops.device
    got = ops.gemm(Wv.to(d), X.to(d), bv.to(d), act=H.ACT_BIAS_ROW, out_dtype=torch.bfloat16).cpu().double()
    assert got.shape == (96, 200) and (got - want).abs().max() <= 2 ** -8 * want.abs().max()


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,stride,pad,res", [(1, 9, 7, 64, 72, 3, 1, 1, False), (2, 8, 8, 128, 136, 3, 2, 1, True),
                                                               (1, 10, 6, 64, 64, 1, 2, 0, False), (1, 6, 6, 320, 96, 3, 1, 1, False)])
def test_conv2d_implicit_gemm(ops, B, H, W, Cin, Cout, k, stride, pad, res):
    """Implicit-GEMM convolution (padded taps read a zero page, filter tap block-uniform per K tile, split-K at large K) vs F.conv2d."""
    from psalm_amd import hip_ops as H_
    g = torch.Generator().manual_seed(B + H + Cin)
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16()
    wgt = (torch.randn(Cout, Cin, k, k, generator=g) * (Cin * k * k) ** -0.5).bfloat16()
    bias = torch.randn(Cout, generator=g)
    want = F.conv2d(x.float(), wgt.float(), bias, stride=stride, padding=pad)
    Ho, Wo = want.shape[-2:]
    want = want.permute(0, 2, 3, 1).reshape(B * Ho * Wo, Cout)
    r = torch.randn(B * Ho * Wo, Cout, generator=g).bfloat16() if res else None
    if res:
        want = torch.relu(want + r.float())
    d = ops.device
    xt = x.permute(0, 2, 3, 1).reshape(B * H * W, Cin).contiguous().to(d)
    wt = wgt.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(d)
    got = ops.conv2d_nhwc(xt, B, H, W, wt, k, stride, pad, bias=bias.to(d), residual=r.to(d) if res else None,
                          act=(H_.ACT_RELU | H_.ACT_POST_RESIDUAL) if res else H_.ACT_NONE).cpu().float()
    assert got.shape == want.shape and (got - want).abs().max() <= tol(torch.bfloat16, want.abs().max())


@pytest.mark.parametrize("B,H,W,C,shift", [(1, 14, 10, 64, 0), (2, 13, 25, 128, 6)])
def test_swin_window_merge_ln(ops, B, H, W, C, shift):
    ws = 12
    g = torch.Generator().manual_seed(C + shift)
    nWh, nWw = (H + ws - 1) // ws, (W + ws - 1) // ws
    win = torch.randn(B * nWh * nWw * ws * ws, C, generator=g).bfloat16()
    sc = torch.randn(B * H * W, C, generator=g)
    ga, be = torch.randn(C, generator=g), torch.randn(C, generator=g)
    d = ops.device
    want_x = ops.swin_window_merge(win.to(d), sc.to(d), B, H, W, ws, shift).cpu()
    want_h = F.layer_norm(want_x, (C,), ga, be, 1e-5)
    got_x, got_h = ops.swin_window_merge_ln(win.to(d), sc.to(d), ga.to(d), be.to(d), B, H, W, ws, shift, h_dtype=torch.bfloat16)
    assert torch.equal(got_x.cpu(), want_x)
    assert (got_h.cpu().float() - want_h).abs().max() <= tol(torch.bfloat16, want_h.abs().max())


@pytest.mark.parametrize("Q,C,k,thing,sig", [(100, 133, 100, True, False), (100, 1, 100, False, True), (12, 9, 12, True, False), (7, 3, 30, False, False),
                                           (100, 847, 100, False, False)])          # open-vocabulary class count (A-847): 84700 candidates
def test_topk_select(ops, Q, C, k, thing, sig):
    """radix-select top-k (value descending, ties by lowest flat index) + thing filter + mask-score product, LP:407-447 / 308-324."""
    g = torch.Generator().manual_seed(Q * C + k)
    stride = C + 1
    vals = torch.rand(Q, stride, generator=g)
    vals[:, :C] = (vals[:, :C] * 50).round() / 50           # many exact ties
    if sig:
        vals = vals * 8 - 4
    is_thing = (torch.rand(C, generator=g) < 0.6).to(torch.int32) if thing else None
    ms = torch.rand(Q, generator=g)
    d = ops.device
    sc, cl, qq, cnt = ops.topk_select(vals.to(d), C, k, is_thing.to(d) if thing else None, ms.to(d), apply_sigmoid=sig)
    n = int(cnt.item())
    flat = (vals[:, :C].sigmoid() if sig else vals[:, :C]).reshape(-1)
    order = sorted(range(flat.numel()), key=lambda i: (-flat[i].item(), i))[:min(k, flat.numel())]
    want = [(flat[i].item() * ms[i // C].item(), i % C, i // C) for i in order if (not thing or is_thing[i % C])]
    assert n == len(want)
    got = list(zip(sc.cpu()[:n].tolist(), cl.cpu()[:n].tolist(), qq.cpu()[:n].tolist()))
    for (gs, gc, gq), (ws_, wc, wq) in zip(got, want):
        assert (gc, gq) == (wc, wq) and abs(gs - ws_) < 1e-6


@pytest.mark.parametrize("Q,C,HW", [(100, 133, 128 * 3 + 40), (12, 9, 300), (128, 160, 256)])
def test_semantic_from_masks(ops, Q, C, HW):
    """fused sigmoid + (probs^T . sigmoid(masks)) on the matrix cores vs float64 on the same bf16-rounded operands."""
    g = torch.Generator().manual_seed(Q + C)
    mask = torch.randn(Q, HW, generator=g) * 4
    cls = torch.randn(Q, C + 1, generator=g) * 2
    d = ops.device
    probs, probsT, score, label = ops.class_softmax(cls.to(d), 128, probsT_dtype=torch.bfloat16)
    got, ms = ops.semantic_from_masks(mask.to(d), probsT, want_mask_score=True)
    got = got.cpu().double()
    assert torch.equal(ops.semantic_from_masks(mask.to(d), probsT).cpu().double(), got)
    pos = (mask > 0).double()
    want_ms = (mask.double().sigmoid() * pos).sum(1) / (pos.sum(1) + 1e-6)                      # LP:443-444
    assert (ms.cpu().double() - want_ms).abs().max() < 1e-5
    assert (ops.mask_scores(mask.to(d)).cpu() - ms.cpu()).abs().max() < 1e-5
    want = probsT.cpu().double()[:, :Q] @ mask.sigmoid().bfloat16().double()
    assert got.shape == (C, HW)
    err = (got - want).abs()
    # the device sigmoid (v_exp based) can land on the other side of a bf16 rounding boundary for a handful of (q, p) pairs: each
    # flip moves one term by one bf16 ulp (<= 2^-8 * prob), so the MEAN error stays at fp32-accumulation level and the MAX is bounded
    assert err.mean() <= 2e-6 * want.abs().max()
    assert err.max() <= 2 ** -7 * probsT.float().max().item() + 1e-5
    # and against the un-rounded reference formula (LP:402-406): bf16 operand rounding only
    full = torch.einsum("qc,qp->cp", cls.softmax(-1)[:, :-1].double(), mask.sigmoid().double())
    assert (got - full).abs().max() <= 2 ** -7 * full.abs().max()


def test_im2col_and_convs(ops):
    g = torch.Generator().manual_seed(9)
    img = torch.randn(2, 3, 18, 13, generator=g)
    wgt = torch.randn(16, 3, 4, 4, generator=g)
    cols = ops.patch_im2col(img.to(ops.device), 4, 48).cpu()
    want = F.conv2d(F.pad(img, (0, 3, 0, 2)), wgt, stride=4)       # pad to multiples of 4 (swin_trans.py:431-434)
    got = (cols @ wgt.view(16, 48).t()).view(2, 5, 4, 16).permute(0, 3, 1, 2)
    assert (got - want).abs().max() < 1e-4
    x = torch.randn(2, 9, 7, 8, generator=g)                       # NHWC
    for k, s, p in ((3, 1, 1), (3, 2, 1), (1, 2, 0)):
        w2 = torch.randn(5, 8, k, k, generator=g)
        cols = ops.im2col_nhwc(x.view(-1, 8).to(ops.device), 2, 9, 7, k, s, p).cpu()
        want = F.conv2d(x.permute(0, 3, 1, 2), w2, stride=s, padding=p)
        got = (cols @ w2.permute(0, 2, 3, 1).reshape(5, -1).t()).view(2, want.shape[2], want.shape[3], 5).permute(0, 3, 1, 2)
        assert (got - want).abs().max() < 1e-4


def test_resize_planes_row_walking_kernel(ops):
    """The fp32 -> fp32, W % 4 == 0, H >= 8 form of psalm_resize_planes (the mask up-sampling of LP:1401-1406: a block walks 8 output rows and
    re-fetches source values only when the source row pair changes) against F.interpolate, and against the one-pixel-per-thread kernel of the
    same entry point (reached through an output view that is not 16-byte aligned): 4x up-sampling, a non-integer scale, a crop, a row count
    that is not a multiple of 8, a down-sampling (source row pairs that skip), a single source row / column."""
    g = torch.Generator().manual_seed(21)
    cases = [((3, 24, 24), (96, 96), None), ((2, 20, 28), (52, 64), None), ((2, 20, 28), (50, 36), (17, 23)), ((1, 40, 40), (12, 16), None),
             ((2, 1, 5), (9, 8), None), ((2, 6, 1), (16, 4), None), ((1, 9, 7), (8, 1028), None), ((2, 6, 6), (5, 8), None)]      # (the last: H < 8, the 4-pixels-per-thread kernel)
    for shape, (H, W), crop in cases:
        x = torch.randn(shape, generator=g)
        src = x if crop is None else x[:, :crop[0], :crop[1]]
        want = F.interpolate(src[None], size=(H, W), mode="bilinear", align_corners=False)[0]
        xd = x.to(ops.device)
        got = ops.resize_planes(xd, H, W, crop=crop)
        assert (got.cpu() - want).abs().max() < 1e-5, (shape, H, W)
        buf = ops.empty(shape[0] * H * W + 1, dtype=torch.float32)
        other = ops.resize_planes(xd, H, W, crop=crop, out=buf[1:].view(shape[0], H, W))
        assert other.data_ptr() % 16 != 0
        assert torch.equal(got, other), (shape, H, W)      # one blend with its fused multiply-adds written out (bilin_blend): the same bits


def test_resize_and_layout(ops):
    g = torch.Generator().manual_seed(10)
    x = torch.randn(3, 16, 16, generator=g)
    for (H, W) in ((64, 64), (5, 7), (16, 16), (33, 20)):
        want = F.interpolate(x[None], size=(H, W), mode="bilinear", align_corners=False)[0]
        got = ops.resize_planes(x.to(ops.device), H, W).cpu()
        assert (got - want).abs().max() < 1e-5
    want = F.interpolate(x[None, :, :12, :10], size=(20, 24), mode="bilinear", align_corners=False)[0]
    got = ops.resize_planes(x.to(ops.device), 20, 24, crop=(12, 10)).cpu()
    assert (got - want).abs().max() < 1e-5
    small = torch.randn(2, 4, 4, 8, generator=g)
    lat = torch.randn(2, 8, 8, 8, generator=g)
    want = lat + F.interpolate(small.permute(0, 3, 1, 2), size=(8, 8), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    got = ops.upsample_add_nhwc(lat.view(-1, 8).to(ops.device), small.view(-1, 8).to(ops.device), 2, 4, 4, 8, 8).cpu().view(2, 8, 8, 8)
    assert (got - want).abs().max() < 1e-5
    t = torch.randn(2, 6, 5, generator=g)
    nhwc = ops.permute_layout(t.to(ops.device), 2, 6, 5, True).cpu()
    assert torch.equal(nhwc.view(2, 5, 6), t.transpose(1, 2))
    back = ops.permute_layout(nhwc.to(ops.device), 2, 6, 5, False).cpu()
    assert torch.equal(back, t)


@pytest.mark.parametrize("Lq,Lk,masked", [(100, 1024, True), (100, 100, False), (37, 203, True), (128, 700, True), (12, 64, True)])
def test_mha_attention_f32_matrix_core_split_kv(ops, Lq, Lk, masked):
    """fp32 matrix-core MHA (psalm_mha_attention_f32: 256-key chunks, partial softmax states merged by the combine kernel) vs torch fp32."""
    B, heads, hd = 2, 4, 32
    D = heads * hd
    g = torch.Generator().manual_seed(Lq + Lk)
    q = torch.randn(B * Lq, D + 8, generator=g)
    kv = torch.randn(B * Lk, 2 * D + 4, generator=g)
    qq, kk, vv = q[:, 8:8 + D], kv[:, :D], kv[:, D + 4:]
    mask = flags = None
    a = (qq.view(B, Lq, heads, hd).transpose(1, 2) * hd ** -0.5) @ kk.reshape(B, Lk, heads, hd).transpose(1, 2).transpose(-2, -1)
    if masked:
        mask = torch.rand(B, Lq, Lk, generator=g) < 0.6
        mask[0, 1, :] = True                                       # an all-masked row: flagged -> attends everywhere (TD:647)
        mask[1, 2, : Lk - 3] = True                                # a row whose only visible keys sit in the last chunk
        flags = mask.all(-1)
        wm = mask.clone()
        wm[flags] = False
        a = a.masked_fill(wm[:, None], float("-inf"))
    want = (a.softmax(-1) @ vv.reshape(B, Lk, heads, hd).transpose(1, 2)).transpose(1, 2).reshape(B * Lq, D)
    d = ops.device
    qd, kvd = q.to(d), kv.to(d)
    got = ops.mha_attention(qd[:, 8:8 + D], kvd[:, :D], kvd[:, D + 4:], B, Lq, Lk, heads,
                            mask.to(torch.uint8).to(d) if masked else None, flags.to(torch.uint8).to(d) if masked else None).cpu()
    assert (got - want).abs().max() <= 2e-5 * want.abs().max()


@pytest.mark.parametrize("Q,C,HW", [(100, 133, 128 * 3 + 40), (12, 9, 300), (128, 160, 256)])
def test_semantic_from_masks_x3_fp32_class(ops, Q, C, HW):
    """Split-f16 form of the fused semantic pass (precision="f16x3"): fp32-class agreement with the float64 product of the UNROUNDED
    probabilities and sigmoids (the bf16 form above is only good to 2^-8)."""
    g = torch.Generator().manual_seed(Q + C)
    mask = torch.randn(Q, HW, generator=g) * 4
    mask[:, ::3] -= 25.0                              # pixels whose masks are ALL strongly negative (sigmoid ~1e-11): most of a real image
    cls = torch.randn(Q, C + 1, generator=g) * 2
    d = ops.device
    probs, probsT, score, label = ops.class_softmax(cls.to(d), 128, probsT_dtype=torch.float32)
    got, ms = ops.semantic_from_masks(mask.to(d), probsT, want_mask_score=True)
    got = got.cpu().double()
    pos = (mask > 0).double()
    want_ms = (mask.double().sigmoid() * pos).sum(1) / (pos.sum(1) + 1e-6)
    assert (ms.cpu().double() - want_ms).abs().max() < 1e-5
    want = probsT.cpu().double()[:, :Q] @ mask.double().sigmoid()
    assert got.shape == (C, HW)
    # operands carried to 2^-22 relative, device sigmoid (v_exp) good to ~1e-6 relative: compare at 4e-6 of the output scale
    assert (got - want).abs().max() <= 4e-6 * want.abs().max() + 1e-7
    # ... and at every PIXEL relative to that pixel's own largest class score (per-pixel operand scale): label decisions at pixels
    # where every sigmoid is tiny see the same relative accuracy as anywhere else
    assert ((got - want).abs() <= 4e-6 * want.abs().amax(0, keepdim=True) + 1e-37).all()


@pytest.mark.parametrize("rows,C", [(37, 256), (5, 2048), (130, 64), (9, 200)])
def test_layernorm_split(ops, rows, C):
    """psalm_layernorm_split == LayerNorm (fp32) followed by psalm_split_f16 of y and of y + add, bit for bit."""
    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, C + 8, generator=g)[:, 4:4 + C] * 3 + 1                       # row-strided view
    ga, be = torch.randn(C, generator=g), torch.randn(C, generator=g)
    add = torch.randn(7, C, generator=g)
    d = ops.device
    xd = x.to(d) if x.is_contiguous() else torch.randn(rows, C + 8, generator=torch.Generator().manual_seed(rows + C)).to(d)[:, 4:4 + C] * 3 + 1
    y, s1, s2 = ops.layernorm_split(xd, ga.to(d), be.to(d), 1e-5, want_y=True, add=add.to(d))
    y_ref = ops.layernorm(xd, ga.to(d), be.to(d), 1e-5)
    assert torch.equal(y, y_ref)
    want = torch.nn.functional.layer_norm(xd.cpu(), (C,), ga, be, 1e-5)
    assert (y.cpu() - want).abs().max() < 1e-5 * want.abs().max() + 1e-6
    r1 = ops.split_f16(y_ref)
    r2 = ops.split_f16(ops.add_bcast(y_ref, add.to(d)))
    for got, ref in ((s1, r1), (s2, r2)):
        assert got.K == C and got.Kp == ref.Kp
        assert torch.equal(got.t.cpu().view(torch.int16), ref.t.cpu().view(torch.int16)) and torch.equal(got.inv_scale.cpu(), ref.inv_scale.cpu())
    # split-only form (no fp32 output)
    y0, s0, _ = ops.layernorm_split(xd, ga.to(d), be.to(d), 1e-5)
    assert y0 is None and torch.equal(s0.t.cpu().view(torch.int16), r1.t.cpu().view(torch.int16))


@pytest.mark.parametrize("B,H,W,C,shift", [(1, 14, 20, 32, 0), (2, 12, 12, 64, 6), (1, 30, 25, 128, 6)])
def test_swin_split_emitting_kernels(ops, B, H, W, C, shift):
    """f16x3 forms of the Swin LN-fused kernels == the fp32 kernels followed by psalm_split_f16, bit for bit."""
    ws = 12
    g = torch.Generator().manual_seed(H * W + C + shift)
    d = ops.device
    x = (torch.randn(B * H * W, C, generator=g) * 2 + 0.5).to(d)
    ga, be = torch.randn(C, generator=g).to(d), torch.randn(C, generator=g).to(d)
    a = ops.swin_window_gather_split(x, ga, be, B, H, W, ws, shift)
    ref = ops.swin_window_gather(x, ga, be, B, H, W, ws, shift).cpu().double()
    # (the fp32 gather kernel sums the LayerNorm statistics in a different lane order: equal to fp32 round-off, not bit for bit)
    Kp = a.Kp
    rec = (a.t.cpu()[:, :C].double() + a.t.cpu()[:, Kp:Kp + C].double()) * a.inv_scale.cpu().double()[:, None]
    assert (rec - ref).abs().max() <= 3e-6 * ref.abs().max()
    assert (a.t.cpu()[:, C:Kp] == 0).all() and bool((ref.abs().sum(1) == 0).eq(rec.abs().sum(1) == 0).all())      # padded tokens: zero rows
    nW = ((H + ws - 1) // ws) * ((W + ws - 1) // ws)
    win = torch.randn(B * nW * ws * ws, C, generator=g).to(d)
    x1, h1 = ops.swin_window_merge_ln_split(win, x, ga, be, B, H, W, ws, shift)
    x2, h2 = ops.swin_window_merge_ln(win, x, ga, be, B, H, W, ws, shift)
    h2s = ops.split_f16(h2)
    assert torch.equal(x1, x2)
    assert torch.equal(h1.t.cpu().view(torch.int16), h2s.t.cpu().view(torch.int16)) and torch.equal(h1.inv_scale.cpu(), h2s.inv_scale.cpu())


@pytest.mark.parametrize("case", ["nothing_confident", "all_void", "confident_but_empty_masks"])
@pytest.mark.parametrize("Hh,Ww", [(24, 40), (23, 41)])
def test_panoptic_degenerate_inputs_match_the_oracle(ops, case, Hh, Ww):
    """The early exits of class_name_panoptic_inference (llava_phi.py:340-347): no query above the object-mask threshold, every confident query
    labelled void, confident queries whose masks hold no pixel >= 0.5 -- an all-zero id map and an empty segments_info, from both arg-max kernels
    (4 pixels per thread / one pixel per thread), exactly as the oracle's restatement returns them."""
    from oracle import psalm_oracle as O
    Q, C = 20, 9
    g = torch.Generator().manual_seed(len(case) + Hh)
    mask = torch.randn(Q, Hh, Ww, generator=g) * 0.5 + (3.0 if case != "confident_but_empty_masks" else -6.0)
    cls = torch.randn(Q, C + 1, generator=g) * 0.1                      # flat: max probability ~ 0.1 < 0.8
    if case == "all_void":
        cls[:, C] += 9.0
    elif case == "confident_but_empty_masks":
        cls[torch.arange(Q), torch.randint(0, C, (Q,), generator=g)] += 9.0
    thing = [1, 0, 1, 0, 1, 0, 1, 0, 1]
    want_pan, want_info = O.panoptic_inference(cls, mask, thing)
    assert want_info == [] and int(want_pan.abs().sum()) == 0
    d = ops.device
    probs, probsT, score, label = ops.class_softmax(cls.to(d), 64)
    pan, info, ninfo = ops.panoptic(mask.to(d).contiguous(), score, label, torch.tensor(thing, dtype=torch.int32, device=d), C, 0.8, 0.8)
    assert int(ninfo.item()) == 0
    assert torch.equal(pan.cpu(), want_pan)


@pytest.mark.parametrize("Q,C,Hh,Ww", [(100, 133, 96, 96), (24, 9, 37, 41), (100, 133, 64, 100), (16, 5, 20, 12)])
def test_panoptic_matches_oracle_inference(ops, Q, C, Hh, Ww):
    """psalm_panoptic (class_name_panoptic_inference, llava_phi.py:325-386) on blobs that survive the score / overlap tests, against the
    oracle's restatement: identical id map and segments_info.  H * W % 4 == 0 takes the 4-pixels-per-thread arg-max kernel (r05), the
    odd sizes the one-pixel form -- the two must agree with each other through the oracle."""
    from oracle import psalm_oracle as O
    g = torch.Generator().manual_seed(Q + Hh * Ww)
    yy, xx = torch.meshgrid(torch.arange(Hh), torch.arange(Ww), indexing="ij")
    mask = torch.randn(Q, Hh, Ww, generator=g) * 0.7 - 5.0
    for q in range(Q):
        if q % 3 == 2:
            continue                                                   # a third of the queries: empty masks
        cy, cx = int(torch.randint(0, Hh, (1,), generator=g)), int(torch.randint(0, Ww, (1,), generator=g))
        r = int(torch.randint(1, max(2, min(Hh, Ww) // 12), (1,), generator=g))
        mask[q][((yy - cy).abs() <= r) & ((xx - cx).abs() <= r)] += 9.0
    mask[:, 0, :3] = 0.0                                                # sigmoid == 0.5 exactly: the >= 0.5 tests on the boundary
    cls = torch.randn(Q, C + 1, generator=g)
    hot = torch.randint(0, C + 1, (Q,), generator=g)
    cls[torch.arange(Q), hot] += torch.where(torch.rand(Q, generator=g) < 0.7, 8.0, 1.0)      # ~70 % confident queries, some of them void
    thing = [int(v) for v in (torch.rand(C, generator=g) < 0.6)]
    want_pan, want_info = O.panoptic_inference(cls, mask, thing)
    d = ops.device
    probs, probsT, score, label = ops.class_softmax(cls.to(d), (Q + 63) // 64 * 64)
    pan, info, ninfo = ops.panoptic(mask.to(d).contiguous(), score, label, torch.tensor(thing, dtype=torch.int32, device=d), C, 0.8, 0.8)
    n = int(ninfo.item())
    got_info = [{"id": a, "isthing": bool(b), "category_id": c} for a, b, c in info.cpu()[:n].tolist()]
    assert len(want_info) >= (3 if Q >= 24 else 1)                      # the case exercises the merge
    assert got_info == want_info
    assert torch.equal(pan.cpu(), want_pan)


@pytest.mark.parametrize("B,L,heads", [(1, 100, 8), (2, 70, 4), (1, 300, 16)])
def test_causal_attention_xcd_head_placement_is_bitwise_the_plain_placement(ops, B, L, heads):
    """r06: with heads * B a multiple of 8 the fp32 causal kernel re-maps its linear block id so that every query-tile block of a head runs on
    the same XCD (PSALM_TUNE_ATTN_XCD_HEADS).  The map is a bijection of the grid: the outputs are the same words as with blockIdx taken as is."""
    hd, rot = 64, 32
    H = heads * hd
    g = torch.Generator().manual_seed(70 + heads)
    buf = torch.randn(B * L, 3 * H, generator=g) * 0.8
    key_mask = torch.ones(B, L, dtype=torch.uint8)
    if B > 1:
        key_mask[1, L - 20:] = 0
    cos, sin = _rope_tables(L, rot)
    outs = []
    try:
        for v in (0, 1):
            ops.set_tuning(ops.TUNE_ATTN_XCD_HEADS, v)
            out = torch.zeros(B * L, H, device=ops.device)
            ops.causal_attention(buf.to(ops.device), 0, H, 2 * H, out, 0, *dev(ops, cos, sin, key_mask), B, L, heads, hd, rot)
            outs.append(out.cpu())
    finally:
        ops.set_tuning(ops.TUNE_ATTN_XCD_HEADS, 1)
    assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))
    assert outs[0].abs().max() > 0


@pytest.mark.parametrize("rows,C,with_add,with_ln2", [(100, 256, True, True), (100, 256, True, False), (7, 64, False, True), (33, 520, True, True)])
def test_layernorm_chain_is_bitwise_the_three_launches(ops, rows, C, with_add, with_ln2):
    """psalm_layernorm_chain (r06: the mask decoder's post-norm -> + query_pos -> decoder_norm as ONE launch) returns the words of
    psalm_layernorm3, psalm_add_bcast, psalm_layernorm3."""
    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, C, generator=g) * 3 + 0.7
    g1, b1, g2, b2 = (torch.randn(C, generator=g) for _ in range(4))
    add = torch.randn(rows, C, generator=g)
    d = ops.device
    y1 = ops.layernorm(x.to(d), g1.to(d), b1.to(d))
    y2 = ops.add_bcast(y1, add.to(d)) if with_add else None
    y3 = ops.layernorm(y1, g2.to(d), b2.to(d)) if with_ln2 else None
    c1, c2, c3 = ops.layernorm_chain(x.to(d), g1.to(d), b1.to(d), add.to(d) if with_add else None, g2.to(d) if with_ln2 else None,
                                     b2.to(d) if with_ln2 else None)
    assert torch.equal(c1.cpu().view(torch.int32), y1.cpu().view(torch.int32))
    assert (c2 is None) == (y2 is None) and (c2 is None or torch.equal(c2.cpu().view(torch.int32), y2.cpu().view(torch.int32)))
    assert (c3 is None) == (y3 is None) and (c3 is None or torch.equal(c3.cpu().view(torch.int32), y3.cpu().view(torch.int32)))
    want = F.layer_norm(x, (C,), g1, b1, 1e-5)
    assert (c1.cpu() - want).abs().max() < 3e-5 * want.abs().max()


def test_gemm_f32_pair_is_bitwise_two_skinny_gemms(ops):
    """psalm_gemm_f32_pair: [q | k] = (x + pos) . Wqk^T and v = x . Wv^T of the mask decoder's self-attention in one launch -- the same words as the
    two exact-fp32 skinny GEMMs."""
    Q, D = 100, 256
    g = torch.Generator().manual_seed(5)
    a0, a1 = torch.randn(Q, D, generator=g), torch.randn(Q, D, generator=g)
    w0, w1 = torch.randn(2 * D, D, generator=g) * 0.1, torch.randn(D, D, generator=g) * 0.1
    b0, b1 = torch.randn(2 * D, generator=g), torch.randn(D, generator=g)
    d = ops.device
    r0 = ops.gemm(a0.to(d), w0.to(d), b0.to(d))
    r1 = ops.gemm(a1.to(d), w1.to(d), b1.to(d))
    c0, c1 = ops.gemm_f32_pair(a0.to(d), w0.to(d), b0.to(d), a1.to(d), w1.to(d), b1.to(d))
    assert torch.equal(c0.cpu().view(torch.int32), r0.cpu().view(torch.int32)) and torch.equal(c1.cpu().view(torch.int32), r1.cpu().view(torch.int32))
    assert (c0.cpu().double() - (a0.double() @ w0.double().T + b0.double())).abs().max() < 1e-4


@pytest.mark.parametrize("C", [256, 2048, 512 + 256])
def test_segment_mean_vector_path_is_the_scalar_sum_in_the_same_order(ops, C):
    """psalm_segment_mean on fp32 rows of C % 256 == 0 columns (the LLM states) reads whole rows with 16-byte loads and the row index once per row
    (r06); same sums in the same order as the column-at-a-time loop -- compared word for word with a sequential fp32 sum -- incl. an empty segment
    and row strides wider than C."""
    g = torch.Generator().manual_seed(C)
    x = torch.randn(40, C + 64, generator=g)[:, :C]                    # row stride C + 64
    off = torch.tensor([0, 1, 4, 4, 11], dtype=torch.int32)
    rows = torch.tensor([3, 7, 39, 0, 5, 6, 8, 9, 30, 31, 2], dtype=torch.int32)
    d = ops.device
    got = ops.segment_mean(x.to(d), off.to(d), rows.to(d)).cpu()
    want = torch.zeros(4, C)
    for s_ in range(4):
        i0, i1 = int(off[s_]), int(off[s_ + 1])
        acc = torch.zeros(C)
        for i in range(i0, i1):
            acc = acc + x[int(rows[i])]
        want[s_] = acc * (torch.tensor(1.0) / (i1 - i0) if i1 > i0 else 0.0)
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))


@pytest.mark.parametrize("B,H,W,C,shift", [(1, 13, 11, 128, 6), (1, 25, 14, 256, 0), (1, 7, 9, 96, 6), (1, 12, 12, 136, 0)])
def test_row_kernels_several_rows_per_wavefront_same_bits(ops, B, H, W, C, shift):
    """r06 (PSALM_TUNE_ROW_GROUPS): rows of <= 128 / 256 columns share a wavefront four / two at a time in psalm_layernorm_split,
    psalm_swin_window_gather_split and psalm_swin_window_merge_ln_split -- same lane <-> column map inside a row, same reduction order: every word
    equals the one-row-per-wavefront form's, on row counts that leave the last wavefront partly empty and with padded (zero) window tokens."""
    ws = 12
    g = torch.Generator().manual_seed(B * H * W + C)
    d = ops.device
    x = (torch.randn(B * H * W, C, generator=g) * 2 + 0.5).to(d)
    ga, be = torch.randn(C, generator=g).to(d), torch.randn(C, generator=g).to(d)
    add = torch.randn(5, C, generator=g).to(d)
    nW = ((H + ws - 1) // ws) * ((W + ws - 1) // ws)
    win = torch.randn(B * nW * ws * ws, C, generator=g).to(d)
    res = []
    try:
        for groups in (0, 1):
            ops.set_tuning(ops.TUNE_ROW_GROUPS, groups)
            y, s1, s2 = ops.layernorm_split(x, ga, be, 1e-5, want_y=True, add=add)
            a = ops.swin_window_gather_split(x, ga, be, B, H, W, ws, shift)
            x1, h1 = ops.swin_window_merge_ln_split(win, x, ga, be, B, H, W, ws, shift)
            res.append([y, s1.t, s1.inv_scale, s2.t, s2.inv_scale, a.t, a.inv_scale, x1, h1.t, h1.inv_scale])
    finally:
        ops.set_tuning(ops.TUNE_ROW_GROUPS, 1)
    for p, q in zip(*res):
        p, q = p.cpu(), q.cpu()
        if p.dtype == torch.float16:
            p, q = p.view(torch.int16), q.view(torch.int16)
        assert torch.equal(p, q)


@pytest.mark.parametrize("B,H,W,C", [(1, 7, 9, 128), (2, 6, 6, 256), (1, 3, 5, 512)])
def test_patch_merge_ln_row_in_registers_same_bits(ops, B, H, W, C):
    """r06: psalm_patch_merge_ln on fp32 rows of 4C = 512 / 1024 / 2048 values reads the row once into registers (no per-element division, no re-reads);
    lane <-> column map, summation order and expressions are the generic kernel's: every word equal (odd H / W: zero-padded quadrants), and the
    LayerNorm of the 2 x 2 gather-concat to fp32 round-off."""
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(B, H, W, C, generator=g) * 2 + 0.3
    ga, be = 1 + 0.1 * torch.randn(4 * C, generator=g), 0.1 * torch.randn(4 * C, generator=g)
    d = ops.device
    outs = []
    try:
        for v in (0, 1):
            ops.set_tuning(ops.TUNE_ROW_GROUPS, v)
            outs.append(ops.patch_merge_ln(x.view(-1, C).to(d), ga.to(d), be.to(d), B, H, W).cpu())
    finally:
        ops.set_tuning(ops.TUNE_ROW_GROUPS, 1)
    assert torch.equal(outs[0], outs[1])
    xx = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    cat = torch.cat([xx[:, 0::2, 0::2], xx[:, 1::2, 0::2], xx[:, 0::2, 1::2], xx[:, 1::2, 1::2]], -1)
    want = F.layer_norm(cat.reshape(-1, 4 * C), (4 * C,), ga, be, 1e-5)
    assert (outs[1] - want).abs().max() <= 1e-5 * want.abs().max()
