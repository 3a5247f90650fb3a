"""MSDA forward: HIP kernel vs the oracle, including the reference's own known-answer test
(ops/test.py:24-63: seed 3, N,M,D=1,2,2, Lq,L,P=2,2,2, shapes [(6,4),(3,2)], fp32 tol rtol 1e-2 / atol 1e-3)."""
import ctypes

import numpy as np
import pytest
import torch

from ops_backend import ops  # noqa: F401
from oracle import build_oracle
from oracle import psalm_oracle as O


def _starts(shapes):
    s = [0]
    for h, w in shapes[:-1]:
        s.append(s[-1] + h * w)
    return s


def _c_ref(value, shapes, starts, loc, w, f64=False):
    libs = build_oracle.build()
    lib = ctypes.CDLL(libs["libmsda_ref_f64.so" if f64 else "libmsda_ref.so"])
    B, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    out = torch.empty(B, Lq, M * D)
    sh = (ctypes.c_int64 * (2 * L))(*[x for hw in shapes for x in hw])
    st = (ctypes.c_int64 * L)(*starts)
    lib.msda_forward_ref(ctypes.c_void_p(value.contiguous().data_ptr()), sh, st, ctypes.c_void_p(loc.contiguous().data_ptr()),
                         ctypes.c_void_p(w.contiguous().data_ptr()), ctypes.c_void_p(out.data_ptr()), B, S, M, D, L, Lq, P)
    return out


def _rand_case(seed, B, M, D, Lq, shapes, P, spread=0.3):
    g = torch.Generator().manual_seed(seed)
    S = sum(h * w for h, w in shapes)
    L = len(shapes)
    value = torch.randn(B, S, M, D, generator=g)
    loc = torch.rand(B, Lq, M, L, P, 2, generator=g) * (1 + 2 * spread) - spread     # some samples out of bounds
    w = torch.softmax(torch.randn(B, Lq, M, L * P, generator=g), -1).view(B, Lq, M, L, P)
    return value, loc, w


def test_oracle_known_answer_reference_test_py():
    """CPU: the reference's own check, oracle-vs-oracle (gather formula and C loops vs grid_sample formula)."""
    torch.manual_seed(3)
    N, M, D, Lq, L, P = 1, 2, 2, 2, 2, 2
    shapes = [(6, 4), (3, 2)]
    S = sum(h * w for h, w in shapes)
    for _ in range(2):
        value = torch.rand(N, S, M, D) * 0.01
        loc = torch.rand(N, Lq, M, L, P, 2)
        w = torch.rand(N, Lq, M, L, P) + 1e-5
        w /= w.sum(-1, keepdim=True).sum(-2, keepdim=True)
        ref = O.msda_core_grid_sample(value.double(), shapes, loc.double(), w.double()).float()
        got = O.msda_core(value, shapes, _starts(shapes), loc, w)
        assert torch.allclose(got, ref, rtol=1e-2, atol=1e-3)          # the reference's fp32 bar
        assert (got - ref).abs().max() < 1e-7
        assert (_c_ref(value, shapes, _starts(shapes), loc, w) - ref).abs().max() < 1e-7


def test_known_answer_hip(ops):
    torch.manual_seed(3)
    N, M, D, Lq, L, P = 1, 2, 4, 2, 2, 2      # D=4: the kernels vectorise 4 channels per lane (PSALM uses D=32)
    shapes = [(6, 4), (3, 2)]
    S = sum(h * w for h, w in shapes)
    value = torch.rand(N, S, M, D) * 0.01
    loc = torch.rand(N, Lq, M, L, P, 2)
    w = torch.rand(N, Lq, M, L, P) + 1e-5
    w /= w.sum(-1, keepdim=True).sum(-2, keepdim=True)
    ref = O.msda_core_grid_sample(value.double(), shapes, loc.double(), w.double()).float()
    got = ops.msda_forward(value.to(ops.device), shapes, _starts(shapes), loc.to(ops.device), w.to(ops.device)).cpu()
    assert torch.allclose(got, ref, rtol=1e-2, atol=1e-3)
    assert (got - ref).abs().max() < 1e-6


@pytest.mark.parametrize("B,M,D,Lq,shapes,P", [
    (1, 8, 32, 70, [(5, 7), (3, 4), (2, 2)], 4),
    (2, 2, 32, 33, [(9, 6), (4, 3)], 3),
    (1, 1, 8, 5, [(1, 1)], 1),
])
def test_random_vs_oracle(ops, B, M, D, Lq, shapes, P):
    value, loc, w = _rand_case(0, B, M, D, Lq, shapes, P)
    st = _starts(shapes)
    ref64 = _c_ref(value, shapes, st, loc, w, f64=True)
    ref = O.msda_core(value, shapes, st, loc, w)
    got = ops.msda_forward(value.to(ops.device), shapes, st, loc.to(ops.device), w.to(ops.device)).cpu()
    tol = 2e-6 * value.abs().max().item() * 4
    assert (got - ref64).abs().max() <= tol, (got - ref64).abs().max()
    assert (ref - ref64).abs().max() <= tol
    # bf16 value / bf16 out variant: error bounded by bf16 quantisation of value and output
    gotb = ops.msda_forward(value.bfloat16().to(ops.device), shapes, st, loc.to(ops.device), w.to(ops.device)).cpu().float()
    refb = _c_ref(value.bfloat16().float(), shapes, st, loc, w, f64=True)
    assert (gotb - refb).abs().max() <= 2 ** -8 * refb.abs().max() + 1e-6


def test_border_semantics(ops):
    """Locations exactly on the validity limits: h_im in {-1, -0.5, 0, H-1, H-0.5, H}."""
    shapes = [(4, 5)]
    H, W = shapes[0]
    vals = [(-1.0 + 0.5) / H, (-0.999 + 0.5) / H, 0.5 / H, (H - 1 + 0.5) / H, (H - 0.5 + 0.5) / H, (H + 0.5) / H, 0.37]
    pts = [(x * H / W if False else x, y) for y in vals for x in [0.5 / W, (W - 0.01 + 0.5) / W, (W + 0.5) / W, -0.5 / W]]
    Lq = len(pts)
    value = torch.arange(H * W * 4, dtype=torch.float32).view(1, H * W, 1, 4) / 7.0
    loc = torch.tensor(pts, dtype=torch.float32).view(1, Lq, 1, 1, 1, 2)
    w = torch.ones(1, Lq, 1, 1, 1)
    ref = _c_ref(value, shapes, [0], loc, w)
    got = ops.msda_forward(value.to(ops.device), shapes, [0], loc.to(ops.device), w.to(ops.device)).cpu()
    assert torch.equal(got, ref) or (got - ref).abs().max() < 1e-5


@pytest.mark.parametrize("B,shapes", [(2, [(4, 4), (8, 8), (16, 16)]), (1, [(8, 8), (16, 16), (32, 32)]), (1, [(8, 4), (16, 8), (24, 16)])])
def test_fused_matches_explicit(ops, B, shapes):
    """Fused kernel (softmax + location arithmetic in-kernel) == ms_deform_attn.py:101-110 done in torch + explicit op.
    B = 1 with level heights that are multiples of 8 takes the XCD-band query order (each XCD owns a horizontal band of every level)."""
    M, D, P = 8, 32, 4
    L = 3
    S = sum(h * w for h, w in shapes)
    st = _starts(shapes)
    g = torch.Generator().manual_seed(5)
    value = torch.randn(B, S, M * D, generator=g)
    ow = torch.randn(B, S, M * L * P * 3, generator=g) * 2.0
    off = ow[..., : M * L * P * 2].reshape(B, S, M, L, P, 2)
    aw = torch.softmax(ow[..., M * L * P * 2:].reshape(B, S, M, L * P), -1).view(B, S, M, L, P)
    refs = []
    for (H_, W_) in shapes:
        ry, rx = torch.meshgrid(torch.linspace(0.5, H_ - 0.5, H_), torch.linspace(0.5, W_ - 0.5, W_), indexing="ij")
        refs.append(torch.stack((rx.reshape(-1) / W_, ry.reshape(-1) / H_), -1))
    ref_pts = torch.cat(refs, 0)[None, :, None, None, None, :]
    norm = torch.tensor([[w_, h_] for h_, w_ in shapes], dtype=torch.float32)[None, None, None, :, None, :]
    loc = ref_pts + off / norm
    want = O.msda_core(value.view(B, S, M, D), shapes, st, loc, aw)
    got = ops.msda_fused(value.to(ops.device), shapes, st, ow.to(ops.device), M).cpu()
    assert (got - want).abs().max() < 2e-5 * want.abs().max().clamp(min=1)
    # r04, bf16 value: the bilinear taps of a sample are computed once per (query, head) and shared by its 4 channel-group lanes (quad
    # broadcasts); the form in which every lane computes all samples (policy 0) differs only in where the attention weight is multiplied in
    vb = value.to(torch.bfloat16).to(ops.device)
    quad = ops.msda_fused(vb, shapes, st, ow.to(ops.device), M, out_dtype=torch.float32).cpu()
    ops.msda_policy(0)
    try:
        per_lane = ops.msda_fused(vb, shapes, st, ow.to(ops.device), M, out_dtype=torch.float32).cpu()
    finally:
        ops.msda_policy(1)
    assert (quad - per_lane).abs().max() < 1e-5 * want.abs().max().clamp(min=1) and (quad - want).abs().max() < 2e-2 * want.abs().max().clamp(min=1)


def test_plugin_module_by_name_device_side_level_table(ops, monkeypatch):
    """Seam B1: import the extension module by the reference's name and call it the way MSDeformAttnFunction.forward does
    (ops/functions/ms_deform_attn_func.py:34-39): `spatial_shapes` / `level_start_index` are int64 TENSORS on the op's device and stay
    there -- the module must not copy them to the host (no `.tolist()` / `.item()` sync on a seam the reference calls asynchronously)."""
    import inspect
    import MultiScaleDeformableAttention as MSDA
    src = inspect.getsource(MSDA.ms_deform_attn_forward)
    assert ".tolist(" not in src and ".item(" not in src and ".cpu(" not in src
    monkeypatch.setattr(MSDA, "get_ops", lambda: ops)           # CPU run: the emulated library; GPU run: the product library
    B, M, D, Lq, shapes, P = 2, 8, 32, 45, [(5, 7), (3, 4), (2, 2)], 4
    value, loc, w = _rand_case(1, B, M, D, Lq, shapes, P)
    st = _starts(shapes)
    d = ops.device
    ss = torch.as_tensor(shapes, dtype=torch.long, device=d)                        # msdeformattn.py:139-145 builds exactly these
    lsi = torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]))
    got = MSDA.ms_deform_attn_forward(value.to(d), ss, lsi, loc.to(d), w.to(d), 128).cpu()
    ref64 = _c_ref(value, shapes, st, loc, w, f64=True)
    assert got.shape == (B, Lq, M * D)
    assert (got - ref64).abs().max() <= 2e-6 * value.abs().max().item() * 4
    assert torch.equal(got, ops.msda_forward(value.to(d), shapes, st, loc.to(d), w.to(d)).cpu())      # == the host-table entry point
    with pytest.raises(NotImplementedError):
        MSDA.ms_deform_attn_backward(None)
    with pytest.raises(RuntimeError):                                                                    # contiguity contract, ms_deform_attn_cuda.cu:33-43
        MSDA.ms_deform_attn_forward(value.to(d).transpose(1, 2), ss, lsi, loc.to(d), w.to(d), 128)


@pytest.mark.gpu
def test_real_shape_1024_vs_f64_oracle():
    """The op at the shape the 1024 x 1024 pixel decoder runs it at -- value (1, 21504, 8, 32), 3 levels (128^2, 64^2, 32^2), 4 points, one
    query per pixel -- against the fp64 C restatement of the CUDA op: both the plugin entry
    (explicit locations / weights) and the fused kernel the model launches (softmax + location arithmetic in-kernel)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ops_backend import make_ops
    ops = make_ops("hip")
    shapes = [(128, 128), (64, 64), (32, 32)]
    B, M, D, P, L = 1, 8, 32, 4, 3
    S = sum(h * w for h, w in shapes)
    assert S == 21504
    st = _starts(shapes)
    g = torch.Generator().manual_seed(11)
    value = torch.randn(B, S, M * D, generator=g)
    ow = torch.randn(B, S, M * L * P * 3, generator=g) * 2.0
    off = ow[..., : M * L * P * 2].reshape(B, S, M, L, P, 2)
    aw = torch.softmax(ow[..., M * L * P * 2:].reshape(B, S, M, L * P), -1).view(B, S, M, L, P)
    refs = []
    for (H_, W_) in shapes:
        ry, rx = torch.meshgrid(torch.linspace(0.5, H_ - 0.5, H_), torch.linspace(0.5, W_ - 0.5, W_), indexing="ij")
        refs.append(torch.stack((rx.reshape(-1) / W_, ry.reshape(-1) / H_), -1))
    ref_pts = torch.cat(refs, 0)[None, :, None, None, None, :]
    norm = torch.tensor([[w_, h_] for h_, w_ in shapes], dtype=torch.float32)[None, None, None, :, None, :]
    loc = (ref_pts + off / norm).contiguous()
    ref64 = _c_ref(value.view(B, S, M, D), shapes, st, loc, aw, f64=True)
    tol = 1e-5 * float(value.abs().max())
    got = ops.msda_forward(value.view(B, S, M, D).cuda(), shapes, st, loc.cuda(), aw.cuda()).cpu()
    assert float((got - ref64).abs().max()) <= tol
    fused = ops.msda_fused(value.cuda(), shapes, st, ow.cuda(), M).cpu()
    assert float((fused - ref64).abs().max()) <= 4 * tol          # in-kernel softmax (__expf) and location arithmetic in fp32
    dev = ops.msda_forward_dev(value.view(B, S, M, D).cuda(), torch.tensor(shapes, dtype=torch.int64).cuda(),
                               torch.tensor(st, dtype=torch.int64).cuda(), loc.cuda(), aw.cuda()).cpu()
    assert torch.equal(dev, got)


def test_device_level_table_is_validated_once_and_guarded(ops):
    """a level table that does not describe `value` must not become out-of-bounds gathers.  The binding checks each distinct
    device-side table once (then stays asynchronous); the kernel itself skips a level whose rows do not lie inside the S rows."""
    from psalm_amd import hip_ops as H
    d = ops.device
    shapes = [(4, 4), (2, 2)]
    value, loc, w = _rand_case(3, 1, 2, 8, 6, shapes, 2)
    sh = torch.tensor(shapes, dtype=torch.int64, device=d)
    st = torch.tensor(_starts(shapes), dtype=torch.int64, device=d)
    good = ops.msda_forward_dev(value.to(d), sh, st, loc.to(d), w.to(d)).cpu()
    assert torch.allclose(good, O.msda_core(value, shapes, _starts(shapes), loc, w), atol=1e-5)
    bad = torch.tensor([(4, 4), (3, 3)], dtype=torch.int64, device=d)          # 16 + 9 rows != 20
    with pytest.raises(H.PsalmHipError):
        ops.msda_forward_dev(value.to(d), bad, st, loc.to(d), w.to(d))
    # the kernel's own guard (what protects a captured graph, where the host check cannot run): level 1 does not fit -> contributes nothing
    key = (bad.data_ptr(), bad._version, st.data_ptr(), st._version, value.shape[1], 2)
    ops._msda_tables.add(key)
    out = ops.msda_forward_dev(value.to(d), bad, st, loc.to(d), w.to(d)).cpu()
    lvl0 = O.msda_core(value[:, :16], shapes[:1], [0], loc[:, :, :, :1], w[:, :, :, :1])
    assert torch.allclose(out, lvl0, atol=1e-5)
