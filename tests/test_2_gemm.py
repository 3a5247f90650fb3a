"""GEMM kernels (bf16 MFMA and exact-fp32 MFMA) vs a float64 torch matmul of the same (rounded) operands."""
import pytest
import torch

from ops_backend import ops  # noqa: F401
from psalm_amd import hip_ops as H


def _ref(a, w, bias, res, act, act_col_start):
    y = a.double() @ w.double().t()
    if bias is not None:
        y = y + bias.double()
    post = bool(act & H.ACT_POST_RESIDUAL)
    act &= 15
    if post:
        y = y + res.double()
        res = None
    if act:
        f = {H.ACT_RELU: torch.relu, H.ACT_GELU: torch.nn.functional.gelu,
             H.ACT_GELU_NEW: lambda v: torch.nn.functional.gelu(v, approximate="tanh")}[act]
        y = torch.cat([y[:, :act_col_start], f(y[:, act_col_start:])], 1)
    if res is not None:
        y = y + res.double()
    return y


CASES = [
    # M, N, K, a_dtype, w_dtype, c_dtype, bias, res, act, act_col_start
    (128, 128, 64, "f32", "bf16", "f32", True, True, H.ACT_NONE, 0),
    (100, 200, 48, "bf16", "bf16", "bf16", True, False, H.ACT_GELU, 0),
    (130, 300, 96, "f32", "bf16", "bf16", False, False, H.ACT_GELU_NEW, 128),
    (257, 130, 40, "bf16", "bf16", "f32", True, True, H.ACT_RELU, 0),
    (64, 136, 32, "f32", "f32", "f32", True, True, H.ACT_GELU, 0),
    (150, 129, 24, "f32", "f32", "f32", False, False, H.ACT_NONE, 0),
    (1, 8, 8, "f32", "f32", "bf16", True, False, H.ACT_RELU, 0),
    # direct-to-LDS fast path (A, W bf16, K % 64 == 0): 64- and 128-row tiles, ragged M/N edges, fused epilogues
    (100, 200, 64, "bf16", "bf16", "bf16", True, False, H.ACT_GELU, 0),
    (300, 130, 128, "bf16", "bf16", "f32", True, True, H.ACT_RELU, 0),
    (257, 260, 192, "bf16", "bf16", "bf16", True, True, H.ACT_GELU_NEW, 128),
    (33, 8, 64, "bf16", "bf16", "f32", False, False, H.ACT_NONE, 0),
    # ... with split-K (small tile grid, long K): partial slabs + reduce-epilogue kernel
    (70, 130, 1024, "bf16", "bf16", "f32", True, True, H.ACT_NONE, 0),
    (200, 100, 2048, "bf16", "bf16", "bf16", True, True, H.ACT_RELU | H.ACT_POST_RESIDUAL, 0),
    (65, 129, 1088, "bf16", "bf16", "bf16", False, False, H.ACT_GELU, 64),
]
DT = {"f32": torch.float32, "bf16": torch.bfloat16}


@pytest.mark.parametrize("M,N,K,ad,wd,cd,has_bias,has_res,act,acs", CASES)
def test_gemm(ops, M, N, K, ad, wd, cd, has_bias, has_res, act, acs):
    g = torch.Generator().manual_seed(M * 7 + N)
    # asymmetric operands (a transposed-output or swapped-fragment bug cannot hide)
    a = (torch.randn(M, K, generator=g) + torch.arange(M)[:, None] * 0.01).to(DT[ad])
    w = (torch.randn(N, K, generator=g) * 0.5 - torch.arange(N)[:, None] * 0.003).to(DT[wd])
    bias = torch.randn(N, generator=g) if has_bias else None
    res = torch.randn(M, N, generator=g).to(DT[cd]) if has_res else None
    a_eff = a.float().bfloat16().float() if wd == "bf16" else a.float()       # bf16 mode rounds A to bf16 when staging
    want = _ref(a_eff, w.float(), bias, res.float() if has_res else None, act, acs)
    d = ops.device
    got = ops.gemm(a.to(d), w.to(d), bias.to(d) if has_bias else None, res.to(d) if has_res else None, act, acs,
                   out_dtype=DT[cd]).cpu().double()
    scale = want.abs().max().item()
    tol = (2 ** -8 if cd == "bf16" else 2e-6) * scale + 1e-6
    err = (got - want).abs().max().item()
    assert err <= tol, f"max err {err} > {tol}"


def test_gemm_strided_views(ops):
    """A and C as column slices of wider buffers (how the fused [q|k|v|fc1] / [attn|mlp] buffers are used)."""
    g = torch.Generator().manual_seed(0)
    d = ops.device
    big_a = torch.randn(70, 96, generator=g).to(d)
    w = torch.randn(40, 32, generator=g).bfloat16().to(d)
    big_c = torch.zeros(70, 100, device=d)
    ops.gemm(big_a[:, 64:96], w, out=big_c[:, 8:48])
    want = big_a[:, 64:96].cpu().bfloat16().double() @ w.cpu().double().t()
    assert (big_c[:, 8:48].cpu().double() - want).abs().max() < 1e-4
    assert big_c[:, :8].abs().max() == 0 and big_c[:, 48:].abs().max() == 0


def test_gemm_fast_path_strided_views(ops):
    """bf16 A / C as column slices of a wider buffer through the direct-to-LDS kernel (Phi: big[:, 2H:] -> x)."""
    g = torch.Generator().manual_seed(1)
    d = ops.device
    big_a = (torch.randn(90, 200, generator=g)).bfloat16().to(d)
    w = (torch.randn(72, 128, generator=g) * 0.3).bfloat16().to(d)
    big_c = torch.zeros(90, 104, dtype=torch.bfloat16, device=d)
    ops.gemm(big_a[:, 72:200], w, out=big_c[:, 16:88])
    want = big_a[:, 72:200].cpu().double() @ w.cpu().double().t()
    err = (big_c[:, 16:88].cpu().double() - want).abs().max()
    assert err <= 2 ** -8 * want.abs().max()
    assert big_c[:, :16].abs().max() == 0 and big_c[:, 88:].abs().max() == 0


@pytest.mark.parametrize("policy", [256, 128, 64, 12864])
@pytest.mark.parametrize("M,N,K,cd,split", [(300, 260, 128, "bf16", False), (270, 300, 1088, "f32", True)])
def test_gemm_forced_tiles(ops, policy, M, N, K, cd, split):
    """Every tile configuration of the direct-to-LDS kernel (256x256 / 8 waves with its two-pass LDS epilogue, 128x128, 64x128)
    on the same ragged problem, with and without split-K."""
    g = torch.Generator().manual_seed(policy + M)
    a = (torch.randn(M, K, generator=g) + torch.arange(M)[:, None] * 0.01).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.5 - torch.arange(N)[:, None] * 0.003).bfloat16()
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g).to(DT[cd])
    want = _ref(a.float(), w.float(), bias, res.float(), H.ACT_GELU, 0)
    d = ops.device
    ops.gemm_tile_policy(policy)
    try:
        got = ops.gemm(a.to(d), w.to(d), bias.to(d), res.to(d), H.ACT_GELU, 0, out_dtype=DT[cd]).cpu().double()
    finally:
        ops.gemm_tile_policy(0)
    scale = want.abs().max().item()
    tol = (2 ** -8 if cd == "bf16" else 4e-6) * scale + 1e-6
    assert (got - want).abs().max().item() <= tol


@pytest.mark.parametrize("policy", [1323, 1324, 128128])
def test_gemm_bk32_ring(ops, policy):
    """128x128 configuration with 32-deep K tiles (64-byte LDS rows, 4-slot swizzle) and a 3 / 4-deep operand ring; and with
    128-deep K tiles (256-byte rows, 16-slot swizzle)."""
    g = torch.Generator().manual_seed(policy)
    M, N, K = 260, 200, 512 if policy == 128128 else 448
    a = (torch.randn(M, K, generator=g) + torch.arange(M)[:, None] * 0.01).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.5 - torch.arange(N)[:, None] * 0.003).bfloat16()
    bias = torch.randn(N, generator=g)
    want = a.double() @ w.double().t() + bias.double()
    d = ops.device
    ops.gemm_tile_policy(policy)
    try:
        got = ops.gemm(a.to(d), w.to(d), bias.to(d), out_dtype=torch.float32).cpu().double()
        assert (got - want).abs().max().item() <= 4e-6 * want.abs().max().item() + 1e-6
    finally:
        ops.gemm_tile_policy(1282)


@pytest.mark.parametrize("ph8", [2568, 2569, 2570])
@pytest.mark.parametrize("M,N,K,cd", [(300, 520, 128, "f32"), (257, 256, 192, "bf16"), (520, 300, 256, "f32"), (260, 250, 448, "f32"),
                                      (200, 256, 1536, "f32")])
def test_gemm_256_ph8_schedule(ops, M, N, K, cd, ph8):
    """256x256 tiles with the 4-phases-per-K-tile K loop (wave rows one barrier interval apart, half-tile copies refilled one
    phase after their last read, counted vmcnt): 2 / 3 / 4 / 7 K tiles = prologue only, one and several steady-state tiles, and the
    drain; ragged M and N; bias + activation + residual epilogue; the last case is split along K (fp32 slabs).  ph8 = 2568: the
    half-tile copies are issued in the read segment; 2569: inside the MFMA segment."""
    g = torch.Generator().manual_seed(M * 7 + K)
    a = (torch.randn(M, K, generator=g) + torch.arange(M)[:, None] * 0.01).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.5 - torch.arange(N)[:, None] * 0.003).bfloat16()
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g).to(DT[cd])
    want = _ref(a.float(), w.float(), bias, res.float(), H.ACT_GELU, 0)
    d = ops.device
    ops.gemm_tile_policy(256)
    ops.gemm_tile_policy(ph8)
    try:
        for _ in range(2):
            got = ops.gemm(a.to(d), w.to(d), bias.to(d), res.to(d), H.ACT_GELU, 0, out_dtype=DT[cd]).cpu().double()
            tol = (2 ** -8 if cd == "bf16" else 4e-6) * want.abs().max().item() + 1e-6
            assert (got - want).abs().max().item() <= tol
    finally:
        ops.gemm_tile_policy(2570)                     # library default
        ops.gemm_tile_policy(0)


@pytest.mark.parametrize("K", [64, 192, 256, 320, 512, 832, 2048])
@pytest.mark.parametrize("cd", ["f32", "bf16"])
def test_gemm_skinny_k_chunking(ops, K, cd):
    """M <= 128 skinny kernel: every lane's K range is walked in unguarded chunks of 8, then 4, then single MFMA steps
    (K / 64 = 1, 3, 4, 5, 8, 13, 32 steps per wave cover all combinations); residual + activation epilogue with batched loads."""
    g = torch.Generator().manual_seed(K)
    M, N = 100, 96
    a = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16()
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g).to(DT[cd])
    want = _ref(a.float(), w.float(), bias, res.float(), H.ACT_RELU, 0)
    d = ops.device
    assert ops.gemm_describe(M, N, K, True, True)[0] == 2            # really the skinny path
    got = ops.gemm(a.to(d), w.to(d), bias.to(d), res.to(d), H.ACT_RELU, 0, out_dtype=DT[cd]).cpu().double()
    tol = (2 ** -8 if cd == "bf16" else 4e-6) * want.abs().max().item() + 1e-6
    assert (got - want).abs().max().item() <= tol


@pytest.mark.parametrize("policy", [643, 644])
def test_gemm_64x128_deep_ring(ops, policy):
    """64x128 configuration with a 3 / 4-deep operand ring (2 / 3 K tiles of copies in flight across the barrier)."""
    g = torch.Generator().manual_seed(policy)
    M, N, K = 150, 200, 448                      # M <= 192 -> 64x128 tiles; 7 K steps: prologue, steady state, drain
    a = (torch.randn(M, K, generator=g) + torch.arange(M)[:, None] * 0.01).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.5 - torch.arange(N)[:, None] * 0.003).bfloat16()
    bias = torch.randn(N, generator=g)
    want = a.double() @ w.double().t() + bias.double()
    d = ops.device
    ops.gemm_tile_policy(64)
    ops.gemm_tile_policy(policy)
    try:
        for _ in range(2):
            got = ops.gemm(a.to(d), w.to(d), bias.to(d), out_dtype=torch.float32).cpu().double()
            assert (got - want).abs().max().item() <= 4e-6 * want.abs().max().item() + 1e-6
    finally:
        ops.gemm_tile_policy(640)                      # automatic ring depth
        ops.gemm_tile_policy(0)


def test_gemm_three_stage_ring(ops):
    """128x128 configuration with a 3-deep operand ring (copies of 2 tiles in flight across the barrier, counted vmcnt)."""
    g = torch.Generator().manual_seed(11)
    M, N, K = 300, 200, 448                      # 7 K-steps: prologue, steady state and drain of the ring
    a = (torch.randn(M, K, generator=g) + torch.arange(M)[:, None] * 0.01).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.5 - torch.arange(N)[:, None] * 0.003).bfloat16()
    want = a.double() @ w.double().t()
    d = ops.device
    ops.gemm_tile_policy(1283)
    try:
        for _ in range(3):
            got = ops.gemm(a.to(d), w.to(d), out_dtype=torch.float32).cpu().double()
            assert (got - want).abs().max().item() <= 4e-6 * want.abs().max().item() + 1e-6
    finally:
        ops.gemm_tile_policy(1282)


@pytest.mark.parametrize("M,N,K", [(70, 256, 1024), (130, 512, 128), (200, 2048, 2048), (40, 4096, 1024), (33, 8192, 1024), (50, 1028, 1024)])
def test_gemm_ln_fused(ops, M, N, K):
    """psalm_gemm_ln: C = a.w^T + bias + residual (fp32) and LayerNorm(C) from one call -- fused into the split-K reduction
    when the problem is split (cases 1 and 3), a GEMM + LayerNorm launch otherwise (case 2)."""
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16()
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ga, be = torch.randn(N, generator=g), torch.randn(N, generator=g)
    want_c = a.double() @ w.double().t() + bias.double() + res.double()
    want_ln = torch.nn.functional.layer_norm(want_c, (N,), ga.double(), be.double(), 1e-5)
    d = ops.device
    for ln_dtype, tol_ln in ((torch.float32, 2e-5), (torch.bfloat16, 2 ** -7)):
        c, ln = ops.gemm_ln(a.to(d), w.to(d), bias.to(d), res.to(d), ga.to(d), be.to(d), 1e-5, ln_dtype=ln_dtype)
        assert (c.cpu().double() - want_c).abs().max() <= 4e-6 * want_c.abs().max()
        assert (ln.cpu().double() - want_ln).abs().max() <= tol_ln * want_ln.abs().max()


# ------------------------------------------------------------------------------------------- split-f16 ("X3") fp32-class GEMM
def _split_ref(x):
    """psalm_split_f16 restated with torch: per-row power-of-two scale, hi = f16(x s), lo = f16(x s - hi)."""
    amax = x.abs().amax(1, keepdim=True)
    e = torch.floor(torch.log2(amax.clamp(min=1e-30)))
    s = torch.where(amax > 0, torch.exp2(13 - e), torch.ones_like(amax))
    xs = x * s
    hi = xs.half()
    lo = (xs - hi.float()).half()
    return hi, lo, 1.0 / s


@pytest.mark.parametrize("rows,K", [(5, 64), (37, 200), (9, 8), (130, 1096)])
def test_split_f16(ops, rows, K):
    g = torch.Generator().manual_seed(rows + K)
    x = torch.randn(rows, K, generator=g) * torch.exp2(torch.randint(-30, 30, (rows, 1), generator=g).float())
    x[0, :] = 0.0                                   # all-zero row: unscaled
    x[min(1, rows - 1), K // 2] = 0.0
    sp = ops.split_f16(x.to(ops.device))
    Kp = (K + 63) // 64 * 64
    assert sp.t.shape == (rows, 2 * Kp) and sp.K == K and sp.Kp == Kp
    hi, lo, inv = _split_ref(x)
    t = sp.t.cpu()
    assert torch.equal(t[:, :K], hi) and torch.equal(t[:, Kp:Kp + K], lo)
    assert (t[:, K:Kp] == 0).all() and (t[:, Kp + K:] == 0).all()
    assert torch.equal(sp.inv_scale.cpu(), inv.view(-1))
    # reconstruction: 22-bit operand (hi + lo) / s == x to 2^-21 of the row maximum
    rec = (t[:, :K].double() + t[:, Kp:Kp + K].double()) * sp.inv_scale.cpu().double()[:, None]
    assert ((rec - x.double()).abs() <= 2.0 ** -21 * x.abs().amax(1, keepdim=True).double() + 1e-300).all()


X3_CASES = [
    # M, N, K, bias, res, act, policy          (policy: forced tile height of the direct-to-LDS kernel, 0 = automatic)
    (100, 72, 64, True, True, H.ACT_RELU, 0),          # skinny kernel (M <= 128)
    (33, 40, 200, True, False, H.ACT_NONE, 0),         # ... K padded to 256 per part
    (128, 256, 512, False, True, H.ACT_GELU, 0),       # ... multi-chunk K
    (300, 260, 128, True, True, H.ACT_GELU, 128),
    (300, 260, 128, True, True, H.ACT_GELU, 64),
    (300, 520, 192, True, True, H.ACT_GELU_NEW, 256),  # 256x256 phased K loop (9 K tiles incl. the hi/lo part boundaries)
    (257, 256, 64, True, False, H.ACT_NONE, 256),      # 3 K tiles: prologue + drain only
    (270, 300, 704, True, True, H.ACT_RELU | H.ACT_POST_RESIDUAL, 128),    # split-K slabs (scaled partials) + reduce
    (200, 130, 704, False, False, H.ACT_NONE, 64),     # 64x128 with the 3-deep ring (K range >= 1024)
]


# K loop forms: 3300 automatic (32-deep slices, 2 stages = 3303) on every case; the others (3301 64-deep slices, 3302 / 3304 deeper rings, 3305 K panel)
# on the tiled (policy != 0) cases
# (r05: 3307 = 64 x 128 tiles, 64-deep slices, 3 stages; 3308 = 128 x 128 tiles, 32-deep slices, 3 stages -- each on its own tile height)
X3_PARAMS = ([c + (3300,) for c in X3_CASES] + [c + (k,) for c in X3_CASES if c[6] in (128, 64) for k in (3301, 3302, 3304, 3305)] +
             [c + (3307 if c[6] == 64 else 3308,) for c in X3_CASES if c[6] in (128, 64)])


@pytest.mark.parametrize("M,N,K,has_bias,has_res,act,policy,kloop", X3_PARAMS)
def test_gemm_x3(ops, M, N, K, has_bias, has_res, act, policy, kloop):
    """fp32-class accuracy from f16 matrix instructions: error vs the float64 product of the UNROUNDED fp32 operands within a few
    2^-22 of sum_k |a||w| (the exact-fp32 MFMA kernel sits at ~2^-24 of it; a bf16 GEMM at 2^-9)."""
    g = torch.Generator().manual_seed(M * 7 + N + K)
    a = (torch.randn(M, K, generator=g) + torch.arange(M)[:, None] * 0.01) * torch.exp2(torch.randint(-6, 6, (M, 1), generator=g).float())
    w = (torch.randn(N, K, generator=g) * 0.5 - torch.arange(N)[:, None] * 0.003) * torch.exp2(torch.randint(-6, 6, (N, 1), generator=g).float())
    bias = torch.randn(N, generator=g) if has_bias else None
    res = torch.randn(M, N, generator=g) if has_res else None
    want = _ref(a, w, bias, res, act, 0)
    mag = (a.abs().double() @ w.abs().double().t())
    d = ops.device
    ops.gemm_tile_policy(policy)
    ops.gemm_tile_policy(kloop)
    try:
        wsp = ops.split_f16(w.to(d))                                  # weights are split once; activations on the fly
        got = ops.gemm(a.to(d), wsp, bias.to(d) if has_bias else None, res.to(d) if has_res else None, act, 0).cpu().double()
        got2 = ops.gemm_x3(a.to(d), w.to(d), bias.to(d) if has_bias else None, res.to(d) if has_res else None, act, 0).cpu().double()
    finally:
        ops.gemm_tile_policy(3300)
        ops.gemm_tile_policy(0)
    assert torch.equal(got, got2)
    tol = 6 * 2.0 ** -22 * mag + 4e-7 * want.abs() + 1e-6           # operand split 3 x 2^-22, fp32 accumulation / epilogue round-off
    bad = (got - want).abs() > tol * (8 if act & 15 else 1)
    assert not bad.any(), f"max err {(got - want).abs().max().item()} rel-to-mag {((got - want).abs() / mag.clamp(min=1e-30)).max().item()}"


def split_bound_par(w, bias, g1=0.0, g0=0.0):
    """bound_par of psalm_gemm_x3_split for weight w (N,K) / bias (N,): {2^14 max_n sum_k |w_nk|, max |bias|, g1, g0}."""
    l1 = float(w.abs().double().sum(1).max()) * (1 + 1e-5)
    return torch.tensor([2.0 ** 14 * l1, float(bias.abs().max()) if bias is not None else 0.0, g1, g0], dtype=torch.float32)


@pytest.mark.parametrize("M,N,K,act,policy,col_start,col_off,glob", [
    (300, 256, 128, H.ACT_GELU, 64, 0, 0, False),          # whole output in split form (Swin fc1 -> fc2)
    (300, 264, 192, H.ACT_RELU, 128, 0, 64, False),        # 128 x 128 tiles, N % 128 != 0, written behind 64 other columns
    (100, 72, 64, H.ACT_NONE, 64, 0, 0, False),            # M <= 128 (this form has no skinny kernel; the fp32 twin is forced onto the same tiles)
    (300, 768, 192, H.ACT_GELU_NEW, 256, 512, 256, True),  # Phi layout: [k | v .. | fc1] -> fc1 columns only, row-independent floor on
    (131, 384, 64, H.ACT_GELU_NEW, 64, 256, 8, True),
    (300, 520, 128, H.ACT_GELU, 256, 256, 0, True),        # 256 x 256 tiles, two LDS passes (m-tiles of both wave rows per pass), ragged last tiles
    (300, 264, 192, H.ACT_RELU, 128, 64, 0, False),        # col_start inside a tile: its fp32 columns leave through the same LDS image
])
def test_gemm_x3_split_output(ops, M, N, K, act, policy, col_start, col_off, glob, kloop=3300):
    """psalm_gemm_x3_split: the columns >= col_start of act(a.w^T + b) leave the GEMM as split-f16 rows [hi | lo] under a per-row power-of-two
    scale derived from a magnitude bound.  Checked: the bound holds (|hi| < 2^13), the scales are powers of two, hi + lo reproduces the fp32
    result of psalm_gemm_x3 to 2^-21 (22-bit operand), the columns below col_start equal psalm_gemm_x3's bit for bit, and the NEXT GEMM on the
    emitted operand equals the GEMM on the fp32 values split by psalm_split_f16 to fp32 round-off."""
    g = torch.Generator().manual_seed(M + 3 * N + K + col_off)
    a = (torch.randn(M, K, generator=g) + torch.arange(M)[:, None] * 0.01) * torch.exp2(torch.randint(-6, 6, (M, 1), generator=g).float())
    w = torch.randn(N, K, generator=g) * 0.2
    bias = torch.randn(N, generator=g)
    d = ops.device
    Ns = N - col_start
    Kp_out = (col_off + Ns + 63) // 64 * 64
    g1, g0 = (2.0 ** 14 * 3.0, 0.5) if glob else (0.0, 0.0)
    par = split_bound_par(w[col_start:], bias[col_start:], g1, g0).to(d)
    ops.gemm_tile_policy(policy)
    ops.gemm_tile_policy(kloop)
    try:
        asp, wsp = ops.split_f16(a.to(d)), ops.split_f16(w.to(d))
        want = ops.gemm_x3(asp, wsp, bias.to(d), None, act, col_start)
        so = torch.zeros(M, 2 * Kp_out, dtype=torch.float16, device=d)
        inv = torch.full((M,), -1.0, device=d)
        out = torch.full((M, N), 7.0, device=d) if col_start else None
        ops.gemm_x3_split(asp, wsp, bias.to(d), act, so, inv, par, split_col_off=col_off, split_col_start=col_start, act_col_start=col_start,
                          out=out, global_rows=glob)
    finally:
        ops.gemm_tile_policy(3300)
        ops.gemm_tile_policy(0)
    so, inv, want = so.cpu(), inv.cpu(), want.cpu()
    if col_start:
        # (bit for bit where whole tiles are fp32; a tile that straddles col_start forms acc * scale + bias in its own order: fp32 round-off)
        assert torch.allclose(out.cpu()[:, :col_start], want[:, :col_start], rtol=2e-6, atol=1e-6) and (out.cpu()[:, col_start:] == 7.0).all()
        if col_start % 256 == 0:
            assert torch.equal(out.cpu()[:, :col_start], want[:, :col_start])
    m, e = torch.frexp(inv)
    assert (m == 0.5).all() and (inv > 0).all()                                      # powers of two
    hi = so[:, col_off:col_off + Ns].float()
    lo = so[:, Kp_out + col_off:Kp_out + col_off + Ns].float()
    assert hi.abs().max() < 2.0 ** 13
    v = want[:, col_start:].double()
    rec = (hi.double() + lo.double()) * inv.double()[:, None]
    assert ((rec - v).abs() <= 2.0 ** -21 * v.abs() + 2.0 ** -24 * inv.double()[:, None]).all()
    # the bound behind the scale: a_scale[r] * par[0] + par[1] (and the global floor) -- the scale puts it in [2^12, 2^13)
    a_inv = asp.inv_scale.cpu().double()
    bound = torch.maximum(a_inv * float(par[0]) + float(par[1]), a_inv.max() * g1 + g0 if glob else torch.zeros(()).double())
    sb = bound / inv.double()
    assert (sb >= 2.0 ** 12 * (1 - 1e-6)).all() and (sb < 2.0 ** 13 * (1 + 1e-6)).all()
    assert (v.abs().amax(1) <= bound).all()
    # untouched columns of the buffer stay zero (K padding of the consumer)
    mask = torch.ones(2 * Kp_out, dtype=torch.bool)
    mask[col_off:col_off + Ns] = False
    mask[Kp_out + col_off:Kp_out + col_off + Ns] = False
    assert (so[:, mask] == 0).all()
    # consumer: GEMM over the emitted operand vs over psalm_split_f16(fp32 values)
    w2 = torch.randn(40, col_off + Ns, generator=g)
    x = torch.zeros(M, col_off + Ns)
    x[:, col_off:] = want[:, col_start:]
    a_emit = H.SplitF16(so.to(d), inv.to(d), col_off + Ns)
    y1 = ops.gemm_x3(a_emit, w2.to(d)).cpu().double()
    y2 = ops.gemm_x3(x.to(d), w2.to(d)).cpu().double()
    mag = x.abs().double() @ w2.abs().double().t()
    assert ((y1 - y2).abs() <= 8 * 2.0 ** -22 * mag + 1e-9).all()


@pytest.mark.parametrize("M,N,K,act,policy,col_start,col_off,glob", [
    (300, 256, 128, H.ACT_GELU, 64, 0, 0, False),
    (300, 512, 192, H.ACT_RELU, 128, 0, 64, False),
    (300, 768, 192, H.ACT_GELU_NEW, 256, 512, 256, True),     # Phi layout on the 256 x 256 tile: [.. | fc1], fc1 columns paired
])
@pytest.mark.parametrize("products", [3, 1])
def test_gemm_x3_split_output_paired_stores(ops, M, N, K, act, policy, col_start, col_off, glob, products):
    """`paired`: with the W rows / bias >= col_start permuted by so_pair_perm the emitted operand, its scales and the fp32 columns are
    bit for bit those of the un-permuted call (same dot products; only the store path differs: registers -> 4-byte stores, no LDS pass).
    products = 1: the same through the K-panel kernels of the one-product side mode (psalm_gemm_x3_set_products), on all three tile sizes."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-4, 4, (M, 1), generator=g).float())
    w = torch.randn(N, K, generator=g) * 0.2
    bias = torch.randn(N, generator=g)
    d = ops.device
    Ns = N - col_start
    Kp_out = (col_off + Ns + 127) // 128 * 128
    par = split_bound_par(w[col_start:], bias[col_start:], 2.0 ** 14 * 3.0 if glob else 0.0, 0.5 if glob else 0.0).to(d)
    perm = torch.cat([torch.arange(col_start), col_start + H.Ops.so_pair_perm(Ns)])
    res = []
    ops.gemm_tile_policy(policy)
    ops.x3_products(products)
    try:
        for paired in (False, True):
            wq, bq = (w[perm], bias[perm]) if paired else (w, bias)
            asp = ops.split_f16(a.to(d))
            wsp = ops.split_f16(wq.to(d))
            so = torch.zeros(M, 2 * Kp_out, dtype=torch.float16, device=d)
            inv = torch.full((M,), -1.0, device=d)
            out = torch.full((M, N), 7.0, device=d) if col_start else None
            ops.gemm_x3_split(asp, wsp, bq.to(d), act, so, inv, par, split_col_off=col_off, split_col_start=col_start, act_col_start=col_start,
                              out=out, global_rows=glob, paired=paired)
            res.append((so.cpu(), inv.cpu(), out.cpu() if out is not None else None))
    finally:
        ops.x3_products(3)
        ops.gemm_tile_policy(0)
    (so0, inv0, out0), (so1, inv1, out1) = res
    assert (so0[:, col_off:col_off + Ns] != 0).any()
    assert torch.equal(inv0, inv1)
    if col_start:
        assert torch.equal(out0, out1)
    # The two store paths run the same source-level arithmetic on the same accumulators; on the emulator the emitted words are identical.  On
    # the hardware the two template instantiations may round ONE fp32 intermediate of the activation differently (fused vs separate multiply-
    # add chosen per instantiation by the compiler: r04b / r04c), which moves a lo word -- so: the VALUES agree to 2^-22 of the row scale and
    # all but a sliver of the words are bit-identical.
    w0, w1 = so0.view(torch.int16), so1.view(torch.int16)
    ndiff = int((w0 != w1).sum())
    v0 = (so0[:, col_off:col_off + Ns].double() + so0[:, Kp_out + col_off:Kp_out + col_off + Ns].double())
    v1 = (so1[:, col_off:col_off + Ns].double() + so1[:, Kp_out + col_off:Kp_out + col_off + Ns].double())
    worst = float((v0 - v1).abs().max())
    assert worst <= 2.0 ** -9 and ndiff <= 0.002 * w0.numel(), (ndiff, w0.numel(), worst)      # scaled values < 2^13: 2^-9 = 2^-22 of the range
    if ops.is_emu:
        assert ndiff == 0
    with pytest.raises(H.PsalmHipError):                       # a column count the permutation is not defined for
        ops.gemm_x3_split(asp, ops.split_f16(w[:N - 8].to(d)), bias[:N - 8].to(d), act, so, inv, par, split_col_off=col_off,
                          split_col_start=col_start, act_col_start=col_start, out=out, paired=True)


@pytest.mark.parametrize("M,N,K,has_bias,has_res,act", [
    (300, 520, 192, True, True, H.ACT_GELU_NEW),                    # 6 slices: steady state + drain
    (257, 256, 64, True, False, H.ACT_NONE),                        # 2 slices: prologue + drain only
    (270, 300, 704, True, True, H.ACT_RELU | H.ACT_POST_RESIDUAL),  # split-K slabs over the TRUE K range (4 x 192 / 128)
    (515, 300, 128, False, False, H.ACT_RELU),                      # three row tiles, ragged last
])
def test_gemm_x3_256_phased_slice_form(ops, M, N, K, has_bias, has_res, act):
    """r04: the 256 x 256 phased K loop on 32-deep slices (policy 2581; four operand images per stage, three products per phase) -- same
    accuracy contract as every split-f16 GEMM form, and the launch really is the <.., 32, 3, 2, ..> instantiation."""
    ops.gemm_tile_policy(2581)
    try:
        test_gemm_x3(ops, M, N, K, has_bias, has_res, act, 256, 3300)
        assert "256, 256, 2, 4, 2, false, 32, 3, 2" in ops.gemm_last_kernel()
    finally:
        ops.gemm_tile_policy(H.Ops.GEMM_X3_256_DEFAULT)


def test_gemm_x3_256_phased_slice_form_split_output(ops):
    """... with split-f16 output: LDS-transposed stores (two passes) and paired stores straight from the accumulators"""
    ops.gemm_tile_policy(2581)
    try:
        test_gemm_x3_split_output(ops, 300, 768, 192, H.ACT_GELU_NEW, 256, 512, 256, True)
        assert "256, 256, 2, 4, 2, false, 32, 3, 2, true" in ops.gemm_last_kernel() or "skinny" in ops.gemm_last_kernel() or "64, 128" in ops.gemm_last_kernel()
        test_gemm_x3_split_output(ops, 300, 520, 128, H.ACT_GELU, 256, 256, 0, True)
        test_gemm_x3_split_output_paired_stores(ops, 300, 768, 192, H.ACT_GELU_NEW, 256, 512, 256, True, 3)
    finally:
        ops.gemm_tile_policy(H.Ops.GEMM_X3_256_DEFAULT)


@pytest.mark.parametrize("M,N,K", [(899, 512, 192), (300, 520, 192), (515, 300, 128), (257, 256, 64), (270, 300, 704), (1, 256, 64)])
def test_gemm_x3_256_phased_slice_form_padding_tiles_left_out(ops, M, N, K):
    """Policy 2582 (the default since r04p): the phased slice kernel leaves out the matrix instructions and fragment reads of the 32-row
    m-tiles of a wave that lie entirely below row M (<.., 32, 4, 2, ..>; Phi's M = 899: wave row 1 of the last row of tiles keeps 1 m-tile of
    4).  Those rows are never stored: every output word must equal the default kernel's (2581), with and without split-K, with split-f16
    output, on M that leaves 0 / 1 / 2 / 3 m-tiles of a wave, and on M = 1 (wave row 1 of the only tile leaves out everything)."""
    g = torch.Generator().manual_seed(5 + M)
    d = ops.device
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * 0.3
    bias = torch.randn(N, generator=g)
    asp, wsp = ops.split_f16(a.to(d)), ops.split_f16(w.to(d))
    outs, names = [], []
    for pol in (2581, 2582):
        ops.gemm_tile_policy(256)
        ops.gemm_tile_policy(pol)
        try:
            outs.append(ops.gemm_x3(asp, wsp, bias.to(d), None, H.ACT_GELU).cpu())
            names.append(ops.gemm_last_kernel())
        finally:
            ops.gemm_tile_policy(H.Ops.GEMM_X3_256_DEFAULT)
            ops.gemm_tile_policy(0)
    assert "32, 3, 2" in names[0] and "32, 4, 2" in names[1], names
    assert torch.equal(outs[0], outs[1])
    want = torch.nn.functional.gelu(a.double() @ w.double().t() + bias.double())
    assert (outs[1].double() - want).abs().max() <= 3e-6 * want.abs().max() * (K ** 0.5)


def test_gemm_x3_256_phased_slice_form_padding_tiles_left_out_split_output(ops):
    ops.gemm_tile_policy(2582)
    try:
        test_gemm_x3_split_output(ops, 300, 768, 192, H.ACT_GELU_NEW, 256, 512, 256, True)
        assert "256, 256, 2, 4, 2, false, 32, 4, 2, true" in ops.gemm_last_kernel() or "skinny" in ops.gemm_last_kernel() or "64, 128" in ops.gemm_last_kernel()
        test_gemm_x3_split_output_paired_stores(ops, 300, 768, 192, H.ACT_GELU_NEW, 256, 512, 256, True, 3)
    finally:
        ops.gemm_tile_policy(H.Ops.GEMM_X3_256_DEFAULT)


def test_gemm_x3_split_output_k_panel_form(ops):
    """the automatic K loop of the 128 / 64-row tiles is the 32-deep slice form (policy 3303); the K-panel form (3305) stays selectable"""
    test_gemm_x3_split_output(ops, 300, 264, 192, H.ACT_RELU, 128, 0, 64, False, kloop=3305)
    test_gemm_x3_split_output(ops, 131, 384, 64, H.ACT_GELU_NEW, 64, 256, 8, True, kloop=3305)


@pytest.mark.parametrize("loose_bits", [0, 10, 13])
def test_gemm_x3_split_output_loose_bound(ops, loose_bits):
    """How loose may the magnitude bound behind the output scale be?  With the bound inflated by 2^loose_bits the emitted operand still
    stands for the fp32 values to max(2^-21 |v|, 2^(loose_bits - 36) * row bound-free maximum): hi + lo keeps 22 bits until lo reaches the
    f16 subnormal floor, so at 2^10 the absolute error stays below 2^-26 of the row maximum (under one fp32 ulp of the largest element)
    and the NEXT GEMM's result moves by less than its own fp32 round-off; at 2^13 it reaches the fp32 ulp (the documented limit)."""
    M, N, K = 200, 256, 128
    g = torch.Generator().manual_seed(31 + loose_bits)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * 0.2
    bias = torch.randn(N, generator=g) * 0.1
    d = ops.device
    par = split_bound_par(w, bias) * torch.tensor([2.0 ** loose_bits, 2.0 ** loose_bits, 1.0, 1.0])
    ops.gemm_tile_policy(64)
    try:
        asp, wsp = ops.split_f16(a.to(d)), ops.split_f16(w.to(d))
        want = ops.gemm_x3(asp, wsp, bias.to(d), None, H.ACT_GELU, 0).cpu().double()
        so = torch.zeros(M, 2 * N, dtype=torch.float16, device=d)
        inv = torch.zeros(M, device=d)
        ops.gemm_x3_split(asp, wsp, bias.to(d), H.ACT_GELU, so, inv, par.to(d))
    finally:
        ops.gemm_tile_policy(0)
    so, inv = so.cpu(), inv.cpu().double()
    rec = (so[:, :N].double() + so[:, N:].double()) * inv[:, None]
    rowmax = want.abs().amax(1, keepdim=True)
    # typical looseness of the un-inflated bound on this data is ~2^5..2^6 (L1 norm vs the actual dot products)
    assert ((rec - want).abs() <= 2.0 ** -21 * want.abs() + 2.0 ** (loose_bits + 7 - 37) * rowmax).all()
    w2 = torch.randn(48, N, generator=g)
    y1 = ops.gemm_x3(H.SplitF16(so.to(d), inv.float().to(d), N), w2.to(d)).cpu().double()
    y2 = want @ w2.double().t()
    mag = want.abs() @ w2.abs().double().t()
    assert ((y1 - y2).abs() <= (8 * 2.0 ** -22 + 2.0 ** (loose_bits + 7 - 37)) * mag + 1e-9).all()


@pytest.mark.parametrize("M,N,K,policy,want_y", [(270, 256, 704, 128, True), (150, 128, 1408, 64, False), (300, 192, 128, 64, True)])
def test_gemm_x3_ln_split(ops, M, N, K, policy, want_y):
    """psalm_gemm_x3_ln_split == psalm_gemm_x3 (+ residual) followed by psalm_layernorm_split: x equal to fp32 round-off of the summation
    order, the LayerNorm output in split form carries the exact row-maximum scale and reproduces LN(x) to 22 bits.  First two cases: split-K
    (the fused row pass); last: un-split GEMM (LayerNorm as its own kernel)."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-3, 3, (M, 1), generator=g).float())
    w = torch.randn(N, K, generator=g) * 0.1
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    gamma, beta = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.1
    d = ops.device
    ops.gemm_tile_policy(policy)
    try:
        asp, wsp = ops.split_f16(a.to(d)), ops.split_f16(w.to(d))
        x0 = ops.gemm_x3(asp, wsp, bias.to(d), res.to(d))
        x, hs, y = ops.gemm_x3_ln_split(asp, wsp, bias.to(d), res.to(d), gamma.to(d), beta.to(d), 1e-5, want_y=want_y)
    finally:
        ops.gemm_tile_policy(0)
    x0, x = x0.cpu(), x.cpu()
    assert torch.equal(x, x0)
    want = torch.nn.functional.layer_norm(x.double(), (N,), gamma.double(), beta.double(), 1e-5)
    if want_y:
        assert (y.cpu().double() - want).abs().max() <= 4e-6 * want.abs().max()
    inv = hs.inv_scale.cpu().double()
    hi, lo = hs.t.cpu()[:, :N].double(), hs.t.cpu()[:, N:].double()
    rec = (hi + lo) * inv[:, None]
    assert (rec - want).abs().max() <= 4e-6 * want.abs().max()
    rmax = (hi + lo).abs().amax(1)
    assert (rmax >= 2.0 ** 13 * (1 - 1e-3)).all() and (rmax < 2.0 ** 14).all()          # psalm_split_f16's row scaling


def test_gemm_x3_flag_routes_fp32_gemms(ops):
    """Ops.x3 = True (precision='f16x3'): float32 x float32 gemm() calls run in split-f16 arithmetic; bf16 GEMMs are untouched."""
    g = torch.Generator().manual_seed(3)
    a, w = torch.randn(70, 96, generator=g), torch.randn(50, 96, generator=g)
    d = ops.device
    want = ops.gemm_x3(a.to(d), w.to(d)).cpu()
    exact = ops.gemm(a.to(d), w.to(d)).cpu()
    ops.x3 = True
    try:
        got = ops.gemm(a.to(d), w.to(d)).cpu()
        gb = ops.gemm(a.bfloat16().to(d), w.bfloat16().to(d), out_dtype=torch.float32).cpu()
    finally:
        ops.x3 = False
    assert torch.equal(got, want)
    assert (got - exact).abs().max() < 1e-4 and (gb - exact).abs().max() > 1e-3


@pytest.mark.parametrize("M,N,K,cd,has_res,act", [(100, 256, 256, "f32", True, H.ACT_RELU), (134, 256, 2048, "f32", False, H.ACT_NONE),
                                                   (100, 2048, 256, "bf16", False, H.ACT_RELU), (3, 100, 256, "f32", False, H.ACT_NONE),
                                                   (192, 40, 72, "f32", True, H.ACT_GELU), (100, 134, 264, "f32", False, H.ACT_NONE)])
def test_gemm_f32_skinny_exact(ops, M, N, K, cd, has_res, act):
    """Exact-fp32 skinny kernel (M <= 192, fp32 operands): the mask decoder's M = 100 GEMMs in the fp32 / f16x3 modes."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g) + torch.arange(M)[:, None] * 0.01
    w = torch.randn(N, K, generator=g) * 0.5 - torch.arange(N)[:, None] * 0.003
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g).to(DT[cd]) if has_res else None
    want = _ref(a, w, bias, res.float() if has_res else None, act, 0)
    d = ops.device
    got = ops.gemm(a.to(d), w.to(d), bias.to(d), res.to(d) if has_res else None, act, 0, out_dtype=DT[cd]).cpu().double()
    tol = (2 ** -8 if cd == "bf16" else 4e-6) * want.abs().max().item() + 1e-6
    assert (got - want).abs().max().item() <= tol


@pytest.mark.parametrize("B,H,W,C,k,s,p", [(2, 9, 7, 8, 3, 1, 1), (1, 10, 12, 16, 3, 2, 1), (2, 8, 8, 24, 1, 2, 0), (1, 6, 5, 256, 3, 1, 1),
                                             (1, 5, 4, 264, 3, 2, 1)])      # the last two: few long rows -- a block per row (r06), ragged K padding
def test_im2col_split_equals_im2col_then_split(ops, B, H, W, C, k, s, p):
    """psalm_im2col_split_f16 == psalm_im2col_nhwc followed by psalm_split_f16, bit for bit (hi, lo, scales, zero padding)."""
    g = torch.Generator().manual_seed(H * W + C)
    x = (torch.randn(B * H * W, C, generator=g) * torch.exp2(torch.randint(-8, 8, (B * H * W, 1), generator=g).float())).to(ops.device)
    a = ops.im2col_split(x, B, H, W, k, s, p)
    b = ops.split_f16(ops.im2col_nhwc(x, B, H, W, k, s, p))
    assert a.K == b.K and a.Kp == b.Kp and torch.equal(a.t.cpu().view(torch.int16), b.t.cpu().view(torch.int16))
    assert torch.equal(a.inv_scale.cpu(), b.inv_scale.cpu())


@pytest.mark.gpu
def test_gemm_x3_auto_slice_selection_on_gpu():
    """Long K on a grid that 64 x 128 tiles fill without split-K (Swin stage-2 fc2 shape): select_fast_config picks the slice form by itself;
    same result as the K-panel form to fp32 round-off."""
    from psalm_amd.hip_ops import get_ops
    ops = get_ops()
    g = torch.Generator().manual_seed(11)
    a, w = torch.randn(4096, 2048, generator=g).cuda(), torch.randn(512, 2048, generator=g).cuda()
    assert ops.gemm_describe(4096, 512, 3 * 2048, x3=True)[:3] == (1, 64, 128)
    auto = ops.gemm_x3(a, w)
    ops.gemm_tile_policy(64)                       # a forced tile policy disables the automatic rule: K-panel form on the same tiles
    try:
        panel = ops.gemm_x3(a, w)
    finally:
        ops.gemm_tile_policy(0)
    want = a.double() @ w.double().t()
    assert (auto.double() - want).abs().max() <= 3e-6 * want.abs().max() and (auto - panel).abs().max() <= 3e-6 * want.abs().max()


@pytest.mark.parametrize("policy", [256, 128, 64])
@pytest.mark.parametrize("mode", ["relu_mid_tile", "post_residual", "bias_row", "plain_strided"])
def test_gemm_x3_direct_epilogue_edges(ops, policy, mode):
    """The fp32 epilogue that stores straight from the accumulators (buffer-descriptor bounds instead of per-element branches): ragged last
    row / column tiles, a column-sliced output whose neighbours must stay untouched, ReLU starting inside a tile, ReLU after the residual,
    per-row bias -- on every tile configuration."""
    M, N, K = 301, 296, 128                                       # 301 = 256 + 45 rows, 296 = 256 + 40 columns (ragged both ways)
    g = torch.Generator().manual_seed(policy + len(mode))
    d = ops.device
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * 0.3
    bias = torch.randn(M if mode == "bias_row" else N, generator=g)
    res = torch.randn(M, N, generator=g) if mode in ("post_residual", "relu_mid_tile") else None
    act, acs = {"relu_mid_tile": (H.ACT_RELU, 72), "post_residual": (H.ACT_RELU | H.ACT_POST_RESIDUAL, 0), "bias_row": (H.ACT_BIAS_ROW, 0),
                "plain_strided": (H.ACT_NONE, 0)}[mode]
    big = torch.full((M + 3, N + 24), 7.0, device=d)              # sentinel-filled: rows below M and the columns either side stay 7
    out = big[:M, 8:8 + N]
    big_res = None
    if res is not None:
        big_res = torch.zeros(M, N + 16, device=d)
        big_res[:, 4:4 + N] = res.to(d)
    ops.gemm_tile_policy(policy)
    try:
        ops.gemm_x3(a.to(d), w.to(d), bias.to(d), big_res[:, 4:4 + N] if res is not None else None, act, acs, out=out)
    finally:
        ops.gemm_tile_policy(0)
    y = a.double() @ w.double().t()
    y = y + (bias.double()[:, None] if mode == "bias_row" else bias.double())
    if mode == "post_residual":
        y = torch.relu(y + res.double())
    elif mode == "relu_mid_tile":
        y = torch.cat([y[:, :acs], torch.relu(y[:, acs:])], 1) + res.double()
    mag = a.abs().double() @ w.abs().double().t()
    got = big.cpu().double()
    assert ((got[:M, 8:8 + N] - y).abs() <= 6 * 2.0 ** -22 * mag + 4e-7 * y.abs() + 1e-6).all()
    assert (got[M:] == 7).all() and (got[:, :8] == 7).all() and (got[:, 8 + N:] == 7).all()


def test_gemm_x3_row_bias_is_never_read_past_its_m_entries():
    """A per-ROW bias (ACT_BIAS_ROW: the transposed projections of the mask decoder, model.py `pr.lvl*.v`) has M entries while the output has N >> M
    columns.  The fp32 epilogue's operands are fetched before the K loop (r05); fetching `bias[column]` there for such a GEMM reads up to N - M
    floats past the tensor.  Host emulator only: the bias sits at the very end of a page whose successor is PROT_NONE, so one stray read is a
    segfault, not a silently unused value."""
    from ops_backend import make_ops
    ops = make_ops("emu")                                             # guard-page check: host pointers, so the emulator build only
    import ctypes, mmap
    M, N, K = 128, 1024, 64
    page = mmap.PAGESIZE
    buf = mmap.mmap(-1, 2 * page)
    addr = ctypes.addressof(ctypes.c_char.from_buffer(buf))
    libc = ctypes.CDLL(None, use_errno=True)
    libc.mprotect.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    assert libc.mprotect(addr + page, page, 0) == 0                # PROT_NONE behind the bias
    try:
        whole = torch.frombuffer(buf, dtype=torch.float32, count=page // 4)
        g = torch.Generator().manual_seed(9)
        bias = whole[page // 4 - M:]
        bias.copy_(torch.randn(M, generator=g))
        a = torch.randn(M, K, generator=g)
        w = torch.randn(N, K, generator=g) * 0.3
        for policy in (64, 128):
            ops.gemm_tile_policy(policy)
            try:
                got = ops.gemm_x3(a, w, bias, None, H.ACT_BIAS_ROW, 0)
            finally:
                ops.gemm_tile_policy(0)
            y = a.double() @ w.double().t() + bias.double()[:, None]
            assert (got.double() - y).abs().max() <= 3e-6 * y.abs().max() * K ** 0.5
        del bias, whole
    finally:
        libc.mprotect(addr + page, page, 3)
    buf.close()


def test_split_output_bound_debug_check_heavy_tailed_weights(ops):
    """the L1 bound behind the split-output scale was validated on Gaussian weights only.  With heavy-tailed weights
    (a few outlier channels carry most of every row's L1 norm while typical activations never excite them) the bound is much looser; the
    debug check (Ops.debug_bounds / PSALM_DEBUG_BOUNDS=1) measures it per call and refuses an operand past the documented 2^14."""
    M, N, K = 200, 256, 256
    g = torch.Generator().manual_seed(77)
    d = ops.device
    w = torch.randn(N, K, generator=g) * 0.02
    w[:, :4] = torch.randn(N, 4, generator=g) * 40.0              # outlier input channels: ~90 % of each row's L1 norm
    a = torch.randn(M, K, generator=g)
    a[:, :4] *= 1e-3                                               # ... which this layer's inputs barely excite
    bias = torch.randn(N, generator=g) * 0.01
    par = split_bound_par(w, bias)
    asp, wsp = ops.split_f16(a.to(d)), ops.split_f16(w.to(d))
    so = torch.zeros(M, 2 * N, dtype=torch.float16, device=d)
    inv = torch.zeros(M, device=d)
    ops.debug_bounds, ops.bound_looseness_max = True, 1.0
    try:
        ops.gemm_x3_split(asp, wsp, bias.to(d), H.ACT_NONE, so, inv, par.to(d))
        loose = ops.bound_looseness_max
        assert 2.0 ** 6 < loose <= H.BOUND_LOOSENESS_LIMIT, loose    # far looser than the Gaussian case (2^4..2^7), still inside the limit
        want = ops.gemm_x3(asp, wsp, bias.to(d)).cpu().double()
        rec = (so.cpu()[:, :N].double() + so.cpu()[:, N:].double()) * inv.cpu().double()[:, None]
        assert ((rec - want).abs() <= 2.0 ** -20 * want.abs().amax(1, keepdim=True)).all()
        with pytest.raises(H.PsalmHipError, match="2\\^14"):
            ops.gemm_x3_split(asp, wsp, bias.to(d), H.ACT_NONE, so, inv, (par * torch.tensor([2.0 ** 12, 2.0 ** 12, 1.0, 1.0])).to(d))
    finally:
        ops.debug_bounds = False


@pytest.mark.parametrize("M,N,K,policy", [(300, 520, 192, 256), (300, 260, 128, 128), (200, 130, 704, 64), (100, 72, 64, 0), (270, 300, 704, 128)])
def test_gemm_x3_single_product_is_the_hi_hi_term(ops, M, N, K, policy):
    """psalm_gemm_x3_set_products(1) -- the reduced-precision LLM side mode of BASELINE.json configs[4]: the split-f16 GEMM stops after the hi.hi
    product, i.e. it is the plain f16 GEMM of the operands' hi halves under the same per-row scales (fp32 accumulate: exact to fp32 round-off
    against the float64 product of those halves), and a few 2^-11 of sum |a||w| away from the fp32-class three-product result.  The setting is
    per host thread and read at launch; 3 restores the default."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-4, 4, (M, 1), generator=g).float())
    w = torch.randn(N, K, generator=g) * 0.5
    bias = torch.randn(N, generator=g)
    d = ops.device
    asp, wsp = ops.split_f16(a.to(d)), ops.split_f16(w.to(d))
    ops.gemm_tile_policy(policy)
    try:
        full = ops.gemm(asp, wsp, bias.to(d)).cpu().double()
        ops.x3_products(1)
        got = ops.gemm(asp, wsp, bias.to(d)).cpu().double()
        ops.x3_products(3)
        again = ops.gemm(asp, wsp, bias.to(d)).cpu().double()
    finally:
        ops.x3_products(3)
        ops.gemm_tile_policy(0)
    hi_a = asp.t[:, :asp.Kp].cpu().double() * asp.inv_scale.cpu().double()[:, None]
    hi_w = wsp.t[:, :wsp.Kp].cpu().double() * wsp.inv_scale.cpu().double()[:, None]
    want = hi_a @ hi_w.t() + bias.double()
    mag = a.abs().double() @ w.abs().double().t()
    assert (got - want).abs().max() <= 2e-6 * mag.max()                       # the hi.hi product, fp32 accumulation
    exact = a.double() @ w.double().t() + bias.double()
    assert ((full - exact).abs() <= 6 * 2.0 ** -22 * mag + 1e-6).all()        # default: fp32 class
    err1 = ((got - exact).abs() / mag).max().item()
    assert 2.0 ** -16 < err1 < 2.0 ** -9                                      # one product: f16 operands (11-bit mantissas)
    assert torch.equal(again, full)                                           # the default is back
    with pytest.raises(Exception):
        ops.x3_products(2)


@pytest.mark.parametrize("M,N,K,policy", [(270, 300, 704, 128), (300, 520, 1536, 256), (130, 260, 3072, 64)])      # 4, 8, 16 K slices
def test_gemm_x3_split_k_xcd_placement_is_bitwise_the_plain_placement(ops, M, N, K, policy):
    """r06: in split-K launches whose slice count divides 8 (or is a multiple of it) the K slice, not the tile, decides a block's XCD
    (PSALM_TUNE_GEMM_XCD_KSPLIT): (tile, slice) = (lin / splits, lin % splits) of the linear block id.  A bijection of the grid -- slabs, the
    reduce and the output are the same words as with (blockIdx.x, blockIdx.y)."""
    g = torch.Generator().manual_seed(M + K)
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    d = ops.device
    outs, kernels = [], []
    ops.gemm_tile_policy(policy)
    try:
        assert ops.gemm_describe(M, N, 3 * K, x3=True)[3] in (4, 8, 16)
        for v in (0, 1):
            ops.set_tuning(ops.TUNE_GEMM_XCD_KSPLIT, v)
            outs.append(ops.gemm_x3(a.to(d), w.to(d), bias.to(d), res.to(d), H.ACT_NONE).cpu())
            kernels.append(ops.gemm_last_kernel())
    finally:
        ops.set_tuning(ops.TUNE_GEMM_XCD_KSPLIT, 1)
        ops.gemm_tile_policy(0)
    assert "splitk_reduce" in kernels[0] and kernels[0] == kernels[1], kernels
    assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))
    want = a.double() @ w.double().T + bias.double() + res.double()
    assert (outs[1].double() - want).abs().max() <= 2e-5 * want.abs().max()


MID_KERNELS = {6: "256, 128, 4, 2, 3, false, 32, 6, 2", 7: "128, 256, 2, 4, 3, false, 32, 6, 2", 8: "256, 128, 4, 2, 3, false, 32, 7, 2",
               10: "128, 128, 2, 2, 3, false, 32, 6, 2", 11: "64, 128, 2, 2, 3, false, 32, 6, 2"}


@pytest.mark.parametrize("form", [6, 7, 8, 10, 11])
@pytest.mark.parametrize("M,N,K", [(300, 520, 192), (515, 300, 64), (257, 260, 128)])
def test_gemm_x3_mid_forms_are_bitwise_the_r05_kernels(ops, form, M, N, K):
    """r06 mid-size forms of the split-f16 slice GEMM (policy 4400 + form): blocks with dedicated LOADER wavefronts -- the matrix waves issue no
    global -> LDS copy, the loaders nothing else; three stages.  Every output element is the same chain of matrix instructions over the same
    32-deep slices whatever the tile or who copies: the results equal the r05 kernels' (form 9) word for word -- fp32 output with bias / ReLU /
    residual, and split-f16 output (LDS-transposed and paired stores) with GELU.  On hardware this is also the hazard test of the three-stage
    hand-off (the emulator's copies are synchronous)."""
    g = torch.Generator().manual_seed(M + N + K + form)
    a = torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-4, 4, (M, 1), generator=g).float())
    w = torch.randn(N, K, generator=g) * 0.2
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    d = ops.device
    asp, wsp = ops.split_f16(a.to(d)), ops.split_f16(w.to(d))
    Ns = N // 64 * 64                                          # paired stores: a multiple of 64 columns
    Kp_out = (Ns + 127) // 128 * 128
    par = split_bound_par(w[:Ns], bias[:Ns]).to(d)
    perm = H.Ops.so_pair_perm(Ns)
    wq = ops.split_f16(w[:Ns][perm].to(d))
    outs = {}
    try:
        for f in (9, form):
            ops.gemm_tile_policy(4400 + f)
            c = ops.gemm_x3(asp, wsp, bias.to(d), res.to(d), H.ACT_RELU | H.ACT_POST_RESIDUAL).cpu()
            k0 = ops.gemm_last_kernel()
            so = torch.zeros(M, 2 * Kp_out, dtype=torch.float16, device=d)
            inv = torch.zeros(M, device=d)
            ops.gemm_x3_split(asp, ops.split_f16(w[:Ns].to(d)), bias[:Ns].to(d), H.ACT_GELU, so, inv, par, split_col_off=0, split_col_start=0, act_col_start=0)
            k1 = ops.gemm_last_kernel()
            so2 = torch.zeros(M, 2 * Kp_out, dtype=torch.float16, device=d)
            inv2 = torch.zeros(M, device=d)
            ops.gemm_x3_split(asp, wq, bias[:Ns][perm].to(d), H.ACT_GELU, so2, inv2, par, split_col_off=0, split_col_start=0, act_col_start=0, paired=True)
            k2 = ops.gemm_last_kernel()
            outs[f] = (c, so.cpu(), inv.cpu(), so2.cpu(), inv2.cpu(), (k0, k1, k2))
    finally:
        ops.gemm_tile_policy(4400)
    assert all(MID_KERNELS[form] in k for k in outs[form][5]), outs[form][5]
    assert not any(", 6, 2" in k or ", 7, 2" in k for k in outs[9][5]), outs[9][5]
    assert torch.equal(outs[9][0].view(torch.int32), outs[form][0].view(torch.int32))
    for i in (1, 3):
        assert torch.equal(outs[9][i].view(torch.int16), outs[form][i].view(torch.int16))
    for i in (2, 4):
        assert torch.equal(outs[9][i], outs[form][i])
    assert torch.equal(outs[form][1].view(torch.int16), outs[form][3].view(torch.int16))      # paired stores == LDS-transposed stores
    want = torch.relu(a.double() @ w.double().T + bias.double() + res.double())
    assert (outs[form][0].double() - want).abs().max() <= 3e-5 * want.abs().max()


def test_gemm_x3_mid_form_selection_rule():
    """select_mid_form through psalm_gemm_describe (host arithmetic only): the loader-wave blocks are chosen for long K loops (Kp >= 512) whose grid
    is one round of blocks filling most of the chip; everything else keeps its r05 tile.  The shapes are the f16x3 image's (profiles/r05_bench_breakdown.json)."""
    from ops_backend import make_ops
    ops = make_ops("emu")
    want = {(5184, 1536, 512): (256, 128), (4096, 2048, 512): (256, 128), (21504, 256, 1024): (256, 128), (16384, 256, 1024): (128, 128),
            (5184, 512, 512): (128, 128), (1296, 1024, 1024): (64, 128), (4096, 512, 1024): (64, 128), (1024, 4096, 1024): (128, 128),
            (1296, 3072, 1024): (64, 128),                        # 264 tiles of 128 x 128 would be two rounds: the 64 x 128 loader form (504 <= 512)
            (21504, 1024, 256): (64, 128), (21504, 256, 256): (64, 128), (65536, 512, 128): (64, 128), (16384, 1024, 256): (64, 128)}     # Kp <= 256: r05 tiles
    for (M, N, K), (bm, bn) in want.items():
        path, BM, BN, splits = ops.gemm_describe(M, N, 3 * K, x3=True)
        assert (path, BM, BN, splits) == (1, bm, bn, 1), ((M, N, K), (path, BM, BN, splits))
    try:
        ops.set_tuning(ops.TUNE_GEMM_MID, 0)                          # switch off: the r05 tiles everywhere
        assert ops.gemm_describe(5184, 1536, 3 * 512, x3=True)[1:3] == (128, 128)
        assert ops.gemm_describe(21504, 256, 3 * 1024, x3=True)[1:3] == (64, 128)
    finally:
        ops.set_tuning(ops.TUNE_GEMM_MID, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(5184, 512, 512), (1296, 1024, 1024), (21504, 256, 1024)])
def test_gemm_x3_mid_automatic_selection_is_bitwise_the_r05_kernel_on_gpu(M, N, K):
    """The automatic selection at three of the image's shapes (128 x 128, 64 x 128 and 256 x 128 loader-wave blocks) against the r05 kernel: equal words."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ops_backend import make_ops
    ops = make_ops("hip")
    g = torch.Generator().manual_seed(M + K)
    a, w = torch.randn(M, K, generator=g).cuda(), (torch.randn(N, K, generator=g) * 0.1).cuda()
    asp, wsp = ops.split_f16(a), ops.split_f16(w)
    outs, ks = [], []
    try:
        for pol in (4409, 4400):
            ops.gemm_tile_policy(pol)
            outs.append(ops.gemm_x3(asp, wsp).cpu())
            ks.append(ops.gemm_last_kernel())
    finally:
        ops.gemm_tile_policy(4400)
    assert ", 3, false, 32, 6, 2" in ks[1] or ", 3, false, 32, 7, 2" in ks[1], ks
    assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))


@pytest.mark.parametrize("form,M,N,K", [(12, 300, 260, 1024), (13, 300, 260, 1024), (12, 515, 130, 512)])
def test_gemm_x3_mid_forms_with_split_k(ops, form, M, N, K):
    """Loader-wave blocks with the K range split over the grid (fp32 slabs + reduce: bias / activation / residual in the reduce pass): the accuracy
    contract of every split-f16 GEMM form, and the launch really is the split one."""
    g = torch.Generator().manual_seed(M + N + K + form)
    a = torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-4, 4, (M, 1), generator=g).float())
    w = torch.randn(N, K, generator=g) * 0.2
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    d = ops.device
    try:
        ops.gemm_tile_policy(4400 + form)
        got = ops.gemm_x3(a.to(d), w.to(d), bias.to(d), res.to(d), H.ACT_RELU | H.ACT_POST_RESIDUAL).cpu().double()
        k = ops.gemm_last_kernel()
    finally:
        ops.gemm_tile_policy(4400)
    assert "splitk_reduce_kernel" in k and ("32, 7, 2" in k or "32, 6, 2" in k), k
    want = torch.relu(a.double() @ w.double().T + bias.double() + res.double())
    mag = a.abs().double() @ w.abs().double().T
    assert ((got - want).abs() <= 6 * 2.0 ** -22 * mag + 4e-7 * want.abs() + 1e-6).all()
