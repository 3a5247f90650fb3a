"""Experiment (GPU box): the 256 x 256 loader-wave form (policy 4414) against the phased 256 x 256 loop on the Phi shapes and a square problem."""
import sys, torch
sys.path.insert(0, '.')
from psalm_amd.hip_ops import get_ops
ops = get_ops()
for (M, N, K) in [(899, 2048, 10240), (928, 2048, 10240), (899, 14336, 2048), (4096, 4096, 4096), (65536, 256, 2304)]:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda")
    asp, wsp = ops.split_f16(a), ops.split_f16(w)
    c = torch.empty(M, N, device="cuda")
    ref = None
    for name, pol in [("phased", [4400]), ("lw256", [4414]), ("phased2", [4400]), ("lw256_2", [4414])]:
        for p in pol: ops.gemm_tile_policy(p)
        for _ in range(3): ops.gemm_x3(asp, wsp, out=c)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.gemm_x3(asp, wsp, out=c)
        e1.record(); torch.cuda.synchronize()
        same = None
        if ref is None: ref = c.clone()
        else: same = bool(torch.equal(ref, c))
        print(M, N, K, name, round(e0.elapsed_time(e1) / 20 * 1e3, 1), "us", round(2.0 * M * N * K / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e12, 1), "TF", ops.gemm_last_kernel(), "same_words" if same else same, flush=True)
        ops.gemm_tile_policy(4400)
