import sys, json, torch
sys.path.insert(0, '.')
from psalm_amd.hip_ops import get_ops
ops = get_ops()
for (M, N, K) in [(100, 65536, 256), (100, 16384, 256)]:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda")
    asp, wsp = ops.split_f16(a), ops.split_f16(w)
    c = torch.empty(M, N, device="cuda")
    for name, pol in [("auto", [0]), ("t128", [128]), ("t64", [64]), ("t256", [256]), ("auto2", [0])]:
        for p in pol: ops.gemm_tile_policy(p)
        for _ in range(3): ops.gemm_x3(asp, wsp, out=c)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): ops.gemm_x3(asp, wsp, out=c)
        e1.record(); torch.cuda.synchronize()
        print(M, N, K, name, round(e0.elapsed_time(e1) / 30 * 1e3, 1), ops.gemm_last_kernel())
        ops.gemm_tile_policy(0)
