"""Runs a few launches of the 256x256 GEMM with the baseline K loop and with the PH8 K loop (for rocprofv3 --pmc passes; the two
instantiations differ in the kernel's last template argument).   python tools/prof_gemm_ph8.py [M N K]"""
import sys

import torch

sys.path.insert(0, ".")
from psalm_amd.hip_ops import get_ops  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (4096, 4096, 4096)
ops = get_ops()
a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
w = (torch.randn(N, K, device="cuda") * 0.5).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for ph8 in (0, 2, 3):
    ops.gemm_tile_policy(256)
    ops.gemm_tile_policy(2567 + ph8 if ph8 else 2560)
    for _ in range(8):
        ops.gemm(a, w, None, out=out)
    torch.cuda.synchronize()
ops.gemm_tile_policy(2570)
ops.gemm_tile_policy(0)
