"""Micro-benchmark of the MSDA gather at the real pixel-decoder shape (1024^2: S=Lq=21504, M=8, D=32, L=3, P=4).
Reports time per launch (HIP events on the launch stream) and algorithmic GB/s
(value + locations + weights + output bytes)."""
import json
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psalm_amd.hip_ops import get_ops  # noqa: E402


def main():
    ops = get_ops()
    B, M, D, P = 1, 8, 32, 4
    shapes = [(32, 32), (64, 64), (128, 128)]
    L = 3
    S = sum(h * w for h, w in shapes)
    starts = [0, 1024, 5120]
    res = {}
    for dt in (torch.float32, torch.bfloat16):
        value = torch.randn(B, S, M * D, device="cuda").to(dt)
        ow = torch.randn(B, S, M * L * P * 3, device="cuda")
        loc = torch.rand(B, S, M, L, P, 2, device="cuda")
        aw = torch.softmax(torch.randn(B, S, M, L * P, device="cuda"), -1).view(B, S, M, L, P)
        es = value.element_size()
        for name, fn, bytes_ in (
            ("explicit", lambda: ops.msda_forward(value.view(B, S, M, D), shapes, starts, loc, aw),
             S * 256 * es + S * 96 * 2 * 4 + S * 96 * 4 + S * 256 * es),
            ("fused", lambda: ops.msda_fused(value, shapes, starts, ow, M), S * 256 * es + S * 288 * 4 + S * 256 * es),
        ):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 50
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / n * 1e3
            res[f"{name}_{'f32' if dt == torch.float32 else 'bf16'}"] = {"us": round(us, 2), "alg_MB": round(bytes_ / 1e6, 2),
                                                                         "GBps": round(bytes_ / us / 1e3, 1)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
