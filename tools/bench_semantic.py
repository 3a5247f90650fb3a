"""Fused semantic pass (psalm_semantic_from_masks_x3) at the bench shape: Q = 100 mask logits x 1024^2 pixels -> 133 class planes.
    python tools/bench_semantic.py            -> one JSON line (warm / cold Infinity Cache), two runs in fresh processes
(r03 compared two tile orders through an environment switch the library no longer reads: profiles/r03n_semantic_tile_order.jsonl keeps that A/B.)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    import torch
    from psalm_amd.hip_ops import get_ops
    ops = get_ops()
    Q, C, HW = 100, 133, 1024 * 1024
    g = torch.Generator().manual_seed(0)
    mask = (torch.randn(Q, HW, generator=g) * 4 - 6).cuda()
    cls = torch.randn(Q, C + 1, generator=g).cuda()
    _, probsT, _, _ = ops.class_softmax(cls, 128, probsT_dtype=torch.float32)
    big = torch.empty(96 << 20, device="cuda")
    res = {}
    for mode in ("warm", "cold"):
        ts = []
        for _ in range(6):
            if mode == "cold":
                big.fill_(1.0)                               # 384 MB through the Infinity Cache
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out, ms = ops.semantic_from_masks(mask, probsT, want_mask_score=True)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
            del out, ms
        us = sorted(ts)[len(ts) // 2]
        res[mode] = {"us": round(us, 1), "TB_s": round((Q + C) * HW * 4 / us / 1e6, 3)}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    if "--one" in sys.argv:
        one()
    else:
        for _ in range(2):
            subprocess.call([sys.executable, os.path.abspath(__file__), "--one"])
