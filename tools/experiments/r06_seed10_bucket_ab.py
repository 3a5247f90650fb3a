"""Experiment (GPU box): referring 640^2 batch 4, inputs seed 10 (image 2: 58 pixels / 2.2e-3 from the oracle since r05) -- the product with the LLM sequence
length bucketed to 32 (default) against the SAME product un-bucketed (len_bucket = 0: the split-K geometry of r04, which had this image at 1.6e-6), both
against the CPU oracle and against each other.  The bucketing itself is exact on the CPU oracle (tools/exp_referring_controls.py `padL`: 0 flipped
pixels, rel. error 0.0): whatever separates the two GPU runs is summation order.   python tools/experiments/r06_seed10_bucket_ab.py [seed=10]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import psalm_oracle as O  # noqa: E402
from psalm_amd.config import PsalmConfig  # noqa: E402
from psalm_amd.model import PSALM  # noqa: E402
from psalm_amd.synthetic import make_inputs, make_state_dict  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cfg = PsalmConfig(seg_task="referring")
sd = make_state_dict(cfg, seed=0)
inputs = make_inputs(cfg, "referring", size=640, batch=4, seed=seed)
torch.set_num_threads(min(16, os.cpu_count() or 8))
want = O.eval_seg(sd, cfg, **inputs)
m = PSALM(cfg, sd, precision="f16x3", use_graphs=False)
runs = {}
for name, q, tune in (("bucket32", 32, {}), ("bucket0", 0, {}), ("bucket64", 64, {}), ("bucket32_no_xcd_ksplit", 32, {0: 0})):
    m.len_bucket = q
    for k, v in tune.items():
        m.ops.set_tuning(k, v)
    try:
        got = m.eval_seg(**inputs)
        torch.cuda.synchronize()
    finally:
        for k in tune:
            m.ops.set_tuning(k, 1)
    runs[name] = [g["mask_pred"].cpu() for g in got]


def cmp(a, b):
    fl = (a > 0) != (b > 0)
    rng = float(b.abs().max())
    return {"flipped_pixels": int(fl.sum()), "rel_err": float(f"{float((a - b).abs().max()) / rng:.3e}"),
            "margin": float(f"{(float(b[fl].abs().max()) / rng if fl.any() else 0.0):.3e}")}


for name, r in runs.items():
    for b in range(4):
        print(json.dumps({"seed": seed, "image": b, "run": name, "against": "oracle", **cmp(r[b], want[b]["mask_pred"])}), flush=True)
for b in range(4):
    print(json.dumps({"seed": seed, "image": b, "run": "bucket32", "against": "bucket0", **cmp(runs["bucket32"][b], runs["bucket0"][b])}), flush=True)
    fa = ((runs["bucket32"][b] > 0) != (want[b]["mask_pred"] > 0))
    fb = ((runs["bucket0"][b] > 0) != (want[b]["mask_pred"] > 0))
    print(json.dumps({"seed": seed, "image": b, "flip_sets": {"bucket32_only": int((fa & ~fb).sum()), "bucket0_only": int((fb & ~fa).sum()), "both": int((fa & fb).sum())}}), flush=True)
