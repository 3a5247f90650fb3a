"""Experiment (CPU only, not product): WHICH projection of the Phi decoder does not tolerate e4m3 cross terms?

DESIGN.md §0 item 2b: with the [k|v|q|fc1] GEMM in the x8 operand form one referring input (640x640 batch 4, inputs seed 4, image 0) left
the parity bar; with only [dense|fc2] in that form it did not.  The GPU budget of the round ended there.  This script restates the x8
arithmetic in torch on the CPU ORACLE (oracle/psalm_oracle.py, fp32) for a chosen subset of the Phi projections -- the operand split of
psalm_split_f16 (per-row power-of-two scale, row maximum in [2^13, 2^14), hi = f16(x s), lo = f16(x s - hi)), the e4m3 images
e(hi 2^-6), e(lo 2^6) (round to nearest even, 3 mantissa bits, subnormal step 2^-9, saturating at 448) and

    y = s_a^-1 s_w^-1 ( hi_a.hi_w + e(lo_a 2^6).e(hi_w 2^-6) + e(hi_a 2^-6).e(lo_w 2^6) ) + bias        (fp32 accumulation: torch matmul)

and compares the model's outputs with the un-touched oracle on the same input.  Every other GEMM of the model stays exact fp32 (the product
runs them at 2^-22), so the numbers isolate the projection's own effect.

    python tools/exp_x8_cpu.py [task=referring] [size=640] [batch=4] [seeds=4] [subsets=qkv+fc1,qkv,fc1,q,k,v,dense+fc2]
-> one JSON line per (inputs seed, subset, image)."""
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import psalm_oracle as O  # noqa: E402
from psalm_amd.config import PsalmConfig  # noqa: E402
from psalm_amd.synthetic import make_inputs, make_state_dict  # noqa: E402

NAMES = {"q": "self_attn.q_proj", "k": "self_attn.k_proj", "v": "self_attn.v_proj", "fc1": "mlp.fc1", "dense": "self_attn.dense", "fc2": "mlp.fc2"}


def split(x):
    amax = x.abs().amax(1, keepdim=True)
    _, ex = torch.frexp(amax)
    s = torch.where(amax > 0, torch.exp2((14 - ex).float()), torch.ones_like(amax))
    xs = x * s
    hi = xs.half().float()
    lo = (xs - hi).half().float()
    return hi, lo, 1.0 / s


def e4m3(x):
    v = x.clamp(-448.0, 448.0)
    a = v.abs()
    _, e2 = torch.frexp(a)
    e = (e2 - 1).clamp_min(-6).float()
    step = torch.exp2(e - 3)
    return torch.sign(v) * (torch.round(a / step) * step).clamp_max(448.0)      # torch.round: half to even


def lin_x8(x, w, b, three=False):
    shp = x.shape
    x2 = x.reshape(-1, shp[-1]).float()
    ha, la, ia = split(x2)
    hw, lw, iw = split(w.float())
    if three:
        y = ha @ hw.t() + la @ hw.t() + ha @ lw.t()
    else:
        y = ha @ hw.t() + e4m3(la * 64.0) @ e4m3(hw * 0.015625).t() + e4m3(ha * 0.015625) @ e4m3(lw * 64.0).t()
    y = y * ia * iw.t()
    if b is not None:
        y = y + b
    return y.reshape(*shp[:-1], w.shape[0])


def compare(g, w_):
    gm, wm = g["mask_pred"] > 0, w_["mask_pred"] > 0
    inter = (gm & wm).flatten(1).sum(1).float()
    union = (gm | wm).flatten(1).sum(1).float()
    iou = torch.where(union > 0, inter / union.clamp(min=1), torch.ones_like(union))
    return {"mask_iou_mean": round(float(iou.mean()), 6), "mask_iou_pooled": round(float(inter.sum() / union.sum().clamp(min=1)), 6),
            "flipped_pixels": int((gm != wm).sum()),
            "mask_logit_rel_err": float(f"{((g['mask_pred'] - w_['mask_pred']).abs().max() / w_['mask_pred'].abs().max()).item():.3e}")}


def main():
    task = sys.argv[1] if len(sys.argv) > 1 else "referring"
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 640
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    seeds = [int(s) for s in (sys.argv[4] if len(sys.argv) > 4 else "4").split(",")]
    subsets = (sys.argv[5] if len(sys.argv) > 5 else "qkv+fc1,qkv,fc1,q,k,v,dense+fc2").split(",")
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    cfg = PsalmConfig(seg_task=task)
    sd = make_state_dict(cfg, seed=0)
    real_lin = O._lin
    for seed in seeds:
        inputs = make_inputs(cfg, task, size=size, batch=batch, seed=seed)
        torch.manual_seed(1234)
        t0 = time.perf_counter()
        want = O.eval_seg(sd, cfg, **inputs)
        secs = time.perf_counter() - t0
        for sub in subsets:
            three = sub.endswith("@3p")                       # sanity line: the same projections in three f16 products
            keys = sub.replace("@3p", "").replace("qkv", "q+k+v").split("+")
            tails = tuple(NAMES[k] for k in keys)

            def lin(sd_, name, x, bias=True):
                if name.startswith("model.layers.") and name.endswith(tails):
                    return lin_x8(x, sd_[name + ".weight"], sd_[name + ".bias"] if bias and (name + ".bias") in sd_ else None, three)
                return real_lin(sd_, name, x, bias)
            O._lin = lin
            try:
                torch.manual_seed(1234)
                got = O.eval_seg(sd, cfg, **inputs)
            finally:
                O._lin = real_lin
            for b in range(len(got)):
                print(json.dumps({"task": task, "size": size, "inputs_seed": seed, "image": b, "x8_projections": sub, **compare(got[b], want[b]),
                                  "oracle_seconds": round(secs, 1)}), flush=True)


if __name__ == "__main__":
    main()
