"""Experiment (CPU only, not product): the two referring inputs r05's wide run left outside the flip margin -- referring 640^2 batch 4, inputs seed 10
image 2 and seed 11 image 1 -- under CONTROLS of the oracle itself (VERDICT r05 "Next" #3a).  Each control is the CPU oracle
(oracle/psalm_oracle.py, fp32) with one class of its arithmetic evaluated at least as exactly as the reference does, or merely in another order:

    threads1   one host thread (the BLAS blocks and sums the K range differently)
    padL       the LLM sequence padded to the next multiple of 32 with zero-embedding, masked positions -- what PSALM._bucketed does to every
               batch (model.py) and what the reference itself does to the shorter prompts of a ragged batch (llava_phi.py:939-946)
    all64      every nn.Linear in float64, rounded once
    ln64       every LayerNorm in float64, rounded once
    attn64     the three attention forms (Swin windows, Phi causal, mask-decoder multi-head): scores, softmax and value products in float64
    full64     all64 + ln64 + attn64: the float64 control of the whole transformer arithmetic
    threadsN   N host threads (N = 2, 4, ...)
    conv64 / gn64 / interp64   F.conv2d / F.group_norm / F.interpolate (the thresholded attention-mask resize of TD:754-760 among them) in float64
    msda_grid  the oracle's second restatement of the deformable-attention gather (F.grid_sample per level instead of explicit corner taps)
    every64    full64 + conv64 + gn64 + interp64
    splitkN    Phi's residual projections  dense(attn) + fc2(gelu(fc1))  formed as the product forms them: ONE product over the concatenated K range
               (2048 + 8192) summed in N slices of ceil(K / 64 / N) * 64 columns, fp32 -- the split-K order of the fused [dense | fc2] GEMM (N = 8 at
               these sizes), another fp32 summation order of the same numbers
-> one JSON line per (seed, image, variant): flipped pixels, the oracle's |logit| at the flips relative to the logit range (oracle/parity_gate.py's
margin property), mask IoU.  An input that tips under these controls is one on which the REFERENCE's fp32 result is itself within rounding of a
decision of the thresholded attention-mask feedback (mask2former_transformer_decoder.py:754-760); one that does not tip under any of them while the
product moves it would point at the product.

    python tools/exp_referring_controls.py [seeds=10,11] [variants=threads1,padL,all64,ln64,attn64,full64] [--save-control full64]"""
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import parity_gate as PG  # noqa: E402
from oracle import psalm_oracle as O  # noqa: E402
from psalm_amd.config import PsalmConfig  # noqa: E402
from psalm_amd.synthetic import make_inputs, make_state_dict  # noqa: E402


def compare(g, w):
    gm, wm = g["mask_pred"] > 0, w["mask_pred"] > 0
    inter = (gm & wm).flatten(1).sum(1).float()
    union = (gm | wm).flatten(1).sum(1).float()
    iou = torch.where(union > 0, inter / union.clamp(min=1), torch.ones_like(union))
    flips = gm != wm
    rng = float(w["mask_pred"].abs().max())
    margin = float(w["mask_pred"][flips].abs().max()) / rng if flips.any() else 0.0
    return {"flipped_pixels": int(flips.sum()), "flip_margin_rel_max": float(f"{margin:.3e}"), "flips_within_margin": margin <= PG.FLIP_MARGIN_REL,
            "mask_logit_rel_err": float(f"{((g['mask_pred'] - w['mask_pred']).abs().max() / rng):.3e}"),
            "mask_iou_mean": round(float(iou.mean()), 6), "mask_iou_pooled": round(float(inter.sum() / union.sum().clamp(min=1)), 6)}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    seeds = [int(s) for s in (args[0] if args else "10,11").split(",")]
    variants = (args[1] if len(args) > 1 else "threads1,padL,all64,ln64,attn64,full64").split(",")
    save = sys.argv[sys.argv.index("--save-control") + 1] if "--save-control" in sys.argv else None
    size, batch, task = 640, 4, "referring"
    nthr = min(os.cpu_count() or 1, 64)
    cfg = PsalmConfig(seg_task=task)
    sd = make_state_dict(cfg, seed=0)
    real_lin, real_ln, real_phi = O._lin, O._ln, O.phi_forward

    def lin64(sd_, name, x, bias=True):
        b = sd_[name + ".bias"] if bias and (name + ".bias") in sd_ else None
        return F.linear(x.double(), sd_[name + ".weight"].double(), None if b is None else b.double()).float()

    def ln64(sd_, name, x, eps=1e-5):
        return F.layer_norm(x.double(), (x.shape[-1],), sd_[name + ".weight"].double(), sd_[name + ".bias"].double(), eps).float()

    def phi_pad(sd_, cfg_, emb, am, prefix="model."):
        B, L, H = emb.shape
        Lp = (L + 31) // 32 * 32
        if Lp == L:
            Lp += 32
        e2 = torch.zeros(B, Lp, H)
        e2[:, :L] = emb
        m2 = torch.zeros(B, Lp, dtype=am.dtype)
        m2[:, :L] = am
        return real_phi(sd_, cfg_, e2, m2, prefix)[:, :L]

    real_F = O.F

    class FProxy:
        """torch.nn.functional with some entries evaluated in float64 (the oracle reaches them as `F.<name>`)"""
        def __init__(self, names):
            self._names = set(names)

        def __getattr__(self, name):
            fn = getattr(real_F, name)
            if name not in self._names:
                return fn

            def f64(x, *a, **k):
                a = [t.double() if torch.is_tensor(t) and t.is_floating_point() else t for t in a]
                k = {kk: (t.double() if torch.is_tensor(t) and t.is_floating_point() else t) for kk, t in k.items()}
                return fn(x.double(), *a, **k).float()
            return f64

    def lin_splitk(nsl):
        stash = {}

        def lin(sd_, name, x, bias=True):
            if name.startswith("model.layers.") and name.endswith("self_attn.dense"):
                stash["a"] = (name, x)
                return torch.zeros(x.shape[:-1] + (sd_[name + ".weight"].shape[0],))
            if name.startswith("model.layers.") and name.endswith("mlp.fc2"):
                na, xa = stash.pop("a")
                X = torch.cat([xa, x], -1)
                W = torch.cat([sd_[na + ".weight"], sd_[name + ".weight"]], 1)
                K = X.shape[-1]
                ks = -(-(-(-K // 64)) // nsl) * 64
                out = None
                for k0 in range(0, K, ks):
                    part = X[..., k0:k0 + ks] @ W[:, k0:k0 + ks].t()
                    out = part if out is None else out + part
                return out + (sd_[na + ".bias"] + sd_[name + ".bias"])
            return real_lin(sd_, name, x, bias)
        return lin

    def setup(var):
        O._lin, O._ln, O.phi_forward, O.ATTN_FLOAT64, O.F = real_lin, real_ln, real_phi, False, real_F
        if var.startswith("splitk"):
            O._lin = lin_splitk(int(var[6:]))
        torch.set_num_threads(int(var[7:]) if var.startswith("threads") else nthr)
        if var in ("all64", "full64", "every64"):
            O._lin = lin64
        if var in ("ln64", "full64", "every64"):
            O._ln = ln64
        if var in ("attn64", "full64", "every64"):
            O.ATTN_FLOAT64 = True
        if var == "padL":
            O.phi_forward = phi_pad
        f64names = {"conv64": ["conv2d"], "gn64": ["group_norm"], "interp64": ["interpolate"], "every64": ["conv2d", "group_norm", "interpolate"]}.get(var)
        if f64names:
            O.F = FProxy(f64names)

    for seed in seeds:
        inputs = make_inputs(cfg, task, size=size, batch=batch, seed=seed)
        setup("none")
        torch.manual_seed(1234)
        t0 = time.perf_counter()
        want = O.eval_seg(sd, cfg, **inputs)
        secs = time.perf_counter() - t0
        for var in variants:
            setup(var)
            try:
                torch.manual_seed(1234)
                t1 = time.perf_counter()
                got = O.eval_seg(sd, cfg, **inputs, **({"msda_fn": (lambda v_, sh_, st_, loc_, w_: O.msda_core_grid_sample(v_, sh_, loc_, w_))} if var == "msda_grid" else {}))
                vsecs = time.perf_counter() - t1
            finally:
                setup("none")
            for b in range(len(got)):
                print(json.dumps({"task": task, "size": size, "batch": batch, "inputs_seed": seed, "image": b, "variant": var, "against": "oracle_fp32",
                                  "oracle_threads": nthr, **compare(got[b], want[b]), "oracle_seconds": round(secs, 1), "variant_seconds": round(vsecs, 1)}),
                      flush=True)
            if save == var:
                import numpy as np
                os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
                path = os.path.join(ROOT, "tests", "golden", f"referring_640_seed{seed}_{var}_control.npz")
                np.savez_compressed(path, **{f"mask_sign_{b}": np.packbits((got[b]["mask_pred"] > 0).numpy()) for b in range(len(got))},
                                    **{f"shape_{b}": np.array(got[b]["mask_pred"].shape) for b in range(len(got))},
                                    **{f"oracle_sign_{b}": np.packbits((want[b]["mask_pred"] > 0).numpy()) for b in range(len(got))})
                print(json.dumps({"saved": path}), flush=True)


if __name__ == "__main__":
    main()
