"""Experiment (GPU box): WHERE does the product's result for referring 640^2 batch 4, inputs seed 10, image 2 leave the oracle's (58 pixels / 2.2e-3 at full
resolution)?  Stage outputs of the product are substituted into the ORACLE's own remaining stages (CPU, fp32), and the oracle's stage outputs into the
product's predictor: each hybrid's low-resolution mask logits (Q, h, w) are compared with the oracle's.  A hybrid that lands on the product's side names
the stage whose fp32-class rounding tips this input; none of the stages contains length-dependent (ragged / bucketed) code except the LLM.
    python tools/experiments/r06_seed10_stage_bisect.py [seed=10] [image=2]"""
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
IMG = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = PsalmConfig(seg_task="referring")
sd = make_state_dict(cfg, seed=0)
inputs = make_inputs(cfg, "referring", size=640, batch=4, seed=seed)
torch.set_num_threads(min(16, os.cpu_count() or 8))
_, st = O.eval_seg(sd, cfg, **inputs, return_stages=True, postprocess=False)
want = st["pred_masks"]                                          # (B, Q, h, w)
m = PSALM(cfg, sd, precision="f16x3", use_graphs=False)
gs = {}
kw = {k: v for k, v in inputs.items()}
outs = m.forward_logits(stages=gs, **kw)
torch.cuda.synchronize()
B, Q = want.shape[0], want.shape[1]


def cmp(name, a, b=None):
    b = want if b is None else b
    for i in range(B):
        fl = (a[i] > 0) != (b[i] > 0)
        rng = float(b[i].abs().max())
        print(json.dumps({"seed": seed, "image": i, "hybrid": name, "flipped_lowres_pixels": int(fl.sum()), "rel_err": float(f"{float((a[i] - b[i]).abs().max()) / rng:.3e}"),
                          "margin": float(f"{(float(b[i][fl].abs().max()) / rng if fl.any() else 0.0):.3e}")}), flush=True)


gpu_masks = torch.stack([o_["pred_masks"].cpu() for o_ in outs])
cmp("product (all stages on the GPU)", gpu_masks)
# what the stage dictionaries hold
o_ms, o_mf = st["multi_scale_features"], st["mask_features"]     # levels of (B, D, h, w); (B, MD, H2, W2)
g_seg = gs["seg_query"].float().cpu()
g_SEG = gs["SEG_embedding"].float().cpu().view(B, 1, -1)
print(json.dumps({"stage_rel_err": {"seg_query": float((g_seg - st["seg_query"]).abs().max() / st["seg_query"].abs().max()),
                                    "SEG_embedding": float((g_SEG - st["SEG_embedding"]).abs().max() / st["SEG_embedding"].abs().max())}}), flush=True)
# (1) oracle predictor <- product's LLM-side embeddings, oracle's pixel decoder outputs
po = O.predictor_forward(sd, cfg, o_ms, o_mf, g_seg, g_SEG)
cmp("oracle predictor <- product seg_query / SEG embedding", po["pred_masks"])
# (2) oracle predictor <- product's pixel decoder outputs, oracle's embeddings
H2, W2 = o_mf.shape[-2:]
g_mf = torch.stack([t.float().cpu().t().reshape(-1, H2, W2) for t in gs["mask_features"]])
g_ms = []
for lv in range(len(o_ms)):
    h_, w_ = o_ms[lv].shape[-2:]
    cand = [torch.stack([next(t for t in gs["multi_scale_features"][b] if t.shape[0] == h_ * w_).float().cpu().t().reshape(-1, h_, w_) for b in range(B)])]
    g_ms.append(cand[0])
print(json.dumps({"stage_rel_err": {"mask_features": float((g_mf - o_mf).abs().max() / o_mf.abs().max()),
                                    **{f"multi_scale_{lv}": float((g_ms[lv] - o_ms[lv]).abs().max() / o_ms[lv].abs().max()) for lv in range(len(o_ms))}}}), flush=True)
po = O.predictor_forward(sd, cfg, g_ms, g_mf, st["seg_query"], st["SEG_embedding"])
cmp("oracle predictor <- product pixel-decoder outputs", po["pred_masks"])
po = O.predictor_forward(sd, cfg, g_ms, g_mf, g_seg, g_SEG)
cmp("oracle predictor <- ALL product inputs", po["pred_masks"])
# (3) product predictor <- ALL oracle inputs
res = []
for b in range(B):
    ms = [o_ms[lv][b].flatten(1).t().contiguous().cuda() for lv in range(len(o_ms))]
    shapes = [tuple(o_ms[lv].shape[-2:]) for lv in range(len(o_ms))]
    mf = o_mf[b].flatten(1).t().contiguous().cuda()
    r = m.predictor(ms, shapes, mf, (H2, W2), st["seg_query"][b].contiguous().cuda(), st["SEG_embedding"][b].contiguous().cuda(), None, None)
    res.append(r["pred_masks"].float().cpu().view(Q, H2, W2))
torch.cuda.synchronize()
cmp("product predictor <- ALL oracle inputs", torch.stack(res))
