"""Golden for the gRefCOCO metric fusion (VERDICT r04 missing #5): run the REFERENCE's own `compute_metric` / `fuse_masks` of
psalm/eval/eval_grefcoco.py (:113-160, :277-285; AverageMeter :24-79, intersectionAndUnionGPU :66-79) on seeded candidates and store what
it keeps and counts.  Authoring container only (needs /root/reference).  Same technique as make_evalout_golden.py: the definitions are taken
from the source file as text and exec'd; `.cuda()` is the identity for the run and torch.histc gets its CPU integer path.

    python tests/golden/make_grefcoco_golden.py      -> tests/golden/grefcoco.npz"""
import os
import sys
import types
from enum import Enum

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_evalout_golden import take  # noqa: E402

REF = "/root/reference/psalm/eval/eval_grefcoco.py"


def cases(seed=23):
    """(candidate masks, scores, gt): several above the threshold / exactly one / none (-> top-1 fall-back) / a no-object target / ties"""
    g = torch.Generator().manual_seed(seed)
    out = []
    for i, (Q, H, W, hi) in enumerate([(6, 37, 41, 3), (5, 64, 48, 1), (4, 20, 20, 0), (7, 33, 7, 2), (3, 16, 24, 0)]):
        pred = (torch.rand(Q, H, W, generator=g) < 0.15 + 0.05 * i).to(torch.uint8)
        gt = (torch.rand(H, W, generator=g) < 0.3).to(torch.uint8)
        gt[:2] = 255
        scores = torch.rand(Q, generator=g) * 0.55                       # all below 0.6 ...
        scores[torch.randperm(Q, generator=g)[:hi]] += 0.45              # ... except `hi` of them
        if i == 3:
            gt[gt == 1] = 0                                              # no-object target
        if i == 4:
            scores[:] = 0.25                                             # none above thr AND a three-way tie: topk takes the first
        out.append({"pred": pred.numpy(), "gt": gt.numpy(), "scores": scores.numpy().astype(np.float32)})
    return out


def main():
    ns = {"torch": torch, "np": np, "Enum": Enum, "dist": types.SimpleNamespace()}
    exec("class Summary(Enum):\n    NONE = 0\n    AVERAGE = 1\n    SUM = 2\n    COUNT = 3\n", ns)
    exec(take(REF, "AverageMeter", "intersectionAndUnionGPU", "compute_metric", "fuse_masks"), ns)
    store = {}
    real_cuda, real_histc = torch.Tensor.cuda, torch.histc
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.histc = lambda x, bins=100, min=0, max=0: real_histc(x if x.is_floating_point() else x.double(), bins=bins, min=min, max=max)
    try:
        im, um, am = (ns["AverageMeter"](n, ":6.3f", ns["Summary"].SUM) for n in ("Intersec", "Union", "gIoU"))
        for i, smp in enumerate(cases()):
            res = [{"pred": smp["pred"], "gt": smp["gt"], "scores": smp["scores"], "pred_cls": None}]
            i0, u0 = np.array(im.sum, np.float64, copy=True), np.array(um.sum, np.float64, copy=True)
            preds, gts = ns["compute_metric"](im, um, am, None, res, thr=0.6)
            store[f"s{i}/pred"], store[f"s{i}/gt"], store[f"s{i}/scores"] = smp["pred"], smp["gt"], smp["scores"]
            store[f"s{i}/fused"] = np.asarray(preds[0]).astype(np.uint8)
            store[f"s{i}/intersection"] = np.asarray(im.sum, np.float64) - i0
            store[f"s{i}/union"] = np.asarray(um.sum, np.float64) - u0
        store["meters/intersection_sum"], store["meters/union_sum"] = np.asarray(im.sum, np.float64), np.asarray(um.sum, np.float64)
        store["meters/acc_iou_sum"], store["meters/count"] = np.asarray(am.sum, np.float64), np.asarray(am.count)
    finally:
        torch.Tensor.cuda, torch.histc = real_cuda, real_histc
    np.savez_compressed(os.path.join(HERE, "grefcoco.npz"), **store)
    print("wrote grefcoco.npz", len(store), "arrays; I", im.sum, "U", um.sum, "acc", am.sum, am.count)


if __name__ == "__main__":
    main()
