"""Golden for the evaluator-facing outputs (VERDICT r02 missing #2): run the REFERENCE's own evaluator arithmetic on
seeded model-shaped outputs and store what it produces.  Authoring container only (needs /root/reference).

The reference modules cannot be imported here (detectron2 / panopticapi / pycocotools / a CUDA device are absent), so the functions are
taken from their source files as TEXT -- `ast` finds the definition, `exec` runs exactly those lines -- and called:

  psalm/eval/referring_segmentation.py   class AverageMeter (:38-79), intersectionAndUnionGPU (:101-113), compute_metric (:139-171)
  psalm/eval/segmentation_evaluation/panoptic_evaluation.py   my_SemSegEvaluator.process (:114-145: argmax + confusion-matrix update)
  psalm/eval/region_segmentation.py:286-288   the three lines that turn the meters into cIoU / gIoU (copied into `final_metrics` below
                                              verbatim, they are statements inside a 150-line main())

`.cuda()` is made the identity for the run (compute_metric moves its operands to the GPU; the arithmetic is device-independent integer
histogramming) and torch.histc gets its missing CPU integer path (via float64: exact for these counts).  What stays un-pinned: the PNG colour mapping (panopticapi.id2rgb) and the COCO RLE string (pycocotools) -- third-party
code that is not in the image.

    python tests/golden/make_evalout_golden.py      -> tests/golden/evalout.npz"""
import ast
import os
import sys
import types
from enum import Enum

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/psalm/eval"


def take(path, *names):
    """source text of the top-level definitions `names` (classes / functions) of a reference file"""
    src = open(path).read()
    tree = ast.parse(src)
    lines = src.splitlines()
    out = []
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            out.append("\n".join(lines[node.lineno - 1 - len(node.decorator_list):node.end_lineno]))
    assert len(out) == len(names), (path, names)
    return "\n\n".join(out)


def take_method(path, cls, meth):
    src = open(path).read()
    lines = src.splitlines()
    for node in ast.parse(src).body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for m in node.body:
                if isinstance(m, ast.FunctionDef) and m.name == meth:
                    body = lines[m.lineno - 1:m.end_lineno]
                    ind = len(body[0]) - len(body[0].lstrip())
                    return "\n".join(l[ind:] for l in body)
    raise KeyError((cls, meth))


def cases(seed=11):
    """model-shaped outputs: per sample Q candidate masks (uint8) + scores, a ground-truth mask with an ignore band; one (C,H,W) class map"""
    g = torch.Generator().manual_seed(seed)
    samples = []
    for i, (Q, H, W) in enumerate([(5, 37, 41), (3, 64, 48), (4, 20, 20), (2, 33, 7)]):
        pred = (torch.rand(Q, H, W, generator=g) < 0.3 + 0.1 * i).to(torch.uint8)
        gt = (torch.rand(H, W, generator=g) < 0.35).to(torch.uint8)
        gt[:2] = 255                                            # ignore rows
        if i == 2:
            gt[:] = 0                                           # a no-object target: union of the foreground class can be 0
            pred[int(torch.rand(Q, generator=g).argmax())] = 0
        scores = torch.rand(Q, generator=g)
        samples.append({"pred": pred.numpy(), "gt": gt.numpy(), "scores": scores.numpy()})
    C, H, W = 9, 45, 52
    sem = torch.randn(C, H, W, generator=g)
    sem_gt = torch.randint(0, C, (H, W), generator=g)
    sem_gt[torch.rand(H, W, generator=g) < 0.1] = 255
    return samples, sem.numpy(), sem_gt.numpy().astype(np.int64), C


def main():
    ns = {"torch": torch, "np": np, "Enum": Enum, "dist": types.SimpleNamespace()}
    exec("class Summary(Enum):\n    NONE = 0\n    AVERAGE = 1\n    SUM = 2\n    COUNT = 3\n", ns)
    exec(take(os.path.join(REF, "referring_segmentation.py"), "AverageMeter", "intersectionAndUnionGPU", "compute_metric"), ns)
    exec(take_method(os.path.join(REF, "segmentation_evaluation", "panoptic_evaluation.py"), "my_SemSegEvaluator", "process"), ns)
    samples, sem, sem_gt, C = cases()
    store = {}
    real_cuda, real_histc = torch.Tensor.cuda, torch.histc
    torch.Tensor.cuda = lambda self, *a, **k: self
    # torch.histc has no integer kernel on the CPU backend (the reference calls it on CUDA int tensors): same histogram through float64
    torch.histc = lambda x, bins=100, min=0, max=0: real_histc(x if x.is_floating_point() else x.double(), bins=bins, min=min, max=max)
    try:
        im, um, am = (ns["AverageMeter"](n, ":6.3f", ns["Summary"].SUM) for n in ("Intersec", "Union", "gIoU"))
        for i, smp in enumerate(samples):
            res = [{"pred": smp["pred"], "gt": smp["gt"], "scores": smp["scores"], "pred_cls": None}]
            # compute_metric's own signature has a gt_cls argument its callers (region_segmentation.py:280) do not pass: positional, as there
            preds, gts = ns["compute_metric"](im, um, am, None, res) if ns["compute_metric"].__code__.co_argcount == 5 else \
                ns["compute_metric"](im, um, am, res)
            top = int(np.argmax(smp["scores"]))
            inter, union, tgt = ns["intersectionAndUnionGPU"](torch.tensor(smp["pred"][top]).int().clone(), torch.tensor(smp["gt"]).int(), 2, ignore_index=255)
            store[f"s{i}/pred"], store[f"s{i}/gt"], store[f"s{i}/scores"] = smp["pred"], smp["gt"], smp["scores"]
            store[f"s{i}/intersection"], store[f"s{i}/union"], store[f"s{i}/target"] = inter.numpy(), union.numpy(), tgt.numpy()
            store[f"s{i}/kept_pred"] = np.asarray(preds[0])
        # region_segmentation.py:286-288, verbatim
        intersection_meter, union_meter, acc_iou_meter = im, um, am
        iou_class = intersection_meter.sum / (union_meter.sum + 1e-10)
        ciou = iou_class[1]
        giou = acc_iou_meter.avg[1]
        store["meters/intersection_sum"], store["meters/union_sum"] = np.asarray(im.sum, np.float64), np.asarray(um.sum, np.float64)
        store["meters/acc_iou_sum"], store["meters/count"] = np.asarray(am.sum, np.float64), np.asarray(am.count)
        store["meters/ciou"], store["meters/giou"] = np.asarray(ciou, np.float64), np.asarray(giou, np.float64)
        # my_SemSegEvaluator.process on a stand-in `self` holding exactly the attributes the method reads
        fake = types.SimpleNamespace(_cpu_device=torch.device("cpu"), sem_seg_loading_fn=lambda fn, dtype=int: sem_gt.astype(dtype).copy(),
                                     _ignore_label=255, _num_classes=C, _conf_matrix=np.zeros((C + 1, C + 1), dtype=np.int64),
                                     _compute_boundary_iou=False, _b_conf_matrix=np.zeros((C + 1, C + 1), dtype=np.int64),
                                     _predictions=[], encode_json_sem_seg=lambda *a, **k: [])
        ns["process"](fake, [{"sem_seg_file_name": "x.png", "file_name": "x.jpg"}], [{"sem_seg": torch.from_numpy(sem)}])
        store["sem/logits"], store["sem/gt"], store["sem/num_classes"] = sem, sem_gt, np.asarray(C)
        store["sem/conf_matrix"] = fake._conf_matrix
    finally:
        torch.Tensor.cuda, torch.histc = real_cuda, real_histc
    np.savez_compressed(os.path.join(HERE, "evalout.npz"), **store)
    print("wrote", os.path.join(HERE, "evalout.npz"), len(store), "arrays;  cIoU", float(ciou), "gIoU", float(giou))


if __name__ == "__main__":
    main()
