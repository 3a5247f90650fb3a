"""Evaluator-facing outputs computed on the device.

The reference's evaluators start every image with `.cpu().numpy()` of the model's full-resolution outputs -- `sem_seg` (133,H,W) fp32 =
558 MB and `instances.pred_masks` (100,H,W) fp32 = 419 MB at 1024^2 -- and then reduce them on the host to a label map, a confusion
matrix, a PNG, run-length codes or a few intersection / union counts.  These functions produce those SMALL results on the GPU (kernels in
csrc/evalout.hip) in exactly the evaluators' arithmetic, so only KBs..MBs cross PCIe and the metric meters can be combined across ranks
with one tiny RCCL all-reduce (`psalm_amd.dist.reduce_metrics`, the role of AverageMeter.all_reduce, referring_segmentation.py:58-79).

    semantic_labels(sem_seg)                      panoptic_evaluation.py:125        -> (H,W) int32 labels
    ConfusionMatrix(C, ignore).update(pred, gt)   panoptic_evaluation.py:127-134    -> (C+1,C+1) int64 on the device
    panoptic_png_rgb(panoptic_ids)                panoptic_evaluation.py:204 (id2rgb) -> (H,W,3) uint8, ready for the PNG encoder
    masks_to_rle(pred_masks)                      region_segmentation.py:282 (pycocotools mask.encode) -> [{"size": [h,w], "counts": bytes}]
    iou_counts(pred_masks, gt_masks, pairs)       referring_segmentation.py:101-113 (intersectionAndUnionGPU, K=2) -> intersection, union
    fuse_masks_by_score(pred_masks, scores, thr)  eval_grefcoco.py:113-131,277-285 (gRefCOCO: union of the candidates above thr, else top-1)
    IoUMeters                                     referring_segmentation.py:139-177 + :58-79 (cIoU / gIoU bookkeeping + all-reduce)
"""
from __future__ import annotations

from ctypes import c_long
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import hip_ops as H


def _ops(ops):
    return ops if ops is not None else H.get_ops()


def _on_device(o, t: torch.Tensor, dtypes, what: str) -> torch.Tensor:
    """The kernels read raw pointers: a tensor of another dtype would be misread (or read out of bounds), a host tensor -- the reference's
    evaluators habitually hold `.cpu()` copies -- would hand the GPU a host address.  Reject the former, move the latter."""
    if not torch.is_tensor(t) or t.dtype not in dtypes:
        raise H.PsalmHipError(f"{what}: dtype {getattr(t, 'dtype', type(t))} not supported (expected one of {', '.join(str(d) for d in dtypes)})")
    if t.dtype == torch.bool:
        t = t.view(torch.uint8)
    return t.to(o.device).contiguous()


def semantic_labels(sem_seg: torch.Tensor, ops=None) -> torch.Tensor:
    """`sem_seg.argmax(dim=0)` for a (C,H,W) float32 class map -> (H,W) int32 (first maximal class on ties, as torch)."""
    o = _ops(ops)
    if sem_seg.dim() != 3:
        raise H.PsalmHipError("semantic_labels: (C,H,W) class map")
    sem_seg = _on_device(o, sem_seg, (torch.float32,), "semantic_labels")
    C, Hh, Ww = sem_seg.shape
    out = o.empty(Hh, Ww, dtype=torch.int32)
    o._check(o.lib.psalm_semantic_labels(o._p(sem_seg), o._p(out), C, c_long(Hh * Ww), o._stream()), "psalm_semantic_labels")
    return out


class ConfusionMatrix:
    """my_SemSegEvaluator's `_conf_matrix` (panoptic_evaluation.py:127-134) kept on the device: rows = prediction, columns = ground truth,
    last row / column = the ignore bucket.  `update` takes int32 label maps (the prediction from `semantic_labels`)."""

    def __init__(self, num_classes: int, ignore_label: int = 255, ops=None):
        self.ops = _ops(ops)
        self.num_classes, self.ignore_label = num_classes, ignore_label
        self.conf = torch.zeros(num_classes + 1, num_classes + 1, dtype=torch.int64, device=self.ops.device)

    def update(self, pred: torch.Tensor, gt: torch.Tensor) -> None:
        o = self.ops
        gt = gt.to(o.device, torch.int32).contiguous()
        if pred.shape != gt.shape or pred.dtype != torch.int32:
            raise H.PsalmHipError("ConfusionMatrix.update: int32 label maps of equal shape")
        pred = pred.to(o.device).contiguous()
        o._check(o.lib.psalm_confusion_accumulate(o._p(pred), o._p(gt), c_long(pred.numel()), self.num_classes, self.ignore_label,
                                                  o._p(self.conf), o._stream()), "psalm_confusion_accumulate")

    def miou(self) -> Tuple[np.ndarray, float]:
        """Per-class IoU and mIoU as my_SemSegEvaluator.evaluate derives them from the matrix (detectron2 SemSegEvaluator.evaluate)."""
        conf = self.conf.cpu().numpy().astype(np.float64)
        tp = conf.diagonal()[:-1]
        pos_gt = conf[:-1, :-1].sum(0)
        pos_pred = conf[:-1, :-1].sum(1)
        union = pos_gt + pos_pred - tp
        valid = pos_gt > 0
        iou = np.where(union > 0, tp / np.maximum(union, 1), np.nan)
        return iou, float(np.nansum(iou[valid]) / max(valid.sum(), 1))


def panoptic_png_rgb(panoptic_ids: torch.Tensor, ops=None) -> torch.Tensor:
    """panopticapi `id2rgb`: id -> (id % 256, id // 256 % 256, id // 65536 % 256); (H,W) int32 -> (H,W,3) uint8 on the device."""
    o = _ops(ops)
    Hh, Ww = panoptic_ids.shape
    out = o.empty(Hh, Ww, 3, dtype=torch.uint8)
    o._check(o.lib.psalm_panoptic_rgb(o._p(panoptic_ids.to(o.device, torch.int32).contiguous()), o._p(out), c_long(Hh * Ww), o._stream()), "psalm_panoptic_rgb")
    return out


def rle_counts_to_string(counts: Sequence[int]) -> bytes:
    """pycocotools maskApi.c rleToString: LEB128-like, 5 data bits + continuation bit per char, chars 48..111, counts after the third
    stored as differences to the count two positions earlier."""
    out = bytearray()
    for i, c in enumerate(counts):
        x = int(c)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            ch = x & 0x1f
            x >>= 5
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(ch + 48)
    return bytes(out)


def masks_to_rle(masks: torch.Tensor, ops=None) -> List[dict]:
    """Binary masks (n,H,W) float32 | uint8 (nonzero = foreground) -> COCO RLE dicts, == `mask.encode(np.asfortranarray(m))` of pycocotools.
    The device finds the run boundaries (column-major); only those (4 bytes per run) are copied to the host."""
    o = _ops(ops)
    if masks.dim() != 3:
        raise H.PsalmHipError("masks_to_rle: (n,H,W) float32 / uint8 / bool masks")
    masks = _on_device(o, masks, (torch.float32, torch.uint8, torch.bool), "masks_to_rle")
    n, Hh, Ww = masks.shape
    if n == 0:
        return []
    u8 = 1 if masks.dtype == torch.uint8 else 0
    col_cnt = o.empty(n, Ww, dtype=torch.int32)
    col_off = o.empty(n, Ww, dtype=torch.int32)
    total = o.empty(n, dtype=torch.int32)
    o._check(o.lib.psalm_mask_rle_count(o._p(masks), u8, n, Hh, Ww, o._p(col_cnt), o._p(col_off), o._p(total), o._stream()), "psalm_mask_rle_count")
    tot = total.cpu().numpy().astype(np.int64)                 # the one sync: n small integers
    base = np.concatenate(([0], np.cumsum(tot)[:-1]))
    pos = o.empty(max(int(tot.sum()), 1), dtype=torch.int32)
    o._check(o.lib.psalm_mask_rle_emit(o._p(masks), u8, n, Hh, Ww, o._p(col_off), o._p(torch.from_numpy(base).to(o.device)), o._p(pos), o._stream()),
             "psalm_mask_rle_emit")
    pos = pos.cpu().numpy().astype(np.int64)
    out = []
    for i in range(n):
        b = pos[base[i]: base[i] + tot[i]]
        counts = np.diff(np.concatenate(([0], b, [Hh * Ww])))  # run lengths, starting with the zeros run (0 if the mask starts with 1)
        out.append({"size": [Hh, Ww], "counts": rle_counts_to_string(counts.tolist())})
    return out


def iou_counts(pred_masks: torch.Tensor, gt_masks: torch.Tensor, pairs: Sequence[Tuple[int, int]], ops=None):
    """intersectionAndUnionGPU(pred, gt, K=2, ignore_index=255) for each (prediction index, target index) pair, on the device:
    pred_masks (n,H,W) float32|uint8 (nonzero = 1), gt_masks (m,H,W) uint8 with 255 = ignore.
    Returns (intersection (P,2), union (P,2), target (P,2)) int64 tensors on the device (classes: background, foreground)."""
    o = _ops(ops)
    if pred_masks.dim() != 3 or gt_masks.dim() != 3:
        raise H.PsalmHipError("iou_counts: (n,H,W) predictions and (m,H,W) targets")
    pred_masks = _on_device(o, pred_masks, (torch.float32, torch.uint8, torch.bool), "iou_counts")
    gt = gt_masks.to(o.device, torch.uint8).contiguous()
    if tuple(pred_masks.shape[1:]) != tuple(gt.shape[1:]):
        raise H.PsalmHipError("iou_counts: prediction / target size mismatch")
    P = len(pairs)
    n, m = pred_masks.shape[0], gt.shape[0]
    if any(not (0 <= int(p) < n and 0 <= int(t) < m) for p, t in pairs):
        raise H.PsalmHipError(f"iou_counts: pair index out of range (predictions {n}, targets {m})")
    pi = torch.tensor([p for p, _ in pairs], dtype=torch.int32, device=o.device)
    ti = torch.tensor([t for _, t in pairs], dtype=torch.int32, device=o.device)
    counts = torch.zeros(P, 6, dtype=torch.int64, device=o.device)
    HW = int(pred_masks.shape[1] * pred_masks.shape[2])
    o._check(o.lib.psalm_iou_counts(o._p(pred_masks), 1 if pred_masks.dtype == torch.uint8 else 0, o._p(gt), o._p(pi), o._p(ti), P, c_long(HW),
                                    o._p(counts), o._stream()), "psalm_iou_counts")
    inter, outp, tgt = counts[:, 0:2], counts[:, 2:4], counts[:, 4:6]
    return inter, outp + tgt - inter, tgt


def fuse_masks_by_score(pred_masks: torch.Tensor, scores: torch.Tensor, thr: float = 0.6, ops=None) -> torch.Tensor:
    """gRefCOCO's fused prediction (psalm/eval/eval_grefcoco.py:113-131: `compute_metric` + `fuse_masks` :277-285) on the device: the
    union of the candidate masks whose score exceeds `thr`; when none does, the top-1 candidate (the reference's fall-back).
    pred_masks (n,H,W) float32|uint8|bool (nonzero = 1), scores (n) -> (H,W) uint8.  Feed the result to `iou_counts` / `IoUMeters`
    (the no-object target, union == 0 -> accuracy 1, is IoUMeters.update's rule, eval_grefcoco.py:147-148)."""
    o = _ops(ops)
    if pred_masks.dim() != 3 or scores.dim() != 1 or scores.shape[0] != pred_masks.shape[0] or not (1 <= pred_masks.shape[0] <= 1024):
        raise H.PsalmHipError("fuse_masks_by_score: (n,H,W) masks and (n) scores, 1 <= n <= 1024")
    pred_masks = _on_device(o, pred_masks, (torch.float32, torch.uint8, torch.bool), "fuse_masks_by_score")
    sc = _on_device(o, scores, (torch.float32,), "fuse_masks_by_score")
    n, Hh, Ww = pred_masks.shape
    out = torch.empty(Hh, Ww, dtype=torch.uint8, device=o.device)
    from ctypes import c_float
    o._check(o.lib.psalm_fuse_masks(o._p(pred_masks), 1 if pred_masks.dtype == torch.uint8 else 0, o._p(sc), n, c_long(Hh * Ww), c_float(float(thr)),
                                    o._p(out), o._stream()), "psalm_fuse_masks")
    return out


class IoUMeters:
    """The three AverageMeters of the referring / region evaluation loops (referring_segmentation.py:139-177: intersection, union,
    acc_iou) fed from device-side counts; `all_reduce()` = AverageMeter.all_reduce (:58-79) as ONE small SUM all-reduce."""

    def __init__(self):
        self.sum = torch.zeros(7, dtype=torch.float64)         # [I0, I1, U0, U1, acc0, acc1, n]

    def update(self, inter: torch.Tensor, union: torch.Tensor) -> None:
        """inter / union (P,2) for the evaluator's top-1 prediction of each of P samples (compute_metric, :139-171)."""
        i, u = inter.double().cpu(), union.double().cpu()
        acc = i / (u + 1e-5)
        acc[u == 0] = 1.0                                       # no-object target (:158)
        self.sum[0:2] += i.sum(0)
        self.sum[2:4] += u.sum(0)
        self.sum[4:6] += acc.sum(0)
        self.sum[6] += i.shape[0]

    def all_reduce(self, device=None) -> None:
        """SUM over ranks in float64 (pixel counts pass 2^24 within a few images).  `device`: where the collective runs; by default the
        current GPU under the nccl (RCCL) backend -- which reduces device tensors only, as the reference's AverageMeter.all_reduce knows
        (`.cuda()` first) -- and the host otherwise (gloo)."""
        import torch.distributed as dist
        from .dist import reduce_metrics
        if device is None and dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
            device = torch.device("cuda", torch.cuda.current_device())
        t = self.sum.to(device) if device is not None else self.sum.clone()
        self.sum = reduce_metrics(t).double().cpu()

    def results(self) -> dict:
        ciou = float(self.sum[1] / (self.sum[3] + 1e-10))      # iou_class[1], region_segmentation.py:286-287
        giou = float(self.sum[5] / max(float(self.sum[6]), 1e-5))
        return {"ciou": ciou, "giou": giou, "n": int(self.sum[6])}


def compact_results(result: dict, gt_sem: Optional[torch.Tensor] = None, conf: Optional[ConfusionMatrix] = None, ops=None) -> dict:
    """What the reference's panoptic / semantic / instance evaluators keep of ONE eval_seg result (panoptic_evaluation.py:114-145,179-222;
    instance masks as COCO RLE, region_segmentation.py:282), produced on the device, so that KBs..MBs instead of ~1 GB cross PCIe:
        "sem_labels"   (H,W) int32          from result["sem_seg"]           (+ conf.update(labels, gt_sem) when both are given)
        "panoptic_rgb" (H,W,3) uint8, "segments_info"   from result["panoptic_seg"]
        "instances"    {"rle": [...], "scores": (n,), "pred_classes": (n,)}  from result["instances"]
    Keys follow what the result holds (semantic-only / instance-only tasks give the matching subset)."""
    o = _ops(ops)
    out = {}
    if "sem_seg" in result:
        out["sem_labels"] = semantic_labels(result["sem_seg"].contiguous(), ops=o)
        if conf is not None and gt_sem is not None:
            conf.update(out["sem_labels"], gt_sem)
    if "panoptic_seg" in result:
        pan, info = result["panoptic_seg"]
        out["panoptic_rgb"] = panoptic_png_rgb(pan, ops=o)
        out["segments_info"] = info
    if "instances" in result and getattr(result["instances"], "pred_masks", None) is not None:
        inst = result["instances"]
        out["instances"] = {"rle": masks_to_rle(inst.pred_masks, ops=o), "scores": inst.scores}
        if hasattr(inst, "pred_classes"):
            out["instances"]["pred_classes"] = inst.pred_classes
    return out
