"""TEST INFRASTRUCTURE (oracle): numpy restatement of the host arithmetic of the reference's evaluators for the outputs of the hot path
(SURVEY.md §8 f2).  Sources followed:
  * `semantic_confusion`   psalm/eval/segmentation_evaluation/panoptic_evaluation.py:124-134 (my_SemSegEvaluator.process)
  * `id2rgb`               panopticapi/utils.py id2rgb (third party, un-vendored; call site panoptic_evaluation.py:204): base-256 digits
  * `rle_encode` / `rle_to_string` / `rle_from_string` / `rle_decode`
                           pycocotools common/maskApi.c rleEncode / rleToString / rleFrString / rleDecode (third party, un-vendored, not
                           installed here; call site psalm/eval/region_segmentation.py:282 `mask.encode(np.asfortranarray(pred_))`)
  * `intersection_and_union`  psalm/eval/referring_segmentation.py:101-113 (intersectionAndUnionGPU with torch.histc)
  * `compute_metric_update`   psalm/eval/referring_segmentation.py:139-171
PINNING: `intersection_and_union`, `compute_metric_update`, the cIoU / gIoU formulas and `semantic_confusion` are pinned against the
reference's OWN functions run in the authoring container (tests/golden/make_evalout_golden.py executes their source text from
/root/reference on seeded inputs -> tests/golden/evalout.npz; tests/test_8_evalout.py checks this module and the device path against it,
exactly).  "parity unpinned" remains for two third-party formats only -- the byte-exact COCO RLE string (pycocotools) and id2rgb
(panopticapi), both absent from the image: the RLE codec is held by hand-computed run lists, the string round trip and the structural
identities in tests/test_8_evalout.py (the run LENGTHS are unambiguous).  Only tests/ may import this module."""
import numpy as np


def semantic_confusion(sem_seg, gt, num_classes, ignore_label):
    pred = np.array(sem_seg.argmax(0), dtype=int)
    gt = np.array(gt, dtype=int).copy()
    gt[gt == ignore_label] = num_classes
    conf = np.bincount((num_classes + 1) * pred.reshape(-1) + gt.reshape(-1), minlength=(num_classes + 1) ** 2)
    return pred, conf.reshape(num_classes + 1, num_classes + 1)


def id2rgb(id_map):
    id_map_copy = id_map.copy().astype(np.int64)
    rgb = np.zeros(tuple(list(id_map.shape) + [3]), dtype=np.uint8)
    for i in range(3):
        rgb[..., i] = id_map_copy % 256
        id_map_copy //= 256
    return rgb


def rle_encode(mask):
    """maskApi.c rleEncode on np.asfortranarray(mask): counts of alternating 0 / 1 runs in column-major order, starting with 0s."""
    t = np.asfortranarray(mask.astype(np.uint8)).reshape(-1, order="F")
    cnts, p, c = [], 0, 0
    for v in t.tolist():
        if v != p:
            cnts.append(c)
            c, p = 0, v
        c += 1
    cnts.append(c)
    return cnts


def rle_to_string(cnts):
    s = bytearray()
    for i, c in enumerate(cnts):
        x = int(c)
        if i > 2:
            x -= int(cnts[i - 2])
        more = True
        while more:
            ch = x & 0x1f
            x >>= 5
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            s.append(ch + 48)
    return bytes(s)


def rle_from_string(s):
    cnts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = s[p] - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(cnts) > 2:
            x += cnts[-2]
        cnts.append(x)
    return cnts


def rle_decode(cnts, h, w):
    flat = np.zeros(h * w, np.uint8)
    pos, v = 0, 0
    for c in cnts:
        flat[pos:pos + c] = v
        pos += c
        v = 1 - v
    return flat.reshape(h, w, order="F")


def intersection_and_union(output, target, K=2, ignore_index=255):
    output = output.reshape(-1).astype(np.int64).copy()
    target = target.reshape(-1).astype(np.int64)
    output[target == ignore_index] = ignore_index
    inter = output[output == target]
    hist = lambda a: np.array([(a == k).sum() for k in range(K)], np.int64)     # torch.histc(bins=K, min=0, max=K-1): out-of-range dropped
    ai, ao, at = hist(inter), hist(output), hist(target)
    return ai, ao + at - ai, at


def compute_metric_update(meters, pred_top1, gt):
    """One sample of referring_segmentation.py:139-171 (topk = 1): meters = dict(I=, U=, acc=, n=) of float64 arrays / count."""
    inter, union, _ = intersection_and_union(pred_top1, gt)
    acc = inter / (union + 1e-5)
    acc[union == 0] = 1.0
    meters["I"] += inter
    meters["U"] += union
    meters["acc"] += acc
    meters["n"] += 1


def grefcoco_fused_prediction(preds, scores, thr=0.6):
    """eval_grefcoco.py:113-131 (compute_metric) with fuse_masks (:277-285): logical OR of the candidate masks with score > thr; none above
    thr -> the top-1 candidate (torch.topk(scores, 1): first maximal element).  preds (n,H,W) uint8, scores (n) -> (H,W) uint8."""
    preds = np.asarray(preds).astype(np.uint8)
    keep = [i for i, s_ in enumerate(scores) if s_ > thr]
    if not keep:
        keep = [int(np.argmax(np.asarray(scores)))]
    fused = np.zeros(preds.shape[1:], bool)
    for i in keep:
        fused |= preds[i] != 0
    return fused.astype(np.uint8)
