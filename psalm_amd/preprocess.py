"""Host side of the GPU image pre-processing (kernels in csrc/imageops.hip `psalm_image_preprocess`).

Coefficient tables of Pillow's 8-bit bilinear resampler (src/libImaging/Resample.c precompute_coeffs + normalize_coeffs_8bpc), the
resampler detectron2's ResizeTransform runs for the reference's `T.ResizeShortestEdge` (coco_panoptic_mapper.py:83-87): a few KB of
int32 per axis, computed once per (input size, output size) in double precision exactly as the C code does, cached."""
from __future__ import annotations

import functools
import math

import numpy as np

PRECISION_BITS = 22


@functools.lru_cache(maxsize=256)
def pil_bilinear_tables(in_size: int, out_size: int):
    """(bounds (out, 2) int32 [first input index, tap count], kk (out, ksize) int32 fixed-point weights, ksize)."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    filterscale = max(scale, 1.0)
    support = filterscale                                   # bilinear: filter support 1.0
    ksize = int(math.ceil(support)) * 2 + 1
    xx = np.arange(out_size, dtype=np.float64)
    center = 0.0 + (xx + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)          # C (int) cast of a non-negative double == floor
    xmin = np.where(center - support + 0.5 < 0, 0, xmin)
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    x = np.arange(ksize, dtype=np.float64)[None, :]
    t = np.abs((x + xmin[:, None] - center[:, None] + 0.5) * (1.0 / filterscale))
    w = np.where(t < 1.0, 1.0 - t, 0.0)
    w = np.where(x < xmax[:, None], w, 0.0)
    ww = np.zeros(out_size)
    for j in range(ksize):                                   # sequential sum, the C loop's order (bit-identical normalisation)
        ww = ww + w[:, j]
    k = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    kk = (0.5 + k * (1 << PRECISION_BITS)).astype(np.int64).astype(np.int32)   # weights are >= 0: (int)(0.5 + k * 2^22)
    kk = np.where(x < xmax[:, None], kk, 0).astype(np.int32)
    bounds = np.stack((xmin, xmax), 1).astype(np.int32)
    return np.ascontiguousarray(bounds), np.ascontiguousarray(kk), ksize


# ---------------------------------------------------------------------------------------------------- region prompts of the interactive task
# The dataset side of `processor.preprocess(data_dict, region_mask_type=...)` (train_datasets.py:333-336) for the interactive (point / box /
# scribble / mask prompt) task: coco_instance_mapper.py:233-252 turns each annotation's visual-prompt RLE into one (S, S) region mask --
#   decode -> enhance_with_circles(radius 10 for points, 5 for scribbles; :17-32) -> transforms.apply_segmentation (ResizeShortestEdge with the
#   NEAREST filter + FixedSizeCrop(seg_pad_value=0), :81-89) -> instances.region_masks.
# Host code (annotation handling, one prompt per object); the masks go to the device with the rest of `seg_info`.
REGION_MASK_TYPES = ("point_visual_prompt_mask", "mask_visual_prompt_mask", "box_visual_prompt_mask", "scribble_visual_prompt_mask")


def rle_to_mask(rle) -> np.ndarray:
    """pycocotools.mask.decode of one RLE dict {"size": [h, w], "counts": list | str | bytes}: (h, w) uint8, runs in column-major order starting
    with zeros; a str / bytes `counts` is the compressed form of maskApi.c rleFrString (5 data bits per character from '0', continuation bit
    0x20, sign extension by bit 0x10, counts beyond the second stored as differences from the count two back)."""
    h, w = int(rle["size"][0]), int(rle["size"][1])
    counts = rle["counts"]
    if isinstance(counts, str):
        counts = counts.encode("ascii")
    if isinstance(counts, (bytes, bytearray)):
        s, cnts, p = counts, [], 0
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
        counts = cnts
    counts = np.asarray(counts, dtype=np.int64)
    if counts.sum() != h * w:
        raise ValueError(f"RLE counts cover {int(counts.sum())} pixels, the mask has {h * w}")
    vals = (np.arange(len(counts)) & 1).astype(np.uint8)
    return np.repeat(vals, counts).reshape(h, w, order="F")


def enhance_with_circles(mask: np.ndarray, radius: int) -> np.ndarray:
    """coco_instance_mapper.py:17-32: the union of the discs {(y, x): sqrt((x - cx)^2 + (y - cy)^2) <= radius} around every set pixel (cy, cx) --
    a dilation by the integer disc dy^2 + dx^2 <= radius^2 (integer offsets: the square root test and the squared test agree).  Row by row instead of
    one full-image distance map per set pixel: for every vertical offset dy the rows are dilated horizontally by floor(sqrt(radius^2 - dy^2))."""
    m = (np.asarray(mask) == 1)
    H, W = m.shape
    out = np.zeros((H, W), dtype=bool)
    if not m.any():
        return out.astype(np.uint8)
    csum = np.concatenate([np.zeros((H, 1), np.int64), np.cumsum(m, axis=1, dtype=np.int64)], axis=1)      # csum[:, j] = set pixels in columns < j
    cols = np.arange(W)
    for dy in range(-radius, radius + 1):
        hw = int(math.isqrt(radius * radius - dy * dy))
        lo, hi = np.clip(cols - hw, 0, W), np.clip(cols + hw + 1, 0, W)
        dil = (csum[:, hi] - csum[:, lo]) > 0                         # row-wise: any set pixel within hw columns
        if dy >= 0:
            out[dy:, :] |= dil[:H - dy, :] if dy else dil
        else:
            out[:H + dy, :] |= dil[-dy:, :]
    return out.astype(np.uint8)


def apply_segmentation(mask: np.ndarray, transforms: dict) -> np.ndarray:
    """`transforms.apply_segmentation(mask)` of the reference's eval-time augmentation list for the (h, w) -> (S, S) geometry ImagePreprocessor
    recorded in `transforms` ({"resize": (h, w, nh, nw), "pad": (ph, pw)}): detectron2 ResizeTransform.apply_segmentation = Pillow NEAREST
    resize, FixedSizeCrop's PadTransform with seg_pad_value = 0 (coco_instance_mapper.py:86-89)."""
    from PIL import Image
    h, w, nh, nw = [int(v) for v in transforms["resize"]]
    ph, pw = [int(v) for v in transforms["pad"]]
    m = np.ascontiguousarray(np.asarray(mask, dtype=np.uint8))
    if m.shape != (h, w):
        raise ValueError(f"segmentation of shape {m.shape} for an image of {(h, w)}")
    if (nh, nw) != (h, w):
        m = np.asarray(Image.fromarray(m).resize((nw, nh), Image.NEAREST))
    return np.pad(m, ((0, ph), (0, pw)), mode="constant", constant_values=0)


def region_masks_from_annotations(annotations, transforms: dict, region_mask_type=None, rng=None):
    """coco_instance_mapper.py:233-252 for the non-crowd annotations of one image.  Returns (region_masks (k, S, S) uint8 array, indices of the
    annotations that received a prompt -- the `filter_annos` of the reference).  `region_mask_type`: the list of prompt kinds to draw from
    (None: all four, :234-236); `rng`: a `random.Random` (default: the `random` module itself, the stream the reference's `random.choice` draws from).
    Annotations without any non-empty prompt of the requested kinds are skipped (:242-243)."""
    import random as _random
    rng = rng or _random
    annos = [a for a in annotations if a.get("iscrowd", 0) == 0]
    kinds = list(region_mask_type) if region_mask_type is not None else list(REGION_MASK_TYPES)
    masks, kept = [], []
    if not annos or "point_visual_prompt_mask" not in annos[0]:
        return np.zeros((0, 0, 0), np.uint8), kept
    for i, anno in enumerate(annos):
        non_empty = [k for k in kinds if anno.get(k) is not None and rle_to_mask(anno[k]).sum() > 0]       # is_mask_non_empty, :35-39
        if not non_empty:
            continue
        used = rng.choice(non_empty)
        region = rle_to_mask(anno[used])
        if used in ("point_visual_prompt_mask", "scribble_visual_prompt_mask"):
            region = enhance_with_circles(region, 10 if used == "point_visual_prompt_mask" else 5)
        masks.append(apply_segmentation(region, transforms))
        kept.append(i)
    return (np.stack(masks) if masks else np.zeros((0, 0, 0), np.uint8)), kept
