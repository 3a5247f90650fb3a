"""TEST INFRASTRUCTURE (oracle): numpy restatement of the reference's eval-time image pre-processing.

The reference delegates it to third-party code that is not under /root/reference:
  * detectron2 (unpinned, docs/INSTALL.md:30-32) `T.ResizeShortestEdge(short_edge_length=S, max_size=S)` + `T.FixedSizeCrop((S,S))`
    (call site: psalm/model/datasets_mapper/coco_panoptic_mapper.py:60-91), whose ResizeTransform.apply_image resizes uint8 images with
  * Pillow `Image.resize((w, h), Image.BILINEAR)` -- the algorithm restated here from Pillow's src/libImaging/Resample.c
    (precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc, ImagingResample; PRECISION_BITS = 22);
  * then `(image - pixel_mean) / pixel_std` in torch fp32 (coco_panoptic_mapper.py:158).
PINNED: tests/test_8_preprocess.py checks `resize_bilinear_u8` bit-for-bit against the installed Pillow (12.2.0) itself on random
images over up- and down-scaling shapes, so the restatement is as good as the library the reference calls.
Only tests/ may import this module; the product computes the tables in psalm_amd/preprocess.py and the pixels in HIP."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def precompute_coeffs(in_size: int, out_size: int):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle, support 1) filter over the full axis.
    Returns (bounds (out,2) int32 [xmin, count], kk (out,ksize) int32, ksize)."""
    in0, in1 = np.float32(0.0), np.float32(in_size)
    scale = float(in1 - in0) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = float(in0) + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = []
        for x in range(xmax):
            t = (x + xmin - center + 0.5) * ss
            t = -t if t < 0.0 else t
            w.append(1.0 - t if t < 1.0 else 0.0)
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _pass(img, bounds, kk, axis):
    """One 8-bit pass along `axis` (0 = vertical, 1 = horizontal) of an (H, W, C) uint8 image."""
    src = img.astype(np.int64)
    n = bounds.shape[0]
    shape = list(img.shape)
    shape[axis] = n
    out = np.zeros(shape, np.uint8)
    for i in range(n):
        lo, cnt = int(bounds[i, 0]), int(bounds[i, 1])
        k = kk[i, :cnt].astype(np.int64)
        if axis == 1:
            s = (src[:, lo:lo + cnt, :] * k[None, :, None]).sum(1)
        else:
            s = (src[lo:lo + cnt, :, :] * k[:, None, None]).sum(0)
        v = np.clip((s + (1 << (PRECISION_BITS - 1))) >> PRECISION_BITS, 0, 255).astype(np.uint8)
        if axis == 1:
            out[:, i, :] = v
        else:
            out[i, :, :] = v
    return out


def resize_bilinear_u8(img: np.ndarray, new_h: int, new_w: int) -> np.ndarray:
    """== np.asarray(PIL.Image.fromarray(img).resize((new_w, new_h), Image.BILINEAR)) for (H, W, 3) uint8."""
    h, w = img.shape[:2]
    out = img
    if new_w != w:                                   # horizontal pass first (ImagingResample)
        b, k, _ = precompute_coeffs(w, new_w)
        out = _pass(out, b, k, 1)
    if new_h != h:
        b, k, _ = precompute_coeffs(h, new_h)
        out = _pass(out, b, k, 0)
    return out


def resize_shortest_edge_shape(h, w, size, max_size):
    """detectron2 ResizeShortestEdge.get_output_shape."""
    scale = size * 1.0 / min(h, w)
    newh, neww = (size, scale * w) if h < w else (scale * h, size)
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * scale, neww * scale
    return int(newh + 0.5), int(neww + 0.5)


def preprocess(img: np.ndarray, size: int, mean, std):
    """The mapper's inference output: (image (3,S,S) float32, padding_mask (S,S) bool, (nh, nw))."""
    h, w = img.shape[:2]
    nh, nw = resize_shortest_edge_shape(h, w, size, size)
    res = resize_bilinear_u8(img, nh, nw)
    pad = np.full((size, size, 3), 128, np.uint8)
    pad[:nh, :nw] = res
    pm = np.ones((size, size), bool)
    pm[:nh, :nw] = False
    x = pad.transpose(2, 0, 1).astype(np.float32)
    out = (x - np.asarray(mean, np.float32).reshape(3, 1, 1)) / np.asarray(std, np.float32).reshape(3, 1, 1)
    return out, pm, (nh, nw)
