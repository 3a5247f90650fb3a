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
