"""Drop-in for the reference's compiled extension module `MultiScaleDeformableAttention`
(ops/setup.py:58-70; bound functions ops/src/vision.cpp:18-21), backed by the gfx950 kernel in
libpsalm_hip.so.  Put the repo root on sys.path and the reference's
`ops/functions/ms_deform_attn_func.py:21-29` imports this instead of the CUDA build, unchanged.

NOTE the reference wraps the op call in a bare `except:` and silently falls back to its PyTorch
formula (ops/modules/ms_deform_attn.py:112-119).  Set PSALM_MSDA_STRICT=1 to make any failure in
this module print loudly before re-raising, so a broken kernel cannot hide behind that fallback.
"""
import os
import sys
import traceback

import torch

from psalm_amd.hip_ops import get_ops


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    """value (B,S,M,D); spatial_shapes (L,2) int64; level_start_index (L,); sampling_loc (B,Lq,M,L,P,2);
    attn_weight (B,Lq,M,L,P) -> (B,Lq,M*D).  Same contiguity contract as ms_deform_attn_cuda.cu:33-43."""
    try:
        if not (value.is_contiguous() and sampling_loc.is_contiguous() and attn_weight.is_contiguous()):
            raise RuntimeError("value / sampling_loc / attn_weight tensor has to be contiguous")
        ops = get_ops()
        dt = value.dtype
        # the level table stays on the device, as in the reference op (ms_deform_attn_cuda.cu:64-75): no host copy / D2H sync on the seam
        out = ops.msda_forward_dev(value if dt in (torch.float32, torch.bfloat16) else value.float(),
                                   spatial_shapes.to(torch.int64), level_start_index.to(torch.int64),
                                   sampling_loc.float(), attn_weight.float())
        return out.to(dt)
    except Exception:
        if os.environ.get("PSALM_MSDA_STRICT"):
            print("[MultiScaleDeformableAttention/psalm_amd] forward failed:", file=sys.stderr)
            traceback.print_exc()
        raise


def ms_deform_attn_backward(*args, **kwargs):
    raise NotImplementedError("psalm_amd accelerates the inference path only (SURVEY.md §8: backward out of scope)")
