"""ctypes binding of the C ABI in include/psalm_hip.h (libpsalm_hip.so, hand-written gfx950 kernels).

PyTorch is used only as the device-memory container and stream owner: every method takes torch
tensors, checks layout, and hands raw `data_ptr()`s + sizes + the current HIP stream to the library.
There is NO fallback: if the library is missing, or a tensor is not on the GPU, this raises.

(The CPU test-suite constructs `Ops` over tests/emu/_build/libpsalm_emu.so -- the same kernel
sources compiled for the host -- explicitly via `Ops(path)`; `get_ops()` never does.)
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_long, c_void_p
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "lib", "libpsalm_hip.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU, ACT_GELU_NEW = 0, 1, 2, 3
_DT = {torch.float32: F32, torch.bfloat16: BF16}


class PsalmHipError(RuntimeError):
    pass


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise PsalmHipError(f"unsupported dtype {t.dtype} (float32 / bfloat16 only)")


class Ops:
    def __init__(self, lib_path: str = DEFAULT_LIB):
        if not os.path.exists(lib_path):
            raise PsalmHipError(
                f"{lib_path} not found: build the HIP kernels first (python -m psalm_amd.build). "
                "psalm_amd has no CPU / PyTorch fallback by design.")
        self.lib = ctypes.CDLL(lib_path)
        self.lib.psalm_last_error.restype = c_char_p
        self.lib.psalm_backend.restype = c_char_p
        self.backend = self.lib.psalm_backend().decode()
        self.is_emu = self.backend == "emu"
        self.device = torch.device("cpu") if self.is_emu else torch.device("cuda", torch.cuda.current_device())
        self.lib_path = lib_path

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        if self.is_emu:
            return c_void_p(0)
        return c_void_p(torch.cuda.current_stream().cuda_stream)

    def _p(self, t: Optional[torch.Tensor]):
        if t is None:
            return c_void_p(0)
        if not t.is_contiguous():
            raise PsalmHipError("tensor must be contiguous")
        if self.is_emu:
            if t.device.type != "cpu":
                raise PsalmHipError("emulation backend needs CPU tensors")
        elif t.device.type != "cuda":
            raise PsalmHipError("psalm_amd kernels need GPU tensors (no CPU fallback)")
        return c_void_p(t.data_ptr())

    def _check(self, rc: int, name: str):
        if rc != 0:
            raise PsalmHipError(f"{name} failed (rc={rc}): {self.lib.psalm_last_error().decode()}")

    def empty(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype, device=self.device)

    def zeros(self, *shape, dtype=torch.float32):
        return torch.zeros(*shape, dtype=dtype, device=self.device)

    # ------------------------------------------------------------------ MSDA
    def msda_forward(self, value, spatial_shapes: Sequence[Sequence[int]], level_start: Sequence[int], loc, attw,
                     out_dtype=None):
        """value (B,S,M,D); loc (B,Lq,M,L,P,2) f32; attw (B,Lq,M,L,P) f32 -> (B,Lq,M*D).
        Contract of the reference op MSDA.ms_deform_attn_forward (ops/src/ms_deform_attn.h:25-44)."""
        B, S, M, D = value.shape
        _, Lq, _, L, P, _ = loc.shape
        if loc.dtype != torch.float32 or attw.dtype != torch.float32:
            raise PsalmHipError("sampling locations / attention weights must be float32")
        out = self.empty(B, Lq, M * D, dtype=out_dtype or value.dtype)
        sh = (ctypes.c_int64 * (2 * L))(*[int(x) for hw in spatial_shapes for x in hw])
        st = (ctypes.c_int64 * L)(*[int(x) for x in level_start])
        rc = self.lib.psalm_msda_forward(self._p(value), _dt(value), sh, st, self._p(loc), self._p(attw), self._p(out),
                                         _dt(out), B, S, M, D, L, Lq, P, self._stream())
        self._check(rc, "psalm_msda_forward")
        return out

    def msda_fused(self, value, spatial_shapes, level_start, offsets_logits, M, out_dtype=None):
        """value (B,S,M*D); offsets_logits (B,S,M*L*P*3) f32 = [offsets | logits] -> (B,S,M*D)."""
        B, S, C = value.shape
        L, P = len(spatial_shapes), 4
        D = C // M
        if offsets_logits.dtype != torch.float32 or offsets_logits.shape[-1] != M * L * P * 3:
            raise PsalmHipError("offsets_logits must be float32 (B,S,M*L*P*3)")
        out = self.empty(B, S, C, dtype=out_dtype or value.dtype)
        sh = (ctypes.c_int64 * (2 * L))(*[int(x) for hw in spatial_shapes for x in hw])
        st = (ctypes.c_int64 * L)(*[int(x) for x in level_start])
        rc = self.lib.psalm_msda_fused(self._p(value), _dt(value), sh, st, self._p(offsets_logits), self._p(out), _dt(out),
                                       B, S, M, D, L, P, self._stream())
        self._check(rc, "psalm_msda_fused")
        return out


_OPS: Optional[Ops] = None


def get_ops() -> Ops:
    """The product accessor: libpsalm_hip.so on a GPU, or an exception."""
    global _OPS
    if _OPS is None:
        if not torch.cuda.is_available():
            raise PsalmHipError("no GPU visible: psalm_amd runs only on its HIP kernels (MI355X/gfx950); "
                                "there is deliberately no CPU fallback")
        _OPS = Ops(DEFAULT_LIB)
    return _OPS
