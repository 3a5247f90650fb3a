"""ctypes binding of the C ABI in include/psalm_hip.h (libpsalm_hip.so, hand-written gfx950 kernels).

PyTorch is used only as the device-memory container and stream owner: every method takes torch
tensors, checks layout, and hands raw `data_ptr()`s + sizes + the current HIP stream to the library.
There is NO fallback: if the library is missing, or a tensor is not on the GPU, this raises.

(The CPU test-suite constructs `Ops` over tests/emu/_build/libpsalm_emu.so -- the same kernel
sources compiled for the host -- explicitly via `Ops(path)`; `get_ops()` never does.)
"""
from __future__ import annotations

import ctypes
import math
import os
import sys
from ctypes import c_char_p, c_float, c_int, c_long, c_void_p
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "lib", "libpsalm_hip.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU, ACT_GELU_NEW = 0, 1, 2, 3
ACT_POST_RESIDUAL = 16
ACT_BIAS_ROW = 32
_DT = {torch.float32: F32, torch.bfloat16: BF16}


_DEBUG_SYNC = os.environ.get("PSALM_DEBUG_SYNC", "0") not in ("", "0")
# PSALM_DEBUG_BOUNDS=1: after every split-output GEMM, compare the magnitude BOUND behind its row scales with what the rows actually hold
# (one device round trip per call -- a debugging aid, off in production): the emitted operand keeps its 22 bits while bound / actual row
# maximum <= 2^14 (tests/test_2_gemm.py::test_gemm_x3_split_output_loose_bound); trained weights with outlier channels can be far looser
# than the seeded Gaussian ones of the tests, and this is how to find out.
_DEBUG_BOUNDS = os.environ.get("PSALM_DEBUG_BOUNDS", "0") not in ("", "0")
BOUND_LOOSENESS_LIMIT = 2.0 ** 14


class PsalmHipError(RuntimeError):
    pass


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise PsalmHipError(f"unsupported dtype {t.dtype} (float32 / bfloat16 only)")


ABI_VERSION = 7        # == PSALM_ABI_VERSION of include/psalm_hip.h (tests/test_0_abi.py compares the two and the built library's answer)


class _ProfiledLib:
    """Transparent proxy over the CDLL: when `records` is a list, every psalm_* launch is bracketed by a pair of
    HIP events on the launch stream (torch's current stream) so bench.py can attribute time per kernel family."""

    def __init__(self, cdll):
        object.__setattr__(self, "_cdll", cdll)
        object.__setattr__(self, "records", None)

    def __getattr__(self, name):
        fn = getattr(self._cdll, name)
        if not name.startswith("psalm_") or name in ("psalm_last_error", "psalm_backend", "psalm_abi_version", "psalm_gemm_last_kernel",
                                                         "psalm_gemm_set_tile_policy", "psalm_gemm_x3_set_products", "psalm_set_tuning",
                                                         "psalm_get_tuning") or name.endswith("_workspace"):
            return fn

        def call(*args):
            rec = self.records
            if _DEBUG_SYNC:
                # PSALM_DEBUG_SYNC=1: name every launch before it is issued and synchronise after it, so that a GPU memory
                # fault (which aborts the process asynchronously) is attributable: the last "launch" line without "ok".
                sys.stderr.write(f"[psalm launch] {name}\n"); sys.stderr.flush()
                rc = fn(*args)
                if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
                    torch.cuda.synchronize()
                sys.stderr.write(f"[psalm ok] {name}\n"); sys.stderr.flush()
                return rc
            if rec is None:
                return fn(*args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            kern = None
            if name.startswith(("psalm_gemm", "psalm_conv2d")):   # exact template instantiation the library launched (as a kernel trace names it)
                kern = self._cdll.psalm_gemm_last_kernel().decode()
            rec.append((name, tuple(a.value if hasattr(a, "value") else a for a in args), e0, e1, kern))
            return rc
        return call

    def __setattr__(self, k, v):
        if k == "records":
            object.__setattr__(self, k, v)
        else:
            setattr(self._cdll, k, v)


class SplitF16:
    """A matrix in split-f16 form (psalm_split_f16): `t` (rows, 2*Kp) float16 = [hi | lo], `inv_scale` (rows,) float32, logical
    shape (rows, K), Kp = ceil64(K).  Either operand of `Ops.gemm` may be one; in the "f16x3" mode GEMM weights are kept in this form."""
    __slots__ = ("t", "inv_scale", "K", "Kp")

    def __init__(self, t, inv_scale, K):
        self.t, self.inv_scale, self.K, self.Kp = t, inv_scale, K, t.shape[1] // 2

    @property
    def shape(self):
        return (self.t.shape[0], self.K)

    @property
    def dtype(self):
        return torch.float32                      # the values it stands for


class _PostDesc(ctypes.Structure):           # psalm_post_desc of include/psalm_hip.h
    _fields_ = [(n, c_int) for n in ("task", "Q", "h", "w", "Hpad", "Wpad", "crop_h", "crop_w", "out_h", "out_w", "C1", "n_region")] + \
               [("obj_thr", c_float), ("overlap_thr", c_float)]


class _PostIO(ctypes.Structure):             # psalm_post_io
    _fields_ = [(n, c_void_p) for n in ("pred_masks", "mask_up", "cls_logits", "seg_logits", "region_logits", "is_thing", "mask_pred", "sem_seg", "scores",
                                        "classes", "query", "inst_masks", "boxes", "pan", "counts")]


class _PhiLayer(ctypes.Structure):           # psalm_phi_layer of include/psalm_hip.h
    _fields_ = [("w1", c_void_p), ("w1_scale", c_void_p), ("b1", c_void_p), ("w2", c_void_p), ("w2_scale", c_void_p), ("b2", c_void_p),
                ("ln_g", c_void_p), ("ln_b", c_void_p), ("bnd", c_void_p), ("paired", c_int)]


class _PhiDesc(ctypes.Structure):            # psalm_phi_desc
    _fields_ = [("num_layers", c_int), ("hidden", c_int), ("intermediate", c_int), ("heads", c_int), ("head_dim", c_int), ("rot", c_int),
                ("ln_eps", c_float), ("layers", ctypes.POINTER(_PhiLayer)), ("final_g", c_void_p), ("final_b", c_void_p)]


class _SwinBlock(ctypes.Structure):          # psalm_swin_block
    _fields_ = [(n, c_void_p) for n in ("n1_g", "n1_b", "n2_g", "n2_b", "qkv_w", "qkv_ws", "qkv_b", "qkv_bnd", "proj_w", "proj_ws", "proj_b",
                                         "fc1_w", "fc1_ws", "fc1_b", "fc1_bnd")] + [("fc1_paired", c_int)] + \
               [(n, c_void_p) for n in ("fc2_w", "fc2_ws", "fc2_b", "rpb")]


class _SwinStage(ctypes.Structure):          # psalm_swin_stage
    _fields_ = [("depth", c_int), ("heads", c_int), ("dim", c_int), ("blocks", ctypes.POINTER(_SwinBlock)), ("out_g", c_void_p), ("out_b", c_void_p),
                ("ds_g", c_void_p), ("ds_b", c_void_p), ("ds_w", c_void_p), ("ds_ws", c_void_p)]


class _SwinDesc(ctypes.Structure):           # psalm_swin_desc
    _fields_ = [("num_stages", c_int), ("patch", c_int), ("window", c_int), ("pe_kpad", c_int), ("mlp_ratio", c_int), ("pe_w", c_void_p),
                ("pe_ws", c_void_p), ("pe_b", c_void_p), ("pe_ln_g", c_void_p), ("pe_ln_b", c_void_p), ("stages", ctypes.POINTER(_SwinStage))]


class _ProjDesc(ctypes.Structure):           # psalm_projector_desc
    _fields_ = [("in_dim", c_int), ("mid_dim", c_int), ("out_dim", c_int)] + \
               [(n, c_void_p) for n in ("c1_w", "c1_ws", "c1_b", "c2_w", "c2_ws", "c2f_w", "c2f_ws", "c2f_b", "ds_w", "ds_ws", "ds_b", "fc_w", "fc_ws", "fc_b")]


class _PdEncLayer(ctypes.Structure):         # psalm_pd_enc_layer
    _fields_ = [(n, c_void_p) for n in ("value_w", "value_ws", "value_b", "ow_w", "ow_ws", "ow_b", "out_w", "out_ws", "out_b", "n1_g", "n1_b",
                                         "l1_w", "l1_ws", "l1_b", "l1_bnd")] + [("l1_paired", c_int)] + \
               [(n, c_void_p) for n in ("l2_w", "l2_ws", "l2_b", "n2_g", "n2_b")]


class _PdDesc(ctypes.Structure):             # psalm_pd_desc
    _fields_ = [("D", c_int), ("G", c_int), ("M", c_int), ("num_layers", c_int), ("ffn", c_int), ("mask_dim", c_int), ("in_dims", c_int * 4),
                ("ip_w", c_void_p * 3), ("ip_ws", c_void_p * 3), ("ip_b", c_void_p * 3), ("ip_gn_g", c_void_p * 3), ("ip_gn_b", c_void_p * 3)] + \
               [(n, c_void_p) for n in ("adapter_w", "adapter_ws", "adapter_b", "adapter_gn_g", "adapter_gn_b", "layer_w", "layer_ws", "layer_b",
                                         "layer_gn_g", "layer_gn_b", "mf_w", "mf_ws", "mf_b")] + [("layers", ctypes.POINTER(_PdEncLayer))]


class _PrLayer(ctypes.Structure):            # psalm_pr_layer
    _fields_ = [(n, c_void_p) for n in ("cq_w", "cq_b", "co_w", "co_b", "cn_g", "cn_b", "sqk_w", "sqk_b", "sv_w", "sv_b", "so_w", "so_b", "sn_g", "sn_b",
                                         "f1_w", "f1_b", "f2_w", "f2_b", "fn_g", "fn_b")]


class _PrDesc(ctypes.Structure):             # psalm_pr_desc
    _fields_ = [(n, c_int) for n in ("D", "heads", "Q", "num_layers", "num_levels", "ffn", "mask_dim")] + \
               [(n, c_void_p * 3) for n in ("lvl_k_w", "lvl_k_ws", "lvl_k_b", "lvl_v_w", "lvl_v_ws", "lvl_v_b")] + \
               [(n, c_void_p) for n in ("level_embed", "query_embed", "dn_g", "dn_b")] + [("mask_embed_w", c_void_p * 3), ("mask_embed_b", c_void_p * 3)] + \
               [(n, c_void_p * 2) for n in ("SEG_w", "SEG_b", "CLASS_w", "CLASS_b", "REGION_w", "REGION_b")] + [("layers", ctypes.POINTER(_PrLayer))]


class Ops:
    # psalm_gemm_set_tile_policy code of the library's default K loop for split-f16 GEMMs on 256 x 256 tiles (2580 K-panel form, 2581 32-deep
    # slices, 2582 slices with the all-padding m-tiles left out): what tests that switch it restore afterwards.
    GEMM_X3_256_DEFAULT = 2582

    def __init__(self, lib_path: str = DEFAULT_LIB):
        if not os.path.exists(lib_path):
            raise PsalmHipError(
                f"{lib_path} not found: build the HIP kernels first (python -m psalm_amd.build). "
                "psalm_amd has no CPU / PyTorch fallback by design.")
        cdll = ctypes.CDLL(lib_path)
        # the binary interface this binding was written against (include/psalm_hip.h PSALM_ABI_VERSION): checked BEFORE any other call -- a
        # stale prebuilt library would take this binding's integers for pointers (silent memory corruption, not an error)
        got = cdll.psalm_abi_version() if hasattr(cdll, "psalm_abi_version") else -1
        if got != ABI_VERSION:
            raise PsalmHipError(f"{lib_path}: psalm_abi_version() = {got}, this binding needs {ABI_VERSION}: rebuild the library "
                                "(python -m psalm_amd.build --force)")
        cdll.psalm_last_error.restype = c_char_p
        cdll.psalm_backend.restype = c_char_p
        cdll.psalm_gemm_last_kernel.restype = c_char_p
        self._cdll_raw = cdll
        self.lib = _ProfiledLib(cdll)
        self.backend = self.lib.psalm_backend().decode()
        self.is_emu = self.backend == "emu"
        self.device = torch.device("cpu") if self.is_emu else torch.device("cuda", torch.cuda.current_device())
        self.lib_path = lib_path
        self._ws = {}                      # cached kernel workspaces (device buffers owned by this binding)
        self._msda_tables = set()          # device-side MSDA level tables already validated (msda_forward_dev)
        self.debug_bounds = False          # True (or PSALM_DEBUG_BOUNDS=1): verify the scale bound of every split-output GEMM (host round trip)
        self.bound_looseness_max = 1.0
        self.x3 = False                    # True: float32 x float32 GEMMs run in split-f16 arithmetic (precision="f16x3")

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

    def _pv(self, t: Optional[torch.Tensor]):
        """pointer of a possibly row-strided view (caller passes the stride)."""
        if t is None:
            return c_void_p(0)
        if self.is_emu != (t.device.type == "cpu"):
            raise PsalmHipError("tensor on the wrong device for this backend (no CPU fallback)")
        return c_void_p(t.data_ptr())

    def _check(self, rc: int, name: str):
        if rc != 0:
            raise PsalmHipError(f"{name} failed (rc={rc}): {self.lib.psalm_last_error().decode()}")

    def _stage_ws(self, name, nbytes):
        """Workspace of a stage-level call: ONE buffer per (stage, launch stream), grown on demand and kept for the life of the binding (a
        fresh hundreds-of-MB torch.empty per call left the buffer's lifetime to the caching allocator's stream-ordered reuse and churned the allocator
        in eager mode).  A stage's successive calls on one stream are stream-ordered, so they may share it; calls on different streams get their own.
        While a hipGraph is being captured the buffer must belong to the capture's own pool: a fresh allocation then, as before."""
        if self.is_emu or torch.cuda.is_current_stream_capturing():
            return torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        key = ("stage_ws", name, torch.cuda.current_stream().cuda_stream)
        t = self._ws.get(key)
        if t is None or t.numel() < nbytes:
            t = self._ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return t

    GEMM_WS_BYTES = 96 << 20

    def _gemm_ws(self):
        """Caller-owned split-K scratch handed to psalm_gemm: one buffer per launch stream for the life of the binding
        (two GEMMs on different streams may run concurrently and must not share partial-sum slabs)."""
        key = ("gemm", 0 if self.is_emu else torch.cuda.current_stream().cuda_stream)
        ws = self._ws.get(key)
        if ws is None:
            ws = self._ws[key] = torch.empty(self.GEMM_WS_BYTES, dtype=torch.uint8, device=self.device)
        return ws

    def empty(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype, device=self.device)

    def zeros(self, *shape, dtype=torch.float32):
        t = torch.empty(*shape, dtype=dtype, device=self.device)
        self._check(self.lib.psalm_memset_zero(self._p(t), c_long(t.numel() * t.element_size()), self._stream()), "psalm_memset_zero")
        return t

    def to_i64(self, t):
        """int32 vector -> int64 (psalm_cast_i32_i64): the reference's label / index tensors are LongTensors"""
        if t.dtype != torch.int32 or not t.is_contiguous():
            raise PsalmHipError("to_i64: contiguous int32 tensor")
        out = self.empty(*t.shape, dtype=torch.int64)
        self._check(self.lib.psalm_cast_i32_i64(self._p(t), self._p(out), c_long(t.numel()), self._stream()), "psalm_cast_i32_i64")
        return out

    def copy_(self, dst, src):
        """dst <- src for two contiguous device tensors of equal dtype / numel, as a stream copy (no framework kernel)."""
        if dst.dtype != src.dtype or dst.numel() != src.numel() or not (dst.is_contiguous() and src.is_contiguous()) or src.device != dst.device:
            dst.copy_(src)
            return dst
        self._check(self.lib.psalm_copy_d2d(self._p(dst), self._p(src), c_long(dst.numel() * dst.element_size()), self._stream()), "psalm_copy_d2d")
        return dst

    # ------------------------------------------------------------------ GEMM
    def gemm(self, a, w, bias=None, residual=None, act=ACT_NONE, act_col_start=0, out=None, out_dtype=None):
        """out[M,N] = act(a[M,K] @ w[N,K]^T + bias) + residual.  `a`, `out`, `residual` may be row-strided 2-D views
        (last dim contiguous).  w.dtype selects the arithmetic: bfloat16 -> bf16 MFMA / fp32 accumulate,
        float32 -> exact fp32 MFMA."""
        if isinstance(a, SplitF16) or isinstance(w, SplitF16) or (self.x3 and a.dtype == torch.float32 and w.dtype == torch.float32):
            return self.gemm_x3(a, w, bias, residual, act, act_col_start, out, out_dtype)
        if a.dim() != 2 or w.dim() != 2 or a.shape[1] != w.shape[1]:
            raise PsalmHipError(f"gemm shape mismatch {tuple(a.shape)} x {tuple(w.shape)}")
        M, K = a.shape
        N = w.shape[0]
        if out is None:
            out = self.empty(M, N, dtype=out_dtype or (torch.float32 if w.dtype == torch.float32 else a.dtype))
        for t in (a, w, out) + ((residual,) if residual is not None else ()):
            if t.stride(-1) != 1:
                raise PsalmHipError("gemm operands need a contiguous last dimension")
        if residual is not None and (residual.dtype != out.dtype or residual.shape != out.shape):
            raise PsalmHipError("gemm residual must match the output's dtype and shape")
        if bias is not None and (bias.dtype != torch.float32 or bias.numel() != (M if act & ACT_BIAS_ROW else N)):
            raise PsalmHipError("gemm bias must be float32 (N,) -- or (M,) with ACT_BIAS_ROW")
        rc = self.lib.psalm_gemm(self._pv(a), _dt(a), c_long(a.stride(0)), self._pv(w), _dt(w), c_long(w.stride(0)),
                                 self._pv(bias), self._pv(residual), c_long(residual.stride(0) if residual is not None else 0),
                                 self._pv(out), _dt(out), c_long(out.stride(0)), M, N, K, act, act_col_start,
                                 self._p(self._gemm_ws()), c_long(self.GEMM_WS_BYTES), self._stream())
        self._check(rc, "psalm_gemm")
        return out

    def split_f16(self, x):
        """x (rows,K) float32 (row-strided view) -> SplitF16: x * s = hi + lo in float16 with a per-row power-of-two scale."""
        if isinstance(x, SplitF16):
            return x
        if x.dim() != 2 or x.dtype != torch.float32 or x.stride(1) != 1:
            raise PsalmHipError("split_f16: 2-D float32 input with a contiguous last dimension")
        rows, K = x.shape
        Kp = (K + 63) // 64 * 64
        t = self.empty(rows, 2 * Kp, dtype=torch.float16)
        inv = self.empty(rows, dtype=torch.float32)
        rc = self.lib.psalm_split_f16(self._pv(x), c_long(x.stride(0)), self._p(t), c_long(2 * Kp), self._p(inv), rows, K, self._stream())
        self._check(rc, "psalm_split_f16")
        return SplitF16(t, inv, K)

    def _x3_operands(self, a, w):
        """(a, w) for a split-f16 GEMM: float32 tensors are split on the fly."""
        a, w = self.split_f16(a), self.split_f16(w)
        if a.K != w.K or a.Kp != w.Kp:
            raise PsalmHipError(f"gemm shape mismatch {a.shape} x {w.shape}")
        return a, w

    def gemm_x3(self, a, w, bias=None, residual=None, act=ACT_NONE, act_col_start=0, out=None, out_dtype=None):
        """gemm() on split-f16 operands (float32 tensors are split on the fly): fp32-class result on the f16 matrix cores."""
        a, w = self._x3_operands(a, w)
        M, N = a.t.shape[0], w.t.shape[0]
        if out is None:
            if out_dtype not in (None, torch.float32):
                raise PsalmHipError("gemm_x3: float32 output only")
            out = self.empty(M, N, dtype=torch.float32)
        for t in (out,) + ((residual,) if residual is not None else ()):
            if t.dtype != torch.float32 or t.stride(-1) != 1 or tuple(t.shape) != (M, N):
                raise PsalmHipError("gemm_x3: float32 (M,N) output / residual with a contiguous last dimension")
        if bias is not None and (bias.dtype != torch.float32 or bias.numel() != (M if act & ACT_BIAS_ROW else N)):
            raise PsalmHipError("gemm bias must be float32 (N,) -- or (M,) with ACT_BIAS_ROW")
        rc = self.lib.psalm_gemm_x3(self._p(a.t), c_long(a.t.stride(0)), self._p(a.inv_scale), self._p(w.t), c_long(w.t.stride(0)),
                                    self._p(w.inv_scale), a.Kp, self._pv(bias), self._pv(residual),
                                    c_long(residual.stride(0) if residual is not None else 0), self._pv(out), c_long(out.stride(0)),
                                    M, N, act, act_col_start, self._p(self._gemm_ws()), c_long(self.GEMM_WS_BYTES), self._stream())
        self._check(rc, "psalm_gemm_x3")
        return out

    def gemm_x3_ln_split(self, a, w, bias, residual, gamma, beta, eps, want_y=False):
        """x = a.w^T + bias + residual (float32), h = LayerNorm(x): returns (x, SplitF16(h), h float32 | None) -- the split-K reduce, the
        LayerNorm and the split of h are one row pass (psalm_gemm_x3_ln_split)."""
        a, w = self._x3_operands(a, w)
        M, N = a.t.shape[0], w.t.shape[0]
        if residual is not None and (residual.dtype != torch.float32 or tuple(residual.shape) != (M, N) or residual.stride(-1) != 1):
            raise PsalmHipError("gemm_x3_ln_split: float32 (M,N) residual")
        x = self.empty(M, N, dtype=torch.float32)
        y = self.empty(M, N, dtype=torch.float32) if want_y else None
        so, inv = self.empty(M, 2 * N, dtype=torch.float16), self.empty(M, dtype=torch.float32)
        rc = self.lib.psalm_gemm_x3_ln_split(self._p(a.t), c_long(a.t.stride(0)), self._p(a.inv_scale), self._p(w.t), c_long(w.t.stride(0)),
                                             self._p(w.inv_scale), a.Kp, self._pv(bias), self._pv(residual),
                                             c_long(residual.stride(0) if residual is not None else 0), self._p(x), c_long(N), M, N,
                                             self._p(gamma), self._p(beta), c_float(eps), self._pv(y), c_long(N), self._p(so), self._p(inv),
                                             self._p(self._gemm_ws()), c_long(self.GEMM_WS_BYTES), self._stream())
        self._check(rc, "psalm_gemm_x3_ln_split")
        return x, SplitF16(so, inv, N), y

    def gemm_x3_split(self, a, w, bias, act, split_out, split_inv, bound_par, split_col_off=0, split_col_start=0, act_col_start=0,
                      out=None, global_rows=False, paired=False):
        """gemm_x3 whose columns >= split_col_start are written as the split-f16 A operand of the next GEMM: into `split_out` (a SplitF16's
        .t buffer (M, 2*Kp_out) f16) at columns split_col_off.. (hi) / Kp_out + split_col_off.. (lo), row scales (inverse) into split_inv;
        bound_par: 4 device floats, see psalm_gemm_x3_split.  Columns below split_col_start go to `out` (M, >= split_col_start...) fp32.
        paired: the rows of `w` (and bias) >= split_col_start were permuted with `so_pair_perm` (stores straight from the accumulators)."""
        a, w = self._x3_operands(a, w)
        M, N = a.t.shape[0], w.t.shape[0]
        if split_out.dtype != torch.float16 or split_out.dim() != 2 or split_out.shape[0] != M or split_out.stride(1) != 1:
            raise PsalmHipError("gemm_x3_split: split_out must be a (M, 2*Kp) float16 buffer")
        if split_inv.dtype != torch.float32 or split_inv.numel() != M or bound_par.dtype != torch.float32 or bound_par.numel() != 4:
            raise PsalmHipError("gemm_x3_split: split_inv (M,) / bound_par (4,) float32")
        if split_col_start > 0 and (out is None or out.dtype != torch.float32 or out.shape[0] != M or out.stride(-1) != 1):
            raise PsalmHipError("gemm_x3_split: float32 `out` required for the columns below split_col_start")
        if bias is not None and (bias.dtype != torch.float32 or bias.numel() != N):
            raise PsalmHipError("gemm bias must be float32 (N,)")
        rc = self.lib.psalm_gemm_x3_split(self._p(a.t), c_long(a.t.stride(0)), self._p(a.inv_scale), self._p(w.t), c_long(w.t.stride(0)),
                                          self._p(w.inv_scale), a.Kp, self._pv(bias), self._pv(out),
                                          c_long(out.stride(0) if out is not None else 0), M, N, act, act_col_start,
                                          self._p(split_out), c_long(split_out.stride(0)), split_out.shape[1] // 2, split_col_off,
                                          split_col_start, int(bool(paired)), self._p(split_inv), self._p(bound_par), int(bool(global_rows)),
                                          self._p(self._gemm_ws()), c_long(self.GEMM_WS_BYTES), self._stream())
        self._check(rc, "psalm_gemm_x3_split")
        if _DEBUG_BOUNDS or self.debug_bounds:
            self.check_split_bound(split_out, split_col_off, N - split_col_start, "gemm_x3_split")
        return split_out

    @staticmethod
    def so_pair_perm(n: int) -> torch.Tensor:
        """Row permutation of a weight (and its bias) whose GEMM emits split-f16 output with paired stores (psalm_gemm_x3_split,
        `paired`): index tensor `c` with W_physical = W_logical[c]; physical row 64 g + 32 b + j holds logical row 64 g + 2 j + b.  n % 64 == 0."""
        if n % 64:
            raise PsalmHipError("so_pair_perm: n % 64 == 0")
        p = torch.arange(n)
        return (p // 64) * 64 + 2 * (p % 32) + (p % 64) // 32

    def check_split_bound(self, split_out, col_off, ncols, what=""):
        """Looseness of the scale bound of an emitted split-f16 operand: the scale puts the BOUND in [2^12, 2^13), so 2^13 / max |hi| over a
        row's emitted columns is (within 2x) bound / actual row maximum.  Records the worst row in `bound_looseness_max`; raises beyond
        BOUND_LOOSENESS_LIMIT (lo would be reaching the f16 subnormal floor: the operand no longer carries 22 bits)."""
        hi = split_out[:, col_off:col_off + ncols].float().abs().amax(1)
        live = hi > 0
        if not bool(live.any()):
            return 1.0
        loose = float((2.0 ** 13 / hi[live]).max())
        self.bound_looseness_max = max(self.bound_looseness_max, loose)
        if loose > BOUND_LOOSENESS_LIMIT:
            raise PsalmHipError(f"{what}: the magnitude bound behind the split-f16 output scale is 2^{math.log2(loose):.1f} above the "
                                f"actual row maximum (limit 2^14): lo is reaching the f16 subnormal floor -- run this GEMM without the "
                                "fused split output (PSALM.fuse_split = False) or tighten its bound")
        return loose

    def gemm_ln(self, a, w, bias, residual, gamma, beta, eps=1e-5, ln_dtype=torch.bfloat16, act=ACT_NONE):
        """(C fp32, LayerNorm(C) ln_dtype) with C = act(a @ w^T + bias) + residual; bf16 a / w, K % 64 == 0."""
        M, K = a.shape
        N = w.shape[0]
        out = self.empty(M, N, dtype=torch.float32)
        ln_out = self.empty(M, N, dtype=ln_dtype)
        if residual is not None and (residual.dtype != torch.float32 or residual.shape != out.shape):
            raise PsalmHipError("gemm_ln: residual must be float32 (M,N)")
        rc = self.lib.psalm_gemm_ln(self._pv(a), _dt(a), c_long(a.stride(0)), self._pv(w), _dt(w), c_long(w.stride(0)), self._p(bias),
                                    self._pv(residual), c_long(residual.stride(0) if residual is not None else 0), self._p(out), F32,
                                    c_long(N), M, N, K, act, 0, self._p(gamma), self._p(beta), c_float(eps), self._p(ln_out), _dt(ln_out),
                                    c_long(N), self._p(self._gemm_ws()), c_long(self.GEMM_WS_BYTES), self._stream())
        self._check(rc, "psalm_gemm_ln")
        return out, ln_out

    def conv2d_nhwc(self, x, B, H, W, wt, ksize, stride, pad, bias=None, residual=None, act=ACT_NONE, out_dtype=None):
        """Implicit-GEMM convolution: x (B*H*W, Cin) bf16 NHWC tokens, wt (Cout, k*k*Cin) bf16 (K order ky,kx,c) -> (B*Ho*Wo, Cout)."""
        Cin, Cout = x.shape[-1], wt.shape[0]
        if x.dtype != torch.bfloat16 or wt.dtype != torch.bfloat16 or wt.shape[1] != ksize * ksize * Cin:
            raise PsalmHipError("conv2d_nhwc: bf16 operands, weight (Cout, k*k*Cin)")
        Ho, Wo = (H + 2 * pad - ksize) // stride + 1, (W + 2 * pad - ksize) // stride + 1
        out = self.empty(B * Ho * Wo, Cout, dtype=out_dtype or x.dtype)
        if residual is not None and (residual.dtype != out.dtype or residual.shape != out.shape):
            raise PsalmHipError("conv2d_nhwc: residual must match the output")
        z = self._ws.get("zeros")
        if z is None:
            z = self._ws["zeros"] = torch.zeros(256, dtype=torch.uint8, device=self.device)
        rc = self.lib.psalm_conv2d_nhwc(self._p(x), B, H, W, Cin, self._p(wt), Cout, ksize, stride, pad, self._p(bias), self._pv(residual),
                                        c_long(residual.stride(0) if residual is not None else 0), self._p(out), _dt(out),
                                        c_long(out.stride(0)), act, self._p(z), self._p(self._gemm_ws()), c_long(self.GEMM_WS_BYTES),
                                        self._stream())
        self._check(rc, "psalm_conv2d_nhwc")
        return out

    def gemm_describe(self, M, N, K, a_bf16=True, w_bf16=True, x3=False):
        """(path, BM, BN, splits) psalm_gemm / psalm_gemm_x3 (x3=True, K = 3*Kp) uses for this problem (path 1 = direct-to-LDS kernel)."""
        out = (c_int * 4)()
        if x3:
            self._cdll_raw.psalm_gemm_describe(M, N, K, 2, 2, c_long(self.GEMM_WS_BYTES), out)
            return tuple(out)
        self._cdll_raw.psalm_gemm_describe(M, N, K, BF16 if a_bf16 else F32, BF16 if w_bf16 else F32, c_long(self.GEMM_WS_BYTES), out)
        return tuple(out)

    def gemm_last_kernel(self) -> str:
        """template instantiation of this thread's last direct-to-LDS GEMM launch, as a kernel trace spells it"""
        return self._cdll_raw.psalm_gemm_last_kernel().decode()

    # ------------------------------------------------------------------ stage-level entries (csrc/stages.hip)
    def phi_desc(self, layers, hidden, intermediate, heads, head_dim, rot, ln_eps, final_g, final_b):
        """psalm_phi_desc for psalm_phi_forward.  layers: per Phi layer a dict(w1=SplitF16, b1=, w2=SplitF16, b2=, ln_g=, ln_b=, bnd=, paired=bool)
        of device tensors.  The returned object keeps the host array (and, through `keep`, the tensors) alive."""
        arr = (_PhiLayer * len(layers))()
        keep = []
        for i, ly in enumerate(layers):
            w1, w2 = ly["w1"], ly["w2"]
            if not (isinstance(w1, SplitF16) and isinstance(w2, SplitF16)):
                raise PsalmHipError("phi_desc: split-f16 weights (precision 'f16x3')")
            arr[i] = _PhiLayer(self._p(w1.t), self._p(w1.inv_scale), self._p(ly["b1"]), self._p(w2.t), self._p(w2.inv_scale), self._p(ly["b2"]),
                               self._p(ly["ln_g"]), self._p(ly["ln_b"]), self._p(ly["bnd"]), int(bool(ly["paired"])))
            keep.append(ly)
        d = _PhiDesc(len(layers), hidden, intermediate, heads, head_dim, rot, ln_eps, ctypes.cast(arr, ctypes.POINTER(_PhiLayer)),
                     self._p(final_g), self._p(final_b))
        d._keep = (arr, keep, final_g, final_b)
        return d

    def swin_desc(self, w, depths, heads, dims, patch, window, pe_kpad, mlp_ratio, paired):
        """psalm_swin_desc from the model's weight table `w` (names as PSALM._prepare_weights lays them out: swin{s}.{b}.qkv.w ...)."""
        keep, stages = [], (_SwinStage * len(depths))()

        def sp(name):                                # split-f16 weight -> (rows pointer, scales pointer)
            t = w[name]
            if not isinstance(t, SplitF16):
                raise PsalmHipError(f"swin_desc: {name} is not in split-f16 form (precision 'f16x3')")
            return self._p(t.t), self._p(t.inv_scale)
        for s_, (depth, nh, dim) in enumerate(zip(depths, heads, dims)):
            blocks = (_SwinBlock * depth)()
            for b in range(depth):
                q = f"swin{s_}.{b}."
                qw, qs = sp(q + "qkv.w"); pw_, ps_ = sp(q + "proj.w"); f1w, f1s = sp(q + "fc1.w"); f2w, f2s = sp(q + "fc2.w")
                blocks[b] = _SwinBlock(self._p(w[q + "n1.g"]), self._p(w[q + "n1.b"]), self._p(w[q + "n2.g"]), self._p(w[q + "n2.b"]), qw, qs,
                                       self._p(w[q + "qkv.b"]), self._p(w[q + "qkv.bnd"]), pw_, ps_, self._p(w[q + "proj.b"]), f1w, f1s,
                                       self._p(w[q + "fc1.b"]), self._p(w[q + "fc1.bnd"]), int(bool(paired.get(q + "fc1", False))), f2w, f2s,
                                       self._p(w[q + "fc2.b"]), self._p(w[q + "rpb"]))
            keep.append(blocks)
            last = s_ == len(depths) - 1
            dw, dsc = (c_void_p(0), c_void_p(0)) if last else sp(f"swin{s_}.ds.red.w")
            stages[s_] = _SwinStage(depth, nh, dim, ctypes.cast(blocks, ctypes.POINTER(_SwinBlock)), self._p(w[f"swin.out{s_}.g"]),
                                    self._p(w[f"swin.out{s_}.b"]), self._p(None if last else w[f"swin{s_}.ds.ln.g"]),
                                    self._p(None if last else w[f"swin{s_}.ds.ln.b"]), dw, dsc)
        pw0, ps0 = sp("swin.pe.w")
        d = _SwinDesc(len(depths), patch, window, pe_kpad, mlp_ratio, pw0, ps0, self._p(w["swin.pe.b"]), self._p(w["swin.pe.ln.g"]),
                      self._p(w["swin.pe.ln.b"]), ctypes.cast(stages, ctypes.POINTER(_SwinStage)))
        d._keep = (keep, stages, w)
        d._dims = list(dims)
        return d

    def swin_forward(self, desc, images):
        """SwinTransformer.forward as ONE native call (psalm_swin_forward): images (B,3,H,W) float32 -> [(tokens (B*h*w, C) float32, h, w)] per stage."""
        B, _, H, W = images.shape
        self.lib.psalm_swin_forward_workspace.restype = c_long
        nbytes = self.lib.psalm_swin_forward_workspace(ctypes.byref(desc), B, H, W)
        if nbytes < 0:
            raise PsalmHipError(f"psalm_swin_forward_workspace: {self.lib.psalm_last_error().decode()}")
        ws = self._stage_ws("swin", nbytes + 256)
        off = (-ws.data_ptr()) % 256
        outs, hc, wc = [], (H + desc.patch - 1) // desc.patch, (W + desc.patch - 1) // desc.patch
        for C in desc._dims:
            outs.append((self.empty(B * hc * wc, C, dtype=torch.float32), hc, wc))
            hc, wc = (hc + 1) // 2, (wc + 1) // 2
        ptrs = (c_void_p * len(outs))(*[t.data_ptr() for t, _, _ in outs])
        rc = self.lib.psalm_swin_forward(ctypes.byref(desc), self._p(images), B, H, W, ptrs, c_void_p(ws.data_ptr() + off), c_long(nbytes),
                                         self._p(self._gemm_ws()), c_long(self.GEMM_WS_BYTES), self._stream())
        self._check(rc, "psalm_swin_forward")
        return outs

    def projector_desc(self, w):
        def sp(name):
            t = w[name]
            if not isinstance(t, SplitF16):
                raise PsalmHipError(f"projector_desc: {name} is not in split-f16 form")
            return self._p(t.t), self._p(t.inv_scale)
        c1, c2, c2f, ds, fc = sp("proj.c1.w"), sp("proj.c2.w"), sp("proj.c2f.w"), sp("proj.ds.w"), sp("proj.fc.w")
        mid, cin = w["proj.c1.w"].t.shape[0], w["proj.ds.w"].K
        d = _ProjDesc(cin, mid, w["proj.fc.w"].t.shape[0], c1[0], c1[1], self._p(w["proj.c1.b"]), c2[0], c2[1], c2f[0], c2f[1], self._p(w["proj.c2f.b"]),
                      ds[0], ds[1], self._p(w["proj.ds.b"]), fc[0], fc[1], self._p(w["proj.fc.b"]))
        d._keep = w
        return d

    def projector_forward(self, desc, res5, B, h, w_):
        """The conv projector as ONE native call (psalm_projector_forward): res5 (B*h*w, C) float32 -> ((B*ho*wo, hidden) float32, ho*wo)."""
        self.lib.psalm_projector_forward_workspace.restype = c_long
        nbytes = self.lib.psalm_projector_forward_workspace(ctypes.byref(desc), B, h, w_)
        if nbytes < 0:
            raise PsalmHipError("psalm_projector_forward_workspace: bad descriptor / geometry")
        ws = self._stage_ws("projector", nbytes + 256)
        off = (-ws.data_ptr()) % 256
        ho, wo = (h + 2 - 3) // 2 + 1, (w_ + 2 - 3) // 2 + 1
        out = self.empty(B * ho * wo, desc.out_dim, dtype=torch.float32)
        rc = self.lib.psalm_projector_forward(ctypes.byref(desc), self._p(res5), B, h, w_, self._p(out), c_void_p(ws.data_ptr() + off), c_long(nbytes),
                                              self._p(self._gemm_ws()), c_long(self.GEMM_WS_BYTES), self._stream())
        self._check(rc, "psalm_projector_forward")
        return out, ho * wo

    def pd_desc(self, w, D, G, M, num_layers, ffn, mask_dim, in_dims, paired):
        def sp(name):
            t = w[name]
            if not isinstance(t, SplitF16):
                raise PsalmHipError(f"pd_desc: {name} is not in split-f16 form")
            return t.t.data_ptr(), t.inv_scale.data_ptr()

        def P(name):
            return w[name].data_ptr()
        layers = (_PdEncLayer * num_layers)()
        for i in range(num_layers):
            q = f"pd.enc{i}."
            v, ow, ou, l1, l2 = sp(q + "value.w"), sp(q + "ow.w"), sp(q + "out.w"), sp(q + "l1.w"), sp(q + "l2.w")
            layers[i] = _PdEncLayer(v[0], v[1], P(q + "value.b"), ow[0], ow[1], P(q + "ow.b"), ou[0], ou[1], P(q + "out.b"), P(q + "n1.g"), P(q + "n1.b"),
                                    l1[0], l1[1], P(q + "l1.b"), P(q + "l1.bnd"), int(bool(paired.get(q + "l1", False))), l2[0], l2[1], P(q + "l2.b"),
                                    P(q + "n2.g"), P(q + "n2.b"))
        ips = [sp(f"pd.ip{i}.w") for i in range(3)]
        ad, la, mf = sp("pd.adapter.w"), sp("pd.layer.w"), sp("pd.mf.w")
        V3 = c_void_p * 3
        d = _PdDesc(D, G, M, num_layers, ffn, mask_dim, (c_int * 4)(*in_dims), V3(*[x[0] for x in ips]), V3(*[x[1] for x in ips]),
                    V3(*[P(f"pd.ip{i}.b") for i in range(3)]), V3(*[P(f"pd.ip{i}.gn.g") for i in range(3)]), V3(*[P(f"pd.ip{i}.gn.b") for i in range(3)]),
                    ad[0], ad[1], P("pd.adapter.b"), P("pd.adapter.gn.g"), P("pd.adapter.gn.b"), la[0], la[1], P("pd.layer.b"), P("pd.layer.gn.g"),
                    P("pd.layer.gn.b"), mf[0], mf[1], P("pd.mf.b"), ctypes.cast(layers, ctypes.POINTER(_PdEncLayer)))
        d._keep = (layers, w)
        return d

    def pixel_decoder_forward(self, desc, feats, lvl_pos):
        """MSDeformAttn pixel decoder of ONE image as ONE native call.  feats: [(tokens (h*w, C) float32 contiguous, h, w)] res2..res5.
        Returns (mask_features (h2*w2, mask_dim), ms (S, D) level-concatenated [res5 | res4 | res3])."""
        hw = (c_int * 8)(*[int(v) for _, h, w_ in feats for v in (h, w_)])
        self.lib.psalm_pixel_decoder_forward_workspace.restype = c_long
        nbytes = self.lib.psalm_pixel_decoder_forward_workspace(ctypes.byref(desc), hw)
        if nbytes < 0:
            raise PsalmHipError(f"psalm_pixel_decoder_forward_workspace: {self.lib.psalm_last_error().decode()}")
        ws = self._stage_ws("pixel_decoder", nbytes + 256)
        off = (-ws.data_ptr()) % 256
        for t, _, _ in feats:
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise PsalmHipError("pixel_decoder_forward: contiguous float32 feature tokens")
        S = sum(h * w_ for _, h, w_ in feats[1:])
        mf = self.empty(feats[0][1] * feats[0][2], desc.mask_dim, dtype=torch.float32)
        ms = self.empty(S, desc.D, dtype=torch.float32)
        ptrs = (c_void_p * 4)(*[t.data_ptr() for t, _, _ in feats])
        rc = self.lib.psalm_pixel_decoder_forward(ctypes.byref(desc), ptrs, hw, self._p(lvl_pos), self._p(mf), self._p(ms), c_void_p(ws.data_ptr() + off),
                                                  c_long(nbytes), self._p(self._gemm_ws()), c_long(self.GEMM_WS_BYTES), self._stream())
        self._check(rc, "psalm_pixel_decoder_forward")
        return mf, ms

    def pr_desc(self, w, D, heads, Q, num_layers, num_levels, ffn, mask_dim):
        def P(name):
            t = w[name]
            if isinstance(t, SplitF16) or t.dtype != torch.float32:
                raise PsalmHipError(f"pr_desc: {name} must be a plain float32 tensor")
            return t.data_ptr()
        layers = (_PrLayer * num_layers)()
        for i in range(num_layers):
            q = f"pr{i}."
            layers[i] = _PrLayer(*[P(q + n) for n in ("cq.w", "cq.b", "co.w", "co.b", "cn.g", "cn.b", "sqk.w", "sqk.b", "sv.w", "sv.b", "so.w", "so.b", "sn.g",
                                                      "sn.b", "f1.w", "f1.b", "f2.w", "f2.b", "fn.g", "fn.b")])
        V3, V2 = c_void_p * 3, c_void_p * 2
        ks = [w[f"pr.lvl{l}.k.w"] for l in range(num_levels)] + [None] * (3 - num_levels)
        vs = [w[f"pr.lvl{l}.v.w"] for l in range(num_levels)] + [None] * (3 - num_levels)
        if not all(t is None or isinstance(t, SplitF16) for t in ks + vs):
            raise PsalmHipError("pr_desc: the level K / V projection weights must be in split-f16 form")
        pb = lambda t, f: 0 if t is None else f(t)               # noqa: E731
        d = _PrDesc(D, heads, Q, num_layers, num_levels, ffn, mask_dim,
                    V3(*[pb(t, lambda x: x.t.data_ptr()) for t in ks]), V3(*[pb(t, lambda x: x.inv_scale.data_ptr()) for t in ks]),
                    V3(*[P(f"pr.lvl{l}.k.b") if l < num_levels else 0 for l in range(3)]),
                    V3(*[pb(t, lambda x: x.t.data_ptr()) for t in vs]), V3(*[pb(t, lambda x: x.inv_scale.data_ptr()) for t in vs]),
                    V3(*[P(f"pr.lvl{l}.v.b") if l < num_levels else 0 for l in range(3)]),
                    P("pr.level_embed"), P("pr.query_embed"), P("pr.dn.g"), P("pr.dn.b"),
                    V3(*[P(f"pr.mask_embed{j}.w") for j in range(3)]), V3(*[P(f"pr.mask_embed{j}.b") for j in range(3)]),
                    V2(*[P(f"pr.SEG_proj{j}.w") for j in range(2)]), V2(*[P(f"pr.SEG_proj{j}.b") for j in range(2)]),
                    V2(*[P(f"pr.CLASS_proj{j}.w") for j in range(2)]), V2(*[P(f"pr.CLASS_proj{j}.b") for j in range(2)]),
                    V2(*[P(f"pr.REGION_proj{j}.w") for j in range(2)]), V2(*[P(f"pr.REGION_proj{j}.b") for j in range(2)]),
                    ctypes.cast(layers, ctypes.POINTER(_PrLayer)))
        d._keep = (layers, w)
        return d

    def predictor_kv(self, desc, ms, shapes, prpos, mf, mf_size, n_reg=0, slot=0):
        """The LLM-independent front of the masked-attention decoder (psalm_predictor_kv: level K / V projections + mask-feature split) into a workspace
        that `predictor_forward(..., kv=<the returned handle>)` then continues in.  `slot`: which of the caller's images this is (each keeps its own
        workspace until its predictor call)."""
        H2, W2 = mf_size
        hw = (c_int * (2 * len(shapes)))(*[int(v) for s_ in shapes for v in s_])
        self.lib.psalm_predictor_forward_workspace.restype = c_long
        nbytes = self.lib.psalm_predictor_forward_workspace(ctypes.byref(desc), hw, H2, W2, int(n_reg))
        if nbytes < 0:
            raise PsalmHipError(f"psalm_predictor_forward_workspace: {self.lib.psalm_last_error().decode()}")
        ws = self._stage_ws(f"predictor{slot}", nbytes + 256)
        off = (-ws.data_ptr()) % 256
        VP = c_void_p * len(shapes)
        rc = self.lib.psalm_predictor_kv(ctypes.byref(desc), VP(*[t.data_ptr() for t in ms]), hw, VP(*[t.data_ptr() for t in prpos]), self._p(mf), H2, W2, int(n_reg),
                                         c_void_p(ws.data_ptr() + off), c_long(nbytes), self._p(self._gemm_ws()), c_long(self.GEMM_WS_BYTES), self._stream())
        self._check(rc, "psalm_predictor_kv")
        return (ws, off, nbytes, int(n_reg))

    def predictor_forward(self, desc, ms, shapes, prpos, mf, mf_size, seg_query, class_emb=None, seg_emb=None, region_emb=None, kv=None):
        """The masked-attention decoder of ONE image as ONE native call.  ms / prpos: per-level (h*w, D) float32 tensors; mf (H2*W2, mask_dim);
        seg_query (Q, D); *_emb (n, D) float32 or None.  Returns (pred_masks (Q, H2*W2), cls_logits | None, seg_logits | None, region_logits | None)."""
        H2, W2 = mf_size
        Q, D = desc.Q, desc.D
        for t in list(ms) + list(prpos) + [mf, seg_query] + [e for e in (class_emb, seg_emb, region_emb) if e is not None]:
            if t.dtype != torch.float32 or not t.is_contiguous() or t.data_ptr() % 16:
                raise PsalmHipError("predictor_forward: contiguous, 16-byte aligned float32 tensors")
        hw = (c_int * (2 * len(shapes)))(*[int(v) for s_ in shapes for v in s_])
        n_reg = int(region_emb.shape[0]) if region_emb is not None else 0
        self.lib.psalm_predictor_forward_workspace.restype = c_long
        nbytes = self.lib.psalm_predictor_forward_workspace(ctypes.byref(desc), hw, H2, W2, n_reg)
        if nbytes < 0:
            raise PsalmHipError(f"psalm_predictor_forward_workspace: {self.lib.psalm_last_error().decode()}")
        if kv is not None:
            ws, off, kv_bytes, kv_reg = kv
            if kv_bytes != nbytes or kv_reg != n_reg:
                raise PsalmHipError("predictor_forward: the K / V workspace was prepared for another geometry / region count")
        else:
            ws = self._stage_ws("predictor", nbytes + 256)
            off = (-ws.data_ptr()) % 256
        masks = self.empty(Q, H2 * W2, dtype=torch.float32)
        cls = self.empty(Q, class_emb.shape[0], dtype=torch.float32) if class_emb is not None else None
        seg = self.empty(Q, seg_emb.shape[0], dtype=torch.float32) if seg_emb is not None else None
        reg = self.empty(n_reg, Q, dtype=torch.float32) if region_emb is not None else None
        VP = c_void_p * len(shapes)
        rc = self.lib.psalm_predictor_forward(ctypes.byref(desc), VP(*[t.data_ptr() for t in ms]), hw, VP(*[t.data_ptr() for t in prpos]), self._p(mf), H2, W2,
                                              self._p(seg_query), self._p(class_emb), int(class_emb.shape[0]) if class_emb is not None else 0, self._p(seg_emb),
                                              int(seg_emb.shape[0]) if seg_emb is not None else 0, self._p(region_emb), n_reg, self._p(masks), self._p(cls),
                                              self._p(seg), self._p(reg), c_void_p(ws.data_ptr() + off), c_long(nbytes), self._p(self._gemm_ws()),
                                              c_long(self.GEMM_WS_BYTES), 1 if kv is not None else 0, self._stream())
        self._check(rc, "psalm_predictor_forward")
        return masks, cls, seg, reg

    POST_TASKS = {"semantic": 0, "instance": 1, "panoptic": 2, "referring": 3, "region": 4}

    def postprocess(self, task, sizes, pred_masks=None, mask_up=None, cls_logits=None, seg_logits=None, region_logits=None, is_thing=None,
                    obj_thr=0.8, overlap_thr=0.8):
        """llava_phi.py:1401-1466 for one image as ONE native call (psalm_postprocess_<task>): sizes = (Hpad, Wpad, crop_h, crop_w, out_h, out_w);
        pred_masks (Q,h,w) and / or mask_up (Q,Hpad,Wpad) float32.  Returns a dict of the result tensors at their maximum sizes (the caller slices by
        `counts` after its one read-back): mask_pred, sem_seg, scores, classes, query, inst_masks, boxes, pan, counts -- those the task has."""
        Hpad, Wpad, oh, ow, height, width = [int(v) for v in sizes]
        src = pred_masks if pred_masks is not None else mask_up
        Q = int(src.shape[0])
        h_, w_ = (int(pred_masks.shape[1]), int(pred_masks.shape[2])) if pred_masks is not None else (Hpad, Wpad)
        C1 = int(cls_logits.shape[1]) if cls_logits is not None else 0
        k = int(region_logits.shape[0]) if region_logits is not None else 0
        code = self.POST_TASKS[task]
        d = _PostDesc(code, Q, h_, w_, Hpad, Wpad, oh, ow, height, width, C1, k, float(obj_thr), float(overlap_thr))
        self.lib.psalm_postprocess_workspace.restype = c_long
        nbytes = self.lib.psalm_postprocess_workspace(ctypes.byref(d), 1 if mask_up is not None else 0)
        if nbytes < 0:
            raise PsalmHipError(f"psalm_postprocess_workspace: {self.lib.psalm_last_error().decode()}")
        ws = self._stage_ws("postprocess", nbytes + 256)
        off = (-ws.data_ptr()) % 256
        resize_after = (oh, ow, height, width) != (Hpad, Wpad, Hpad, Wpad)
        out = {}
        if task == "semantic":
            if mask_up is None:
                out["mask_pred"] = self.empty(Q, Hpad, Wpad)
            out["sem_seg"] = self.empty(C1 - 1, height, width)
        else:
            if resize_after or mask_up is None:
                out["mask_pred"] = self.empty(Q, height, width)
            out["scores"] = self.empty(Q, k) if task == "region" else self.empty(Q)
            out["inst_masks"] = self.empty(Q, height, width)
            out["boxes"] = self.empty(Q, 4)
            out["counts"] = self.empty(2 + 3 * Q, dtype=torch.int32)
            if task in ("instance", "panoptic"):
                out["classes"] = self.empty(Q, dtype=torch.int64)
            if task != "region":
                out["query"] = self.empty(Q, dtype=torch.int64)
            if task == "panoptic":
                out["sem_seg"] = self.empty(C1 - 1, height, width)
                out["pan"] = self.empty(height, width, dtype=torch.int32)
        io = _PostIO(*[self._pi(t) for t in (pred_masks, mask_up, cls_logits, seg_logits, region_logits, is_thing, out.get("mask_pred"), out.get("sem_seg"),
                                             out.get("scores"), out.get("classes"), out.get("query"), out.get("inst_masks"), out.get("boxes"), out.get("pan"),
                                             out.get("counts"))])
        which = c_int(0)
        fn = getattr(self.lib, "psalm_postprocess_" + task)
        rc = fn(ctypes.byref(d), ctypes.byref(io), ctypes.byref(which), c_void_p(ws.data_ptr() + off), c_long(nbytes), self._stream())
        self._check(rc, "psalm_postprocess_" + task)
        if which.value == 1:
            out["mask_pred"] = mask_up
        return out

    @staticmethod
    def _pi(t):
        return None if t is None else t.data_ptr()

    def phi_forward(self, desc, embeds, key_mask, cos, sin, B, L):
        """PhiModel.forward over inputs_embeds (B*L, hidden) float32 as ONE native call (psalm_phi_forward): returns the final-LayerNorm hidden
        states (B*L, hidden) float32.  Same launches, same order, same bits as PSALM.llm's op-by-op sequence."""
        if embeds.dtype != torch.float32 or embeds.dim() != 2 or embeds.shape[0] != B * L or embeds.shape[1] != desc.hidden:
            raise PsalmHipError("phi_forward: float32 (B*L, hidden) embeddings")
        self.lib.psalm_phi_forward_workspace.restype = c_long
        nbytes = self.lib.psalm_phi_forward_workspace(ctypes.byref(desc), B, L)
        if nbytes < 0:
            raise PsalmHipError(f"psalm_phi_forward_workspace: {self.lib.psalm_last_error().decode()}")
        ws = self._stage_ws("phi", nbytes + 256)
        off = (-ws.data_ptr()) % 256
        out = self.empty(B * L, desc.hidden, dtype=torch.float32)
        rc = self.lib.psalm_phi_forward(ctypes.byref(desc), self._p(embeds), self._p(key_mask), self._p(cos), self._p(sin), B, L, self._p(out),
                                        c_void_p(ws.data_ptr() + off), c_long(nbytes), self._p(self._gemm_ws()), c_long(self.GEMM_WS_BYTES),
                                        self._stream())
        self._check(rc, "psalm_phi_forward")
        return out

    def x3_products(self, n: int):
        """f16 products formed per algorithmic product by this thread's split-f16 GEMMs: 3 (default, fp32-class) or 1 (hi.hi only: plain f16
        operands, a third of the matrix work -- the reduced-precision LLM side mode, not at the parity bar).  See psalm_gemm_x3_set_products."""
        self._check(self.lib.psalm_gemm_x3_set_products(int(n)), "psalm_gemm_x3_set_products")

    TUNE_GEMM_XCD_KSPLIT, TUNE_ATTN_XCD_HEADS, TUNE_GEMM_MID, TUNE_DECODER_FUSE, TUNE_ROW_GROUPS = 0, 1, 2, 3, 4      # PSALM_TUNE_* of include/psalm_hip.h

    def set_tuning(self, key: int, value: int):
        """psalm_set_tuning: process-wide atomic switches between kernel forms with identical results (A/B runs, tests)."""
        self._check(self.lib.psalm_set_tuning(int(key), int(value)), "psalm_set_tuning")

    def get_tuning(self, key: int) -> int:
        return int(self.lib.psalm_get_tuning(int(key)))

    def gemm_tile_policy(self, bm: int):
        """0 = automatic, 256 / 128 / 64 = force the direct-to-LDS kernel's tile height (tuning / tests)."""
        self._check(self.lib.psalm_gemm_set_tile_policy(bm), "psalm_gemm_set_tile_policy")

    # ------------------------------------------------------------------ row ops
    def layernorm(self, x, gamma, beta, eps=1e-5, out=None, out_dtype=None, out2=None, add=None, out3=None):
        """LayerNorm over the last dim of a 2-D (row-strided) view.  out2: optional bf16 (rows,C) second copy;
        out3: optional bf16 (rows,C) = result + add[row % add.shape[0]] (add fp32 (r,C))."""
        rows, C = x.shape
        if out is None:
            out = self.empty(rows, C, dtype=out_dtype or x.dtype)
        for t in (out2, out3):
            if t is not None and t.dtype != torch.bfloat16:
                raise PsalmHipError("layernorm: out2 / out3 must be bfloat16")
        if out3 is not None and (add is None or add.dtype != torch.float32 or add.shape[1] != C):
            raise PsalmHipError("layernorm: out3 needs a float32 (r,C) `add` table")
        rc = self.lib.psalm_layernorm3(self._pv(x), _dt(x), c_long(x.stride(0)), self._pv(out), _dt(out), c_long(out.stride(0)),
                                       self._pv(out2), c_long(out2.stride(0) if out2 is not None else 0),
                                       self._p(add) if out3 is not None else c_void_p(0), c_long(add.shape[0] if out3 is not None else 0),
                                       self._pv(out3), c_long(out3.stride(0) if out3 is not None else 0),
                                       self._p(gamma), self._p(beta), rows, C, c_float(eps), self._stream())
        self._check(rc, "psalm_layernorm3")
        return out

    def layernorm_split(self, x, gamma, beta, eps=1e-5, want_y=False, want_split=True, add=None):
        """LayerNorm of float32 rows whose result leaves as the next GEMM's split-f16 A operand (f16x3 mode).
        Returns (y float32 | None, SplitF16(y) | None, SplitF16(y + add[row % r]) | None)."""
        rows, C = x.shape
        if x.dtype != torch.float32 or x.stride(1) != 1:
            raise PsalmHipError("layernorm_split: float32 rows")
        Kp = (C + 63) // 64 * 64
        y = self.empty(rows, C, dtype=torch.float32) if want_y else None
        s1 = i1 = s2 = i2 = None
        if want_split:
            s1, i1 = self.empty(rows, 2 * Kp, dtype=torch.float16), self.empty(rows, dtype=torch.float32)
        if add is not None:
            if add.dtype != torch.float32 or add.shape[1] != C:
                raise PsalmHipError("layernorm_split: float32 (r,C) `add` table")
            s2, i2 = self.empty(rows, 2 * Kp, dtype=torch.float16), self.empty(rows, dtype=torch.float32)
        rc = self.lib.psalm_layernorm_split(self._pv(x), c_long(x.stride(0)), self._p(y), c_long(C), self._p(gamma), self._p(beta), rows, C,
                                            c_float(eps), self._p(s1), self._p(i1), self._p(add), c_long(add.shape[0] if add is not None else 0),
                                            self._p(s2), self._p(i2), self._stream())
        self._check(rc, "psalm_layernorm_split")
        return y, (SplitF16(s1, i1, C) if want_split else None), (SplitF16(s2, i2, C) if add is not None else None)

    def swin_window_gather(self, x, gamma, beta, B, H, W, ws, shift, eps=1e-5, out_dtype=None):
        """x (B*H*W, C) -> LN + pad + roll(-shift) + window partition -> (B*nW*ws*ws, C)."""
        C = x.shape[-1]
        nWh, nWw = (H + ws - 1) // ws, (W + ws - 1) // ws
        out = self.empty(B * nWh * nWw * ws * ws, C, dtype=out_dtype or x.dtype)
        rc = self.lib.psalm_swin_window_gather(self._p(x), _dt(x), self._p(out), _dt(out), self._p(gamma), self._p(beta), B, H, W,
                                               C, ws, shift, c_float(eps), self._stream())
        self._check(rc, "psalm_swin_window_gather")
        return out

    def swin_window_gather_split(self, x, gamma, beta, B, H, W, ws, shift, eps=1e-5):
        """swin_window_gather whose rows leave as the qkv GEMM's split-f16 A operand (f16x3 mode): x (B*H*W, C) float32 -> SplitF16."""
        C = x.shape[-1]
        if x.dtype != torch.float32 or C % 8 or C > 2048:
            raise PsalmHipError("swin_window_gather_split: float32, C % 8 == 0, C <= 2048")
        nWh, nWw = (H + ws - 1) // ws, (W + ws - 1) // ws
        rows, Kp = B * nWh * nWw * ws * ws, (C + 63) // 64 * 64
        t = self.empty(rows, 2 * Kp, dtype=torch.float16)
        inv = self.empty(rows, dtype=torch.float32)
        rc = self.lib.psalm_swin_window_gather_split(self._p(x), self._p(t), self._p(inv), self._p(gamma), self._p(beta), B, H, W, C, ws, shift,
                                                     c_float(eps), self._stream())
        self._check(rc, "psalm_swin_window_gather_split")
        return SplitF16(t, inv, C)

    def swin_window_merge_ln_split(self, win, shortcut, gamma, beta, B, H, W, ws, shift, eps=1e-5):
        """swin_window_merge_ln with norm2's result as the fc1 GEMM's split-f16 A operand: (x_new float32, SplitF16(LayerNorm(x_new)))."""
        C = shortcut.shape[-1]
        if shortcut.dtype != torch.float32 or win.dtype != torch.float32 or C % 8 or C > 2048:
            raise PsalmHipError("swin_window_merge_ln_split: float32 operands, C % 8 == 0, C <= 2048")
        rows, Kp = shortcut.shape[0], (C + 63) // 64 * 64
        out_x = torch.empty_like(shortcut)
        t = self.empty(rows, 2 * Kp, dtype=torch.float16)
        inv = self.empty(rows, dtype=torch.float32)
        rc = self.lib.psalm_swin_window_merge_ln_split(self._p(win), self._p(shortcut), self._p(out_x), self._p(t), self._p(inv), self._p(gamma),
                                                       self._p(beta), B, H, W, C, ws, shift, c_float(eps), self._stream())
        self._check(rc, "psalm_swin_window_merge_ln_split")
        return out_x, SplitF16(t, inv, C)

    def swin_window_merge(self, win, shortcut, B, H, W, ws, shift, out=None):
        """out (B*H*W, C) = shortcut + window_reverse/roll(+shift)/crop(win)."""
        C = shortcut.shape[-1]
        if out is None:
            out = torch.empty_like(shortcut)
        rc = self.lib.psalm_swin_window_merge(self._p(win), _dt(win), self._p(shortcut), self._p(out), _dt(shortcut), B, H, W, C,
                                              ws, shift, self._stream())
        self._check(rc, "psalm_swin_window_merge")
        return out

    def swin_window_merge_ln(self, win, shortcut, gamma, beta, B, H, W, ws, shift, eps=1e-5, h_dtype=None):
        """(x_new fp32, LayerNorm(x_new) h_dtype): window_reverse/roll/crop + residual fused with the block's norm2."""
        C = shortcut.shape[-1]
        if shortcut.dtype != torch.float32:
            raise PsalmHipError("swin_window_merge_ln: the residual stream must be float32")
        out_x = torch.empty_like(shortcut)
        out_h = self.empty(shortcut.shape[0], C, dtype=h_dtype or win.dtype)
        rc = self.lib.psalm_swin_window_merge_ln(self._p(win), _dt(win), self._p(shortcut), self._p(out_x), self._p(out_h), _dt(out_h),
                                                 self._p(gamma), self._p(beta), B, H, W, C, ws, shift, c_float(eps), self._stream())
        self._check(rc, "psalm_swin_window_merge_ln")
        return out_x, out_h

    def patch_merge_ln(self, x, gamma, beta, B, H, W, eps=1e-5, out_dtype=None):
        C = x.shape[-1]
        out = self.empty(B * ((H + 1) // 2) * ((W + 1) // 2), 4 * C, dtype=out_dtype or x.dtype)
        rc = self.lib.psalm_patch_merge_ln(self._p(x), _dt(x), self._p(out), _dt(out), self._p(gamma), self._p(beta), B, H, W, C,
                                           c_float(eps), self._stream())
        self._check(rc, "psalm_patch_merge_ln")
        return out

    def groupnorm_nhwc(self, x, gamma, beta, B, HW, groups, eps=1e-5, relu=False, out_dtype=None, out=None):
        """x (B*HW, C) NHWC -> GroupNorm(groups) [+ReLU]."""
        C = x.shape[-1]
        if out is None:
            out = self.empty(B * HW, C, dtype=out_dtype or x.dtype)
        ws = self.empty(B * ((HW + 63) // 64 + 1) * groups * 2, dtype=torch.float32)
        rc = self.lib.psalm_groupnorm_nhwc(self._p(x), _dt(x), self._p(out), _dt(out), self._p(gamma), self._p(beta), self._p(ws),
                                           B, HW, C, groups, c_float(eps), int(relu), self._stream())
        self._check(rc, "psalm_groupnorm_nhwc")
        return out

    def layernorm_chain(self, x, g1, b1, add=None, g2=None, b2=None, eps=1e-5):
        """(y1, y2 | None, y3 | None) = (LN(x; g1, b1), y1 + add[row % add_rows], LN(y1; g2, b2)) in one launch (psalm_layernorm_chain, fp32)."""
        rows, C = x.shape
        y1 = self.empty(rows, C, dtype=torch.float32)
        y2 = self.empty(rows, C, dtype=torch.float32) if add is not None else None
        y3 = self.empty(rows, C, dtype=torch.float32) if g2 is not None else None
        rc = self.lib.psalm_layernorm_chain(self._p(x), c_long(x.stride(0)), self._p(y1), c_long(C), self._p(g1), self._p(b1), self._pv(add),
                                            c_long(add.shape[0] if add is not None else 0), self._pv(y2), c_long(C), self._pv(g2), self._pv(b2),
                                            self._pv(y3), c_long(C), rows, C, c_float(eps), self._stream())
        self._check(rc, "psalm_layernorm_chain")
        return y1, y2, y3

    def gemm_f32_pair(self, a0, w0, bias0, a1, w1, bias1, act0=ACT_NONE, act1=ACT_NONE):
        """two exact-fp32 skinny GEMMs in one launch (psalm_gemm_f32_pair): returns (a0 . w0^T + bias0, a1 . w1^T + bias1)"""
        c0 = self.empty(a0.shape[0], w0.shape[0], dtype=torch.float32)
        c1 = self.empty(a1.shape[0], w1.shape[0], dtype=torch.float32)
        rc = self.lib.psalm_gemm_f32_pair(self._p(a0), self._p(w0), self._pv(bias0), self._p(c0), a0.shape[0], w0.shape[0], a0.shape[1], act0,
                                          self._p(a1), self._p(w1), self._pv(bias1), self._p(c1), a1.shape[0], w1.shape[0], a1.shape[1], act1, self._stream())
        self._check(rc, "psalm_gemm_f32_pair")
        return c0, c1

    def add_bcast(self, a, b, out_dtype=None):
        """out[r] = a[r] + b[r % b_rows]   (a (rows,C), b (b_rows,C))."""
        rows, C = a.shape
        out = self.empty(rows, C, dtype=out_dtype or a.dtype)
        rc = self.lib.psalm_add_bcast(self._p(a), _dt(a), self._p(b), _dt(b), self._p(out), _dt(out), c_long(rows), C,
                                      c_long(b.shape[0]), self._stream())
        self._check(rc, "psalm_add_bcast")
        return out

    def gather_rows(self, srcs, src_id, src_row, C, out_dtype=torch.float32):
        """dst[r] = srcs[src_id[r]][src_row[r]]; src_id < 0 -> zero row.  src_id/src_row int32 device tensors."""
        srcs = list(srcs) + [None] * (4 - len(srcs))
        rows = src_id.numel()
        out = self.empty(rows, C, dtype=out_dtype)
        args = []
        for t in srcs:
            args += [self._p(t), _dt(t) if t is not None else 0]
        rc = self.lib.psalm_gather_rows(*args, self._p(src_id), self._p(src_row), self._p(out), _dt(out), c_long(rows), C,
                                        self._stream())
        self._check(rc, "psalm_gather_rows")
        return out

    def segment_mean(self, x, seg_offsets, seg_rows, out_dtype=None):
        """out[s] = mean of x[seg_rows[seg_offsets[s]:seg_offsets[s+1]]]."""
        nseg = seg_offsets.numel() - 1
        C = x.shape[-1]
        out = self.empty(nseg, C, dtype=out_dtype or x.dtype)
        rc = self.lib.psalm_segment_mean(self._pv(x), _dt(x), c_long(x.stride(0)), self._p(seg_offsets), self._p(seg_rows),
                                         self._p(out), _dt(out), nseg, C, self._stream())
        self._check(rc, "psalm_segment_mean")
        return out

    # ------------------------------------------------------------------ attention
    def window_attention(self, qkv, bias_table, B, nWh, nWw, heads, ws, shift):
        C = qkv.shape[-1] // 3
        out = self.empty(qkv.shape[0], C, dtype=qkv.dtype)
        if qkv.dtype == torch.bfloat16 and ws == 12:          # matrix-core kernel (bf16 mode)
            rc = self.lib.psalm_window_attention_mfma(self._p(qkv), self._p(bias_table), self._p(out), B, nWh, nWw, C, heads, ws,
                                                      shift, self._stream())
            self._check(rc, "psalm_window_attention_mfma")
            return out
        rc = self.lib.psalm_window_attention(self._p(qkv), self._p(bias_table), self._p(out), _dt(qkv), B, nWh, nWw, C, heads, ws,
                                             shift, self._stream())
        self._check(rc, "psalm_window_attention")
        return out

    def window_attention_split(self, qkv, bias_table, a_inv, bound_par, B, nWh, nWw, heads, ws, shift):
        """window_attention on a float32 qkv buffer (12 x 12 windows) -> SplitF16 (rows, C): the projection GEMM's A operand, see
        psalm_window_attention_split.  a_inv: row scales of the qkv GEMM's A operand; bound_par: >= 2 device floats."""
        C = qkv.shape[-1] // 3
        if qkv.dtype != torch.float32 or ws != 12 or a_inv.numel() != qkv.shape[0]:
            raise PsalmHipError("window_attention_split: float32 qkv, 12 x 12 windows, one operand scale per row")
        Kp = (C + 63) // 64 * 64
        so = (self.empty if Kp == C else self.zeros)(qkv.shape[0], 2 * Kp, dtype=torch.float16)
        inv = self.empty(qkv.shape[0], dtype=torch.float32)
        rc = self.lib.psalm_window_attention_split(self._p(qkv), self._p(bias_table), self._p(a_inv), self._p(bound_par), self._p(so), Kp,
                                                   self._p(inv), B, nWh, nWw, C, heads, ws, shift, self._stream())
        self._check(rc, "psalm_window_attention_split")
        return SplitF16(so, inv, C)

    def causal_attention(self, buf, q_off, k_off, v_off, out, o_off, cos, sin, key_mask, B, L, heads, head_dim, rot):
        """buf (B*L, ld) holds q|k|v column blocks; out (B*L, ldo) receives the attention output at column o_off.
        bf16 buffers run on the matrix cores (psalm_causal_attention_mfma); fp32 buffers on the exact fp32 kernel."""
        if buf.dtype != out.dtype:
            raise PsalmHipError("causal_attention: buf/out dtype mismatch")
        if buf.dtype == torch.bfloat16:
            self.lib.psalm_causal_attention_mfma_workspace.restype = c_long
            nbytes = self.lib.psalm_causal_attention_mfma_workspace(B, L, heads)
            key = ("causal_ws", nbytes)
            ws = self._ws.get(key)
            if ws is None:
                ws = self._ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            rc = self.lib.psalm_causal_attention_mfma(self._pv(buf), c_long(buf.stride(0)), q_off, k_off, v_off, self._pv(out),
                                                      c_long(out.stride(0)), o_off, self._p(cos), self._p(sin), self._p(key_mask),
                                                      self._p(ws), B, L, heads, head_dim, rot, self._stream())
            self._check(rc, "psalm_causal_attention_mfma")
            return out
        if (buf.dtype == torch.float32 and head_dim == 64 and rot == 32 and buf.stride(0) % 4 == 0 and out.stride(0) % 4 == 0
                and all(v % 4 == 0 for v in (q_off, k_off, v_off, o_off)) and buf.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0):
            self.lib.psalm_causal_attention_f32_workspace.restype = c_long      # fp32 matrix-core kernel, key-split inside the block
            nbytes = self.lib.psalm_causal_attention_f32_workspace(B, L, heads)
            key = ("causal_f32_ws", nbytes)
            ws = self._ws.get(key)
            if ws is None:
                ws = self._ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            rc = self.lib.psalm_causal_attention_f32(self._pv(buf), c_long(buf.stride(0)), q_off, k_off, v_off, self._pv(out),
                                                     c_long(out.stride(0)), o_off, self._p(cos), self._p(sin), self._p(key_mask), self._p(ws),
                                                     B, L, heads, head_dim, rot, self._stream())
            self._check(rc, "psalm_causal_attention_f32")
            return out
        rc = self.lib.psalm_causal_attention(self._pv(buf), _dt(buf), c_long(buf.stride(0)), q_off, k_off, v_off, self._pv(out),
                                             c_long(out.stride(0)), o_off, self._p(cos), self._p(sin), self._p(key_mask), B, L,
                                             heads, head_dim, rot, self._stream())
        self._check(rc, "psalm_causal_attention")
        return out

    def causal_attention_split(self, buf, q_off, k_off, v_off, split_out, split_inv, split_col_off, cos, sin, key_mask, B, L, heads,
                               head_dim, rot):
        """causal_attention on an fp32 buffer whose output goes, in split-f16 form under the row scales 1/split_inv, into columns
        split_col_off.. of `split_out` ((B*L, 2*Kp) float16; lo part Kp columns further) -- see psalm_causal_attention_f32_split."""
        if buf.dtype != torch.float32 or split_out.dtype != torch.float16 or split_inv.dtype != torch.float32:
            raise PsalmHipError("causal_attention_split: float32 qkv buffer, float16 split buffer, float32 scales")
        self.lib.psalm_causal_attention_f32_workspace.restype = c_long
        nbytes = self.lib.psalm_causal_attention_f32_workspace(B, L, heads)
        key = ("causal_f32_ws", nbytes)
        ws = self._ws.get(key)
        if ws is None:
            ws = self._ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        rc = self.lib.psalm_causal_attention_f32_split(self._pv(buf), c_long(buf.stride(0)), q_off, k_off, v_off, self._p(split_out),
                                                       c_long(split_out.stride(0)), split_out.shape[1] // 2, split_col_off,
                                                       self._p(split_inv), self._p(cos), self._p(sin), self._p(key_mask), self._p(ws),
                                                       B, L, heads, head_dim, rot, self._stream())
        self._check(rc, "psalm_causal_attention_f32_split")
        return split_out

    def mha_attention(self, q, k, v, B, Lq, Lk, heads, mask=None, row_all_masked=None):
        """q (B*Lq, D) / k, v (B*Lk, D) row-strided views, head_dim 32; mask (B,Lq,Lk) u8 1 = blocked."""
        D = heads * 32
        out = self.empty(B * Lq, D, dtype=q.dtype)
        if q.dtype == torch.float32 and Lq <= 128 and all(t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0 for t in (q, k, v)):
            self.lib.psalm_mha_attention_f32_workspace.restype = c_long          # fp32 matrix-core kernel, split over 256-key chunks
            nbytes = self.lib.psalm_mha_attention_f32_workspace(B, heads, Lq, Lk)
            ws = None
            if nbytes:
                key = ("mha_f32_ws", nbytes)
                ws = self._ws.get(key)
                if ws is None:
                    ws = self._ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            rc = self.lib.psalm_mha_attention_f32(self._pv(q), c_long(q.stride(0)), self._pv(k), c_long(k.stride(0)), self._pv(v),
                                                  c_long(v.stride(0)), self._p(out), c_long(D), self._p(mask), self._p(row_all_masked),
                                                  self._p(ws), B, Lq, Lk, heads, 32, self._stream())
            self._check(rc, "psalm_mha_attention_f32")
            return out
        rc = self.lib.psalm_mha_attention(self._pv(q), c_long(q.stride(0)), self._pv(k), c_long(k.stride(0)), self._pv(v),
                                          c_long(v.stride(0)), self._p(out), c_long(D), _dt(q), self._p(mask),
                                          self._p(row_all_masked), B, Lq, Lk, heads, 32, self._stream())
        self._check(rc, "psalm_mha_attention")
        return out

    def mha_attention_t(self, q, k, vt, B, Lq, Lk, heads, mask=None, row_all_masked=None):
        """Matrix-core split-KV form: q (B*Lq, D) / k (B*Lk, D) row-strided bf16 views; vt (B*D, ldvt) = V transposed
        (row h*32+d, columns = keys, zero padded).  mask (B,Lq,Lk) u8 1 = blocked."""
        D = heads * 32
        out = self.empty(B * Lq, D, dtype=torch.bfloat16)
        self.lib.psalm_mha_attention_mfma_workspace.restype = c_long
        nbytes = self.lib.psalm_mha_attention_mfma_workspace(B, heads, Lk)
        ws = None
        if nbytes:
            key = ("mha_ws", nbytes)
            ws = self._ws.get(key)
            if ws is None:
                ws = self._ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        rc = self.lib.psalm_mha_attention_mfma(self._pv(q), c_long(q.stride(0)), self._pv(k), c_long(k.stride(0)), self._pv(vt),
                                               c_long(vt.stride(0)), self._p(out), c_long(D), self._p(mask), self._p(row_all_masked),
                                               self._p(ws), B, Lq, Lk, heads, 32, self._stream())
        self._check(rc, "psalm_mha_attention_mfma")
        return out

    def attn_mask(self, masks, Ht, Wt):
        """masks (B,Q,h,w) f32 logits -> (u8 (B,Q,Ht*Wt) 1 = blocked, u8 (B,Q) all-masked flags)."""
        B, Q, h, w = masks.shape
        out = self.empty(B, Q, Ht * Wt, dtype=torch.uint8)
        flags = self.empty(B, Q, dtype=torch.uint8)
        rc = self.lib.psalm_attn_mask(self._p(masks), self._p(out), self._p(flags), B * Q, h, w, Ht, Wt, self._stream())
        self._check(rc, "psalm_attn_mask")
        return out, flags

    # ------------------------------------------------------------------ image / layout ops
    def patch_im2col(self, img, ps, Kpad, out_dtype=torch.float32):
        B, Cin, H, W = img.shape
        Hp, Wp = (H + ps - 1) // ps, (W + ps - 1) // ps
        out = self.empty(B * Hp * Wp, Kpad, dtype=out_dtype)
        rc = self.lib.psalm_patch_im2col(self._p(img), self._p(out), _dt(out), B, Cin, H, W, ps, Kpad, self._stream())
        self._check(rc, "psalm_patch_im2col")
        return out

    def image_preprocess(self, img, nh, nw, S, mean, std):
        """img (H,W,3) uint8 RGB on the device -> (image (3,S,S) float32 normalised, padding_mask (S,S) bool): Pillow-bilinear resize to
        (nh, nw) (bit-identical to PIL / detectron2 ResizeShortestEdge), pad bottom/right with 128, (x - mean) / std."""
        from .preprocess import pil_bilinear_tables
        if img.dim() != 3 or img.shape[2] != 3 or img.dtype != torch.uint8:
            raise PsalmHipError("image_preprocess: (H,W,3) uint8 image expected")
        H, W = int(img.shape[0]), int(img.shape[1])
        key = ("pp_tables", H, W, nh, nw)
        tb = self._ws.get(key)
        if tb is None:
            if len([k for k in self._ws if isinstance(k, tuple) and k[0] == "pp_tables"]) > 64:
                for k in [k for k in self._ws if isinstance(k, tuple) and k[0] == "pp_tables"]:
                    del self._ws[k]
            bh, kh, ksh = pil_bilinear_tables(W, nw) if nw != W else (None, None, 0)
            bv, kv, ksv = pil_bilinear_tables(H, nh) if nh != H else (None, None, 0)
            dev = lambda a: torch.from_numpy(a).to(self.device) if a is not None else None
            tb = self._ws[key] = (dev(bh), dev(kh), ksh, dev(bv), dev(kv), ksv)
        bh, kh, ksh, bv, kv, ksv = tb
        out = self.empty(3, S, S, dtype=torch.float32)
        pm = self.empty(S, S, dtype=torch.uint8)
        tmp = self.empty(H * nw * 3, dtype=torch.uint8) if nw != W else None
        m = (c_float * 3)(*[float(v) for v in mean.reshape(-1)])
        d = (c_float * 3)(*[float(v) for v in std.reshape(-1)])
        rc = self.lib.psalm_image_preprocess(self._p(img), H, W, self._p(out), self._p(pm), S, nh, nw, self._p(bh), self._p(kh), ksh,
                                             self._p(bv), self._p(kv), ksv, self._p(tmp), m, d, self._stream())
        self._check(rc, "psalm_image_preprocess")
        return out, pm.view(torch.bool)

    def im2col_nhwc(self, x, B, H, W, k, stride, pad):
        C = x.shape[-1]
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        out = self.empty(B * Ho * Wo, k * k * C, dtype=x.dtype)
        rc = self.lib.psalm_im2col_nhwc(self._p(x), self._p(out), _dt(x), B, H, W, C, k, stride, pad, self._stream())
        self._check(rc, "psalm_im2col_nhwc")
        return out

    def im2col_split(self, x, B, H, W, k, stride, pad):
        """im2col_nhwc + split_f16 in one pass: x (B*H*W, C) float32 NHWC tokens -> SplitF16 of the (B*Ho*Wo, k*k*C) patch matrix."""
        C = x.shape[-1]
        if x.dtype != torch.float32 or C % 8:
            raise PsalmHipError("im2col_split: float32 input, C % 8 == 0")
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        K = k * k * C
        Kp = (K + 63) // 64 * 64
        t = self.empty(B * Ho * Wo, 2 * Kp, dtype=torch.float16)
        inv = self.empty(B * Ho * Wo, dtype=torch.float32)
        rc = self.lib.psalm_im2col_split_f16(self._p(x), self._p(t), self._p(inv), B, H, W, C, k, stride, pad, self._stream())
        self._check(rc, "psalm_im2col_split_f16")
        return SplitF16(t, inv, K)

    def resize_planes(self, x, H, W, crop=None, out_dtype=None, out=None):
        """x (N,h,w) -> bilinear (align_corners=False) -> (N,H,W); optional crop (hc,wc) of the input first.  `out`: a contiguous (N,H,W)
        buffer to write into (tests use a view that is not 16-byte aligned to reach the one-pixel-per-thread kernel)."""
        N, h, w = x.shape
        hc, wc = crop if crop is not None else (h, w)
        if out is None:
            out = self.empty(N, H, W, dtype=out_dtype or x.dtype)
        elif tuple(out.shape) != (N, H, W) or not out.is_contiguous():
            raise PsalmHipError("resize_planes: out must be a contiguous (N, H, W) tensor")
        rc = self.lib.psalm_resize_planes(self._p(x), _dt(x), self._p(out), _dt(out), c_long(N), h, w, hc, wc, H, W, self._stream())
        self._check(rc, "psalm_resize_planes")
        return out

    def upsample_add_nhwc(self, lateral, small, B, h, w, H, W, out_dtype=None):
        C = lateral.shape[-1]
        out = self.empty(B * H * W, C, dtype=out_dtype or lateral.dtype)
        rc = self.lib.psalm_upsample_add_nhwc(self._p(lateral), _dt(lateral), self._p(small), _dt(small), self._p(out), _dt(out), B, h,
                                              w, H, W, C, self._stream())
        self._check(rc, "psalm_upsample_add_nhwc")
        return out

    def region_pool(self, tokens, img_of_region, pts, h, w, n_img):
        """tokens (B*h*w, C) f32; img_of_region (R,) i32; pts (R,n,2) f32 (y,x) -> (R, C) f32."""
        R, n = pts.shape[0], pts.shape[1]
        C = tokens.shape[-1]
        out = self.empty(R, C, dtype=torch.float32)
        if tokens.dtype != torch.float32:
            raise PsalmHipError("region_pool expects fp32 image tokens")
        rc = self.lib.psalm_region_pool(self._p(tokens), self._p(img_of_region), self._p(pts), self._p(out), R, h, w, C, n,
                                        self._stream())
        self._check(rc, "psalm_region_pool")
        return out

    def permute_layout(self, x, B, C, HW, to_nhwc, out_dtype=None):
        out = self.empty((B * HW, C) if to_nhwc else (B, C, HW), dtype=out_dtype or x.dtype)
        rc = self.lib.psalm_permute_layout(self._p(x), _dt(x), self._p(out), _dt(out), B, C, c_long(HW), int(to_nhwc), self._stream())
        self._check(rc, "psalm_permute_layout")
        return out

    # ------------------------------------------------------------------ post-processing
    def class_softmax(self, cls, Kpad, probsT_dtype=torch.float32):
        """cls (Q,C1) f32 -> probs (Q,C1), probsT (C1-1,Kpad) zero padded, score (Q), label (Q) i32."""
        Q, C1 = cls.shape
        probs = self.empty(Q, C1)
        probsT = self.zeros(C1 - 1, Kpad, dtype=probsT_dtype)
        score = self.empty(Q)
        label = self.empty(Q, dtype=torch.int32)
        rc = self.lib.psalm_class_softmax(self._p(cls), self._p(probs), self._p(probsT), _dt(probsT), self._p(score), self._p(label),
                                          Q, C1, Kpad, self._stream())
        self._check(rc, "psalm_class_softmax")
        return probs, probsT, score, label

    def sigmoid_transpose(self, mask, Kpad, out_dtype):
        """mask (Q,HW) f32 -> (HW,Kpad) sigmoid, zero padded."""
        Q, HW = mask.shape
        out = self.empty(HW, Kpad, dtype=out_dtype)
        rc = self.lib.psalm_sigmoid_transpose(self._p(mask), self._p(out), _dt(out), Q, c_long(HW), Kpad, self._stream())
        self._check(rc, "psalm_sigmoid_transpose")
        return out

    def semantic_from_masks(self, mask, probsT, want_mask_score=False):
        """mask (Q,HW) f32 logits, probsT (C,128) bf16 -> (C,HW) f32 = probsT @ sigmoid(mask), one pass over the logits; with
        want_mask_score also returns mask_scores(mask) accumulated from the same read."""
        Q, HW = mask.shape
        C, Kpad = probsT.shape
        if probsT.dtype not in (torch.bfloat16, torch.float32) or mask.dtype != torch.float32:
            raise PsalmHipError("semantic_from_masks: f32 logits, bf16 probsT (bf16 MFMA) or f32 probsT (split-f16, fp32-class)")
        out = self.empty(C, HW, dtype=torch.float32)
        ms = self.empty(Q) if want_mask_score else None
        ws = self.empty(Q * 512 * 2) if want_mask_score else None
        fn = self.lib.psalm_semantic_from_masks_x3 if probsT.dtype == torch.float32 else self.lib.psalm_semantic_from_masks
        rc = fn(self._p(mask), self._p(probsT), self._p(out), self._p(ms) if want_mask_score else None,
                self._p(ws) if want_mask_score else None, Q, C, c_long(HW), Kpad, self._stream())
        self._check(rc, "psalm_semantic_from_masks")
        return (out, ms) if want_mask_score else out

    def mask_scores(self, mask):
        """mask (Q,HW) f32 -> (Q) f32: sum(sigmoid*[m>0]) / (sum([m>0]) + 1e-6)."""
        Q, HW = mask.shape
        score = self.empty(Q)
        ws = self.empty(Q * 64 * 2)
        rc = self.lib.psalm_mask_scores(self._p(mask), self._p(score), self._p(ws), Q, c_long(HW), self._stream())
        self._check(rc, "psalm_mask_scores")
        return score

    def topk_select(self, vals, C, k, is_thing=None, mask_score=None, apply_sigmoid=False, count_out=None):
        """vals (Q,stride) f32, first C columns are candidates.  Returns (score (k), class (k) i32, query (k) i32, count (1) i32),
        entries [0,count) valid, in descending candidate order.  count_out: a zeroed (1,) int32 view to write the count into."""
        Q, stride = vals.shape
        sc = self.zeros(k)
        cl = self.zeros(k, dtype=torch.int32)
        qq = self.zeros(k, dtype=torch.int32)
        cnt = count_out if count_out is not None else self.zeros(1, dtype=torch.int32)
        rc = self.lib.psalm_topk_select(self._p(vals), Q, C, stride, k, self._p(is_thing), self._p(mask_score), self._p(sc), self._p(cl),
                                        self._p(qq), self._p(cnt), int(apply_sigmoid), self._stream())
        self._check(rc, "psalm_topk_select")
        return sc, cl, qq, cnt

    def binarize_gather(self, mask, n, query=None, count=None):
        """out[i] = (mask[query[i]] > 0).float() for i < count (rows >= count untouched); mask (Q,H,W)."""
        Q, Hh, Ww = mask.shape
        out = self.empty(n, Hh, Ww)
        rc = self.lib.psalm_binarize_gather(self._p(mask), self._p(query), self._p(count), self._p(out), n, c_long(Hh * Ww), self._stream())
        self._check(rc, "psalm_binarize_gather")
        return out

    def panoptic(self, mask, score, label, is_thing, num_classes, obj_thr, overlap_thr, info_out=None, ninfo_out=None):
        """mask (Q,H,W) f32 logits; returns (pan (H,W) i32, info (Q,3) i32, ninfo (1) i32).  info_out / ninfo_out: zeroed int32 views
        ((Q,3) / (1,)) to write into -- the caller fetches its data-dependent counts in ONE device-to-host copy."""
        Q, Hh, Ww = mask.shape
        HW = Hh * Ww
        argq = self.empty(HW, dtype=torch.int32)
        counts = self.empty(Q * 3, dtype=torch.int32)
        final_id = self.empty(Q + num_classes + 1, dtype=torch.int32)
        pan = self.empty(Hh, Ww, dtype=torch.int32)
        info = info_out if info_out is not None else self.zeros(Q, 3, dtype=torch.int32)
        ninfo = ninfo_out if ninfo_out is not None else self.zeros(1, dtype=torch.int32)
        rc = self.lib.psalm_panoptic(self._p(mask), self._p(score), self._p(label), self._p(is_thing), self._p(argq), self._p(counts),
                                     self._p(final_id), self._p(pan), self._p(info), self._p(ninfo), Q, c_long(HW), num_classes,
                                     c_float(obj_thr), c_float(overlap_thr), self._stream())
        self._check(rc, "psalm_panoptic")
        return pan, info, ninfo

    def region_scores(self, logits, mask_score):
        """logits (K,Q) f32 -> (Q,K) sigmoid(logits).T * mask_score[:,None]."""
        K, Q = logits.shape
        out = self.empty(Q, K)
        rc = self.lib.psalm_region_scores(self._p(logits), self._p(mask_score), self._p(out), K, Q, self._stream())
        self._check(rc, "psalm_region_scores")
        return out

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

    def msda_forward_dev(self, value, spatial_shapes, level_start, loc, attw, out_dtype=None):
        """msda_forward with `spatial_shapes` (L,2) / `level_start` (L) as int64 tensors ON THE DEVICE (the reference op's signature):
        nothing is copied to the host, the call is asynchronous."""
        B, S, M, D = value.shape
        _, Lq, _, L, P, _ = loc.shape
        for t in (spatial_shapes, level_start):
            if t.dtype != torch.int64 or t.device != value.device:
                raise PsalmHipError("msda_forward_dev: int64 level tensors on the value's device")
        if tuple(spatial_shapes.shape) != (L, 2) or level_start.numel() != L:
            raise PsalmHipError(f"msda_forward_dev: level table shapes {tuple(spatial_shapes.shape)} / {tuple(level_start.shape)} for L = {L}")
        # each DISTINCT level table is checked against S once (one small D2H copy the first time; keyed by storage + version counter, so
        # the steady state stays asynchronous); not during stream capture, where the kernel's own bounds guard is the safety net
        key = (spatial_shapes.data_ptr(), spatial_shapes._version, level_start.data_ptr(), level_start._version, S, L)
        if key not in self._msda_tables and not (value.is_cuda and torch.cuda.is_current_stream_capturing()):
            hw, st = spatial_shapes.cpu().tolist(), level_start.reshape(-1).cpu().tolist()
            if any(h <= 0 or w_ <= 0 or s0 < 0 or s0 + h * w_ > S for (h, w_), s0 in zip(hw, st)) or sum(h * w_ for h, w_ in hw) != S:
                raise PsalmHipError(f"msda_forward_dev: level table {hw} / starts {st} does not describe the {S} rows of value")
            if len(self._msda_tables) > 64:
                self._msda_tables.clear()
            self._msda_tables.add(key)
        out = self.empty(B, Lq, M * D, dtype=out_dtype or value.dtype)
        rc = self.lib.psalm_msda_forward_dev(self._p(value), _dt(value), self._p(spatial_shapes.contiguous()), self._p(level_start.contiguous()),
                                             self._p(loc), self._p(attw), self._p(out), _dt(out), B, S, M, D, L, Lq, P, self._stream())
        self._check(rc, "psalm_msda_forward_dev")
        return out

    def msda_policy(self, v: int):
        """psalm_msda_set_policy: 1 (default) / 0 = quad-shared bilinear taps on / off (A/B runs, tests)"""
        self._check(self._cdll_raw.psalm_msda_set_policy(int(v)), "psalm_msda_set_policy")

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
