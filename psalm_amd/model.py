"""PSALM segmentation inference (`eval_seg`) on hand-written gfx950 kernels.

Host-side mirror of the reference's model API for this path:
    psalm/model/language_model/llava_phi.py:146-1472  (class PSALM: eval_seg and everything it calls)
Same call signature and result dicts as `PSALM.eval_seg` (LP:1317-1336 -> LP:1424-1466), same checkpoint layout
(`state_dict` names of the reference, see psalm_amd/synthetic.py), but every tensor operation is a kernel from
libpsalm_hip.so called through psalm_amd.hip_ops (ctypes; torch = device memory + stream only).  There is no
torch / CPU fallback: without the HIP library and a GPU, construction raises.

Precision modes
    "bf16": weights bf16, GEMMs on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; GEMM-feeding activations bf16;
            residual streams, LayerNorm/GroupNorm/softmax statistics, sampling offsets, mask and class logits fp32.
    "fp32": weights and activations fp32, GEMMs on the exact fp32 MFMA -- structural parity mode against the fp32
            CPU reference (differences are summation-order round-off only).
    "f16x3": the qualifying fast mode.  "fp32" with every GEMM in split-f16 arithmetic (psalm_split_f16 / psalm_gemm_x3): each fp32
            operand is carried as two f16 values (22 mantissa bits, per-row power-of-two scale) and the product is one f16 MFMA GEMM
            over the 3x longer panel hi.hi + lo.hi + hi.lo with fp32 accumulation.  bf16 operands do NOT meet the north star's parity
            bar on this network (the mask decoder's thresholded attention-mask feedback amplifies operand rounding into label flips;
            measured threshold between 15 and 17 operand bits, tools/exp_bits.py); this mode does, at 3 f16 MFMA passes instead of
            the fp32 MFMA's 16x lower rate.  Activations, norms, softmax and the attention kernels are those of "fp32".
            (r03 carried an opt-in form of the Phi GEMMs with e4m3 cross terms -- BASELINE.json configs[4]'s "fp8 MFMA LLM path"; it moved
            ~5 % of the inputs by 1e-3 .. 6e-2 of the logit range and was removed in r04: DESIGN.md section 0, tools/exp_x8_cpu.py.)

Layout: activations are token-major (rows = pixels/tokens, cols = channels; NHWC for feature maps), so 1x1
convolutions are GEMMs, 3x3 / strided convolutions are im2col + GEMM, and LayerNorm/softmax rows are contiguous.

Deliberate, value-preserving departures from the reference's control flow (SURVEY.md §7 "quirks"):
  * the Swin tower runs once per image, not twice (LP:787 and LP:1369 feed it the same pixels);
  * every image of the batch is post-processed (LP:1472 returns inside the loop after image 0);
  * the prediction heads' class/SEG/region logits are evaluated only after the last decoder layer (the nine
    earlier evaluations are auxiliary training outputs, TD:672-690); the mask logits are needed every layer
    because they define the next layer's attention mask;
  * BatchNorm (eval) is folded into the projector convolutions; q/k/v/fc1 and dense/fc2 of each Phi layer are fused
    into two GEMMs (the parallel-residual layer makes attention and MLP share their input and their sum).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import hip_ops as H
from .config import (CLS_TOKEN_INDEX, IMAGE_TOKEN_INDEX, REFER_TOKEN_INDEX, REGION_TOKEN_INDEX, SEG_TOKEN_INDEX,
                     PsalmConfig)


class Instances:
    """Result container with the attribute surface the reference's evaluators read from detectron2 `Instances`
    (LP:317-323, LP:435-447): image_size, pred_masks, scores, pred_classes, pred_boxes."""

    def __init__(self, image_size, **fields):
        self.image_size = tuple(image_size)
        self._fields = dict(fields)
        for k, v in fields.items():
            setattr(self, k, v)

    def has(self, name):
        return name in self._fields

    def get_fields(self):
        return self._fields

    def to(self, device):
        return Instances(self.image_size, **{k: (v.to(device) if torch.is_tensor(v) else v) for k, v in self._fields.items()})

    def __len__(self):
        return int(self.pred_masks.shape[0])


def _nonzero_2d(m: torch.Tensor) -> torch.Tensor:
    """`m.nonzero()` of a 2-D CPU mask -- (k, 2) int64 (y, x) in row-major order -- without torch's two full passes over the bytes: a
    1024 x 1024 region mask costs torch ~2 ms per region on the host (r04: 7 of the 37.5 ms of a region batch); here the mask is scanned
    as 64-bit words and only the non-zero words are looked at byte by byte."""
    if m.dim() != 2 or m.dtype != torch.bool or not m.is_contiguous() or m.numel() % 8 or m.device.type != "cpu":
        return m.nonzero()
    a = m.numpy().view(np.uint8).reshape(-1)
    words = np.flatnonzero(a.view(np.uint64))
    if words.size == 0:
        return torch.zeros(0, 2, dtype=torch.int64)
    if words.size * 128 > a.size:                     # not sparse (> 1/16 of the words occupied): torch's dense scan is the faster one
        return m.nonzero()
    cand = (words[:, None] * 8 + np.arange(8, dtype=np.int64)[None]).reshape(-1)
    flat = cand[a[cand] != 0]
    W = m.shape[1]
    return torch.from_numpy(np.stack((flat // W, flat % W), 1))


def default_region_point_sampler(nonzero: torch.Tensor, n: int) -> torch.Tensor:
    """Row indices into `nonzero` (context_cluster.py:31-40 rand_sample_repeat; global torch CPU RNG, same call order)."""
    m = nonzero.shape[0]
    if m < n:
        return torch.cat((torch.arange(m), torch.randint(0, m, (n - m,))))
    if m == n:
        return torch.arange(m)
    return torch.randperm(m)[:n]


class _VisionTower:
    """What `model.get_vision_tower()` hands the reference's builder (psalm/model/builder.py:57-65): `.image_processor` -- the dict
    of the three dataset mappers (llava_phi.py:66-69) -- and a `.to(...)` that the builder calls to move the tower."""
    is_loaded = True

    def __init__(self, image_processor):
        self.image_processor = image_processor

    def to(self, *args, **kwargs):
        return self


class PSALM:
    # The default is the mode that meets the reference's fp32 results to the north star's tolerance; "bf16" is the faster, looser mode.
    DEFAULT_PRECISION = "f16x3"

    def __init__(self, cfg: PsalmConfig, state_dict: Dict[str, torch.Tensor], ops: Optional[H.Ops] = None,
                 precision: Optional[str] = None, use_graphs: bool = False, paired_split_stores: Optional[bool] = None,
                 llm_products: int = 3):
        precision = precision or self.DEFAULT_PRECISION
        if precision not in ("bf16", "fp32", "f16x3"):
            raise ValueError("precision must be 'f16x3', 'fp32' or 'bf16'")
        if llm_products not in (1, 3) or (llm_products == 1 and precision != "f16x3"):
            raise ValueError("llm_products: 3, or 1 with precision='f16x3'")
        # BASELINE.json configs[4]'s reduced-precision LLM path, as an opt-in SIDE MODE with its own (looser) stated tolerance: llm_products = 1
        # runs the two fused GEMMs of every Phi layer on plain f16 operands (the hi halves of the split-f16 operands, 11-bit mantissas under the
        # same per-row scales: ONE f16 MFMA product instead of three), everything else -- Swin, attention, pixel decoder, mask decoder -- as in
        # "f16x3".  It does NOT meet the north star's fp32 parity bar (the mask decoder's thresholded feedback needs 15-17 operand bits,
        # tools/exp_bits.py); tests/test_9_e2e_gpu.py::test_config5_region_1024_batch2_reduced_precision_llm states what it does meet.
        self.llm_products = llm_products
        self.cfg = cfg
        self.ops = ops if ops is not None else H.get_ops()        # raises without GPU + libpsalm_hip.so
        self.precision = precision
        self.x3 = precision == "f16x3"                # fp32 activations, GEMM operands in split-f16 form (hip_ops.SplitF16)
        self.wdt = torch.float32 if precision in ("fp32", "f16x3") else torch.bfloat16   # weight / GEMM-operand dtype
        self.adt = self.wdt                                                     # GEMM-feeding activation dtype
        self.device = self.ops.device
        self.seg_task = cfg.seg_task
        self.is_thing_list = None
        self._cache: Dict = {}
        self._graphs: Dict = {}
        self._plan_cache: Dict = {}
        self._prep_cache: Dict = {}                   # _prepare results by prompt-tensor identity (see _prepare)
        self.max_graphs = 8                           # captured input signatures kept alive (oldest dropped first)
        # Input-geometry independence of the captured launch sequence (r05).  The reference's evaluation loops feed images whose un-padded
        # box and original size differ image by image (ResizeShortestEdge + FixedSizeCrop, coco_panoptic_mapper.py:81-89; crop / resize at
        # LP:1418-1429) and referring prompts whose length differs sentence by sentence (train_datasets.py:644-695).  So that such a stream
        # replays ONE graph: (a) the sequence length is padded to a multiple of `len_bucket` with masked, zero rows -- exactly how the
        # shorter prompts of a ragged batch are already carried (LP:939-946) --, the variable-length row sets of the blob to multiples of
        # `len_bucket` entries; (b) the graph ends where the per-image geometry starts: it holds everything up to the mask logits at the
        # padded image size and the class softmax, and the crop / resize / inference tail of LP:1418-1466 (a dozen launches whose sizes are
        # the image's own) is issued after the replay, while the GPU is still inside the graph.  graph_tail = True puts the tail back into
        # the graph (and the geometry back into its key): the r01-r04 behaviour, kept for A/B measurements.
        self.len_bucket = 32
        # Stage-level native calls (csrc/stages.hip, SURVEY section 8(b) "C-ABI groups behind B2"): the Phi decoder's ~100 launches are issued by ONE
        # call into the library instead of one ctypes call each -- same launches, same order, same bits (tests/test_6_model_emu.py); off while
        # bench.py's per-launch events are being recorded (they attribute time per op-level call).
        self.c_stages = True
        self.kv_side = True        # r06: the predictor's K / V front on the side stream beside the LLM (False: at the start of the predictor call, as r05; A/B switch)
        self.graph_tail = False
        self.graph_stats = {"calls": 0, "replays": 0, "eager": 0, "captures": 0}
        self.use_graphs = use_graphs                  # capture each input signature's launch sequence into a hipGraph
        # Graph replay writes its results into buffers owned by the captured graph; "copy" (default) hands the caller private copies,
        # "alias" returns the graph's own buffers, which the NEXT call with the same input signature overwrites -- only for callers that
        # consume a result before the next call (bench.py, the reference's evaluators do; an evaluator that keeps tensors across iterations
        # would silently read the next image's masks).  Since r05 the graph ends before the per-image crop / resize / inference tail
        # (`graph_tail`), whose results are fresh buffers anyway: the one tensor that can still alias the graph is `mask_pred` of an image
        # that needs no crop / resize (0.4 GB at 1024^2, ~0.1 ms to copy).
        self.graph_outputs = "copy"
        # pixel decoder on a second HIP stream, concurrent with the LLM (fork / join, captured into the hipGraph).  bf16 mode, r1k: no
        # gain (the LLM GEMMs fill the chip).  f16x3, r02: the 3x longer LLM GEMMs leave room (224 tiles on 256 CUs) for the decoder's ~150
        # small kernels: 28.96 -> 28.04 ms per image -> on by default in that mode.
        self.overlap_streams = precision == "f16x3"
        # f16x3: GEMM / attention outputs that feed another GEMM leave their kernel already in split-f16 operand form (psalm_gemm_x3_split,
        # psalm_*_attention*_split, psalm_gemm_x3_ln_split) instead of fp32 + a psalm_split_f16 pass.  False: the r02k data flow (tools/exp_modes.py A/B)
        self.fuse_split = precision == "f16x3"
        # ... and where the emitting kernel is a GEMM (Swin fc1, encoder linear1, Phi fc1) the weight rows are stored PERMUTED inside groups
        # of 64 (H.Ops.so_pair_perm) so that the 2-byte operand leaves in 4-byte stores of whole 128-byte row segments straight from the
        # accumulators (psalm_gemm_x3_split, `paired`) -- results bit for bit those of the un-permuted layout.  A construction-time
        # choice (the weights are laid out for it): such a model cannot be switched to fuse_split = False afterwards.
        self.so_paired = self.fuse_split if paired_split_stores is None else (bool(paired_split_stores) and self.fuse_split)
        self._side = None
        self.w: Dict[str, torch.Tensor] = {}
        self.paired: Dict[str, bool] = {}            # linear name -> its weight rows are permuted for paired split-f16 stores
        self.config = None                            # LlavaConfig when built by from_pretrained (llava_phi.py:34)
        self.image_processor = None                   # dict of pre-processors, set by from_pretrained / load_pretrained_model
        self.training = False
        self._prepare_weights(state_dict)

    def replica(self) -> "PSALM":
        """A second instance of this model for ANOTHER HIP stream / host thread (two images in flight on one GPU: the hardware interleaves the
        launches of independent images -- profiles/r03a_inflight2.json, +11 % images/s): shares the weights, owns everything a call
        writes -- its binding's workspaces, constant caches, splice plans, captured graphs and side stream.  `eval_seg` itself stays the
        reference's synchronous call; the caller drives each replica from its own thread under `torch.cuda.stream(...)`."""
        import copy
        r = copy.copy(self)
        r.ops = H.Ops(self.ops.lib_path)
        r.ops.x3, r.ops.debug_bounds = self.ops.x3, self.ops.debug_bounds
        r._cache, r._graphs, r._plan_cache, r._prep_cache = {}, {}, {}, {}
        r._side = None
        r.graph_stats = dict.fromkeys(self.graph_stats, 0)
        return r

    # ======================================================================================= nn.Module / HF surface the eval scripts touch
    def to(self, *args, **kwargs):
        """`model.to(dtype=torch.float32, device=device)` (psalm/eval/panoptic_segmentation.py:127 and siblings).  The weights already
        live on this binding's device in the layout `precision` chose, so: the device must be that device (there is no CPU path), a
        floating dtype is accepted and has no effect -- the arithmetic is selected by `precision`, whose default meets fp32 parity."""
        dev = kwargs.get("device")
        for a in args:
            if isinstance(a, (str, torch.device)):
                dev = a
        if dev is not None:
            dev = torch.device(dev)
            if dev.type != self.device.type or (dev.index is not None and self.device.index is not None and dev.index != self.device.index):
                raise H.PsalmHipError(f"PSALM.to({dev}): this model's weights live on {self.device} (hand-written gfx950 kernels, no CPU "
                                      "fallback, no migration between GPUs: build one model per device)")
        dt = kwargs.get("dtype", next((a for a in args if isinstance(a, torch.dtype)), None))
        if dt is not None and not dt.is_floating_point:
            raise TypeError(f"PSALM.to(dtype={dt}): floating dtype expected")
        return self

    def eval(self):
        return self

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("psalm_amd.PSALM is the inference path (eval_seg / eval_video); training is out of scope")
        return self

    def float(self):
        return self

    def half(self):
        return self

    def bfloat16(self):
        return self

    def cuda(self, device=None):
        return self.to("cuda")

    def requires_grad_(self, flag: bool = False):
        return self

    def get_model(self):                              # llava_phi.py:484
        return self

    def get_vision_tower(self):                       # llava_phi.py:72; consumed at psalm/model/builder.py:57-65
        return _VisionTower(self.image_processor)

    @classmethod
    def from_pretrained(cls, model_path, mask_decoder_cfg=None, *, precision: Optional[str] = None, use_graphs: bool = True, ops=None,
                        seg_task: Optional[str] = None, **hf_kwargs):
        """`PSALM.from_pretrained(model_path, mask_decoder_cfg=mask_cfg, **kwargs)` (psalm/model/builder.py:54): Hugging Face checkpoint
        directory -> model.  `mask_decoder_cfg`: the (attribute-style) mask YAML the reference passes, or None for the released
        defaults.  The Hugging Face loader keywords the reference passes (torch_dtype, device_map, low_cpu_mem_usage ...) are accepted
        and ignored; bitsandbytes quantisation is rejected.  `precision`, `use_graphs`, `ops`, `seg_task` are extensions."""
        from . import builder as B
        from .config import load_mask_config
        if hf_kwargs.get("load_in_8bit") or hf_kwargs.get("load_in_4bit") or hf_kwargs.get("quantization_config") is not None:
            raise NotImplementedError("bitsandbytes 8/4-bit loading is a CUDA feature of the reference loader; not available on this path")
        mask_cfg = mask_decoder_cfg if mask_decoder_cfg is not None else load_mask_config(None)
        task = seg_task or mask_cfg.MODEL.MASK_FORMER.SEG_TASK
        cfg = B.config_from_hf(model_path, mask_cfg, task)
        model = cls(cfg, B.read_checkpoint(model_path), ops=ops, precision=precision or cls.DEFAULT_PRECISION, use_graphs=use_graphs)
        model.config = B.hf_config(model_path)
        m = mask_cfg.MODEL
        proc = B.ImagePreprocessor(mask_cfg.INPUT.IMAGE_SIZE, m.PIXEL_MEAN, m.PIXEL_STD, device=None)
        model.image_processor = {"panoptic": proc, "instance": proc, "semantic": proc}             # llava_phi.py:66-69
        return model

    # ======================================================================================= weights
    @staticmethod
    def _aligned(t):                      # kernels take 16-byte vector accesses; memory-mapped checkpoint views may not be aligned
        return t if t.data_ptr() % 16 == 0 else t.clone()

    def _W(self, t):                      # GEMM weight
        t = self._aligned(t.detach().to(torch.float32).contiguous().to(self.wdt).to(self.device))
        return self.ops.split_f16(t) if self.x3 else t      # f16x3: split once at load time ([hi | lo] f16 + per-row scales)

    def _Ws(self, t):                     # weight of a GEMM whose M is the ~100 decoder queries / a handful of prompt rows: in the f16x3 mode
        #                                   these stay fp32 and run on the exact-fp32 skinny kernel (no split launch, latency-bound anyway)
        return self._aligned(t.detach().to(torch.float32).contiguous().to(self.wdt).to(self.device))

    def _wop(self, t):                    # an ACTIVATION used as the W operand of a GEMM (mask features, class embeddings, ...)
        return self.ops.split_f16(t) if self.x3 and t is not None and t.shape[0] > 4096 else t

    def _pair_rows(self, wt, bias, start=0):
        """(weight, bias, paired?) with the rows >= start permuted for paired split-f16 stores, when the shape allows it"""
        n = wt.shape[0] - start
        if not (self.so_paired and self.x3) or n <= 0 or n % 64 or start % 256:
            return wt, bias, False
        idx = torch.cat([torch.arange(start), start + H.Ops.so_pair_perm(n)]).to(wt.device)
        return wt.detach()[idx], (bias.detach()[idx.to(bias.device)] if bias is not None else None), True

    def _gemm_act_split(self, a, name, act):
        """f16x3: act(a . W^T + b) of linear `name` straight into the split-f16 operand form of the GEMM that consumes it (no fp32 round
        trip, no psalm_split_f16 pass); None when the shape does not allow it (caller falls back to gemm + implicit split)."""
        o, w = self.ops, self.w
        wt = w[name + ".w"]
        bnd = w.get(name + ".bnd")
        N = wt.shape[0]
        paired = self.paired.get(name, False)
        if not self.fuse_split or bnd is None or N % 8 != 0 or not isinstance(wt, H.SplitF16):
            if paired:
                raise H.PsalmHipError(f"{name}: weight rows are laid out for paired split-f16 stores (PSALM(paired_split_stores=True)); "
                                      "build the model with paired_split_stores=False to run it with fuse_split off")
            return None
        rows = a.shape[0]
        Kp = (N + 63) // 64 * 64                                      # the consumer's K padding columns must read as zeros
        so = (o.empty if Kp == N else o.zeros)(rows, 2 * Kp, dtype=torch.float16)
        inv = o.empty(rows, dtype=torch.float32)
        o.gemm_x3_split(a, wt, w.get(name + ".b"), act, so, inv, bnd, paired=paired)
        return H.SplitF16(so, inv, N)

    def _F(self, t):                      # fp32 parameter (bias, norm scale, tables)
        return self._aligned(t.detach().to(torch.float32).contiguous().to(self.device))

    def _prepare_weights(self, sd):
        cfg, w = self.cfg, self.w
        W, Fp = self._W, self._F
        # (the stage descriptors and captured graphs hold raw device pointers into `w`: a second preparation replaces those tensors, so whatever was
        #  derived from the old ones goes -- ADVICE r05; broadcast_weights writes IN PLACE and keeps them valid)
        for k in [k for k in getattr(self, "_cache", {}) if isinstance(k, tuple) and k and isinstance(k[0], str) and k[0].endswith("_desc")]:
            del self._cache[k]
        if getattr(self, "_graphs", None):
            self._graphs.clear()

        def lin(dst, src, bias=True, small=False, pair=False):
            wt, bs = sd[src + ".weight"], (sd[src + ".bias"] if bias and (src + ".bias") in sd else None)
            if pair:                                      # this linear's GEMM emits split-f16 output: rows permuted for paired stores
                wt, bs, self.paired[dst] = self._pair_rows(wt, bs)
            w[dst + ".w"] = (self._Ws if small else W)(wt)
            if bs is not None:
                w[dst + ".b"] = Fp(bs)

        def norm(dst, src):
            w[dst + ".g"] = Fp(sd[src + ".weight"])
            w[dst + ".b"] = Fp(sd[src + ".bias"])

        def bound(dst, wt, bias, wt_g=None, bias_g=None):
            """f16x3: the 4 magnitude-bound parameters of psalm_gemm_x3_split for a GEMM with weight rows `wt` / `bias` whose activated
            output leaves in split-f16 form: {2^14 max_n sum_k |w_nk|, max |bias|, g1, g0}; (wt_g, bias_g): the rows whose outputs reach
            the same operand rows through ANOTHER kernel (Phi: v_proj, through the attention) -> the row-independent term g1, g0."""
            if not self.x3:
                return

            def l1(t):
                return t.detach().to(self.device, torch.float32).abs().sum(1).max() * (1.0 + 1e-5) * 2.0 ** 14

            def amax(t):
                return t.detach().to(self.device, torch.float32).abs().max() if t is not None else torch.zeros((), device=self.device)
            zero = torch.zeros((), device=self.device)
            w[dst] = torch.stack([l1(wt), amax(bias), l1(wt_g) if wt_g is not None else zero, amax(bias_g)]).to(torch.float32).contiguous()

        # ---- Phi decoder.  Fused GEMM 1 rows: [k | v | q | fc1]  (attention output later overwrites the q columns, so
        # [attn | gelu(fc1)] is one contiguous K panel for fused GEMM 2 = [dense | fc2] with the two biases summed).
        w["embed"] = Fp(sd["model.embed_tokens.weight"]) if self.wdt == torch.float32 else \
            sd["model.embed_tokens.weight"].detach().to(torch.bfloat16).to(self.device)
        w["seg_query"] = Fp(sd["seg_query"])
        for i in range(cfg.num_layers):
            p = f"model.layers.{i}."
            a = p + "self_attn."
            w1 = torch.cat([sd[a + "k_proj.weight"], sd[a + "v_proj.weight"], sd[a + "q_proj.weight"], sd[p + "mlp.fc1.weight"]], 0)
            w2 = torch.cat([sd[a + "dense.weight"], sd[p + "mlp.fc2.weight"]], 1)
            b1 = torch.cat([sd[a + "k_proj.bias"], sd[a + "v_proj.bias"], sd[a + "q_proj.bias"], sd[p + "mlp.fc1.bias"]], 0)
            w1, b1, self.paired[f"llm{i}"] = self._pair_rows(w1, b1, 3 * cfg.hidden_size)      # the fc1 rows: gelu(fc1) leaves as fc2's operand
            w[f"llm{i}.w1"], w[f"llm{i}.w2"] = W(w1), W(w2)
            w[f"llm{i}.b1"] = Fp(b1)
            w[f"llm{i}.b2"] = Fp(sd[a + "dense.bias"].float() + sd[p + "mlp.fc2.bias"].float())
            norm(f"llm{i}.ln", p + "input_layernorm")
            bound(f"llm{i}.bnd", sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"], sd[a + "v_proj.weight"], sd[a + "v_proj.bias"])
        norm("llm.final", "model.final_layernorm")

        # ---- Swin
        vt = "model.vision_tower."
        E, ps = cfg.swin_embed_dim, cfg.swin_patch
        pe = sd[vt + "patch_embed.proj.weight"].reshape(E, -1)                 # (E, 3*ps*ps), K order (c,ky,kx)
        # K of the patch-embed GEMM: padded to 64 in the bf16 modes (direct-to-LDS kernel), to 8 in the exact mode
        self.pe_kpad = (pe.shape[1] + 63) // 64 * 64 if self.wdt == torch.bfloat16 else (pe.shape[1] + 7) // 8 * 8
        w["swin.pe.w"] = W(torch.nn.functional.pad(pe, (0, self.pe_kpad - pe.shape[1])))
        w["swin.pe.b"] = Fp(sd[vt + "patch_embed.proj.bias"])
        norm("swin.pe.ln", vt + "patch_embed.norm")
        for s, depth in enumerate(cfg.swin_depths):
            for b in range(depth):
                p, q = f"{vt}layers.{s}.blocks.{b}.", f"swin{s}.{b}."
                norm(q + "n1", p + "norm1")
                norm(q + "n2", p + "norm2")
                lin(q + "qkv", p + "attn.qkv")
                lin(q + "proj", p + "attn.proj")
                lin(q + "fc1", p + "mlp.fc1", pair=True)
                bound(q + "fc1.bnd", sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
                Cq = sd[p + "attn.qkv.weight"].shape[0] // 3                       # bound of the v rows: the window attention's output scale
                bound(q + "qkv.bnd", sd[p + "attn.qkv.weight"][2 * Cq:], sd[p + "attn.qkv.bias"][2 * Cq:])
                lin(q + "fc2", p + "mlp.fc2")
                w[q + "rpb"] = Fp(sd[p + "attn.relative_position_bias_table"])
            if s < len(cfg.swin_depths) - 1:
                norm(f"swin{s}.ds.ln", f"{vt}layers.{s}.downsample.norm")
                lin(f"swin{s}.ds.red", f"{vt}layers.{s}.downsample.reduction", bias=False)
            norm(f"swin.out{s}", f"{vt}norm{s}")

        # ---- projector (BasicBlock with eval BatchNorm folded; conv weights to (Cout, ky, kx, Cin))
        pj = "model.mm_projector.layer1.0."

        def bn_fold(conv_w, bn):                       # in fp32 whatever the checkpoint dtype (bf16 would lose the 1e-5 eps)
            scale = sd[bn + ".weight"].float() / torch.sqrt(sd[bn + ".running_var"].float() + 1e-5)
            shift = sd[bn + ".bias"].float() - sd[bn + ".running_mean"].float() * scale
            return conv_w.float() * scale.view(-1, 1, 1, 1), shift

        def conv_mat(cw):
            return cw.permute(0, 2, 3, 1).reshape(cw.shape[0], -1)
        c1, b1 = bn_fold(sd[pj + "conv1.weight"], pj + "bn1")
        w["proj.c1.w"], w["proj.c1.b"] = W(conv_mat(c1)), Fp(b1)
        w["proj.c2.w"] = W(conv_mat(sd[pj + "conv2.weight"]))
        c2, b2 = bn_fold(sd[pj + "conv2.weight"], pj + "bn2")
        w["proj.c2f.w"], w["proj.c2f.b"] = W(conv_mat(c2)), Fp(b2)
        cd, bd = bn_fold(sd[pj + "downsample.0.weight"], pj + "downsample.1")
        w["proj.ds.w"], w["proj.ds.b"] = W(conv_mat(cd)), Fp(bd)
        lin("proj.fc", "model.mm_projector.fc")

        # ---- LLM -> decoder projectors
        for n in ("seg_query_projector", "SEG_token_projector", "class_name_projector", "region_projector"):
            lin(n, n, small=True)

        # ---- pixel decoder
        pd = "pixel_decoder."
        for i in range(3):
            w[f"pd.ip{i}.w"] = W(sd[f"{pd}input_proj.{i}.0.weight"].flatten(1))
            w[f"pd.ip{i}.b"] = Fp(sd[f"{pd}input_proj.{i}.0.bias"])
            norm(f"pd.ip{i}.gn", f"{pd}input_proj.{i}.1")
        w["pd.level_embed"] = Fp(sd[pd + "transformer.level_embed"])
        for i in range(cfg.md_enc_layers):
            p, q = f"{pd}transformer.encoder.layers.{i}.", f"pd.enc{i}."
            w[q + "ow.w"] = W(torch.cat([sd[p + "self_attn.sampling_offsets.weight"], sd[p + "self_attn.attention_weights.weight"]], 0))
            w[q + "ow.b"] = Fp(torch.cat([sd[p + "self_attn.sampling_offsets.bias"], sd[p + "self_attn.attention_weights.bias"]], 0))
            lin(q + "value", p + "self_attn.value_proj")
            lin(q + "out", p + "self_attn.output_proj")
            norm(q + "n1", p + "norm1")
            lin(q + "l1", p + "linear1", pair=True)
            bound(q + "l1.bnd", sd[p + "linear1.weight"], sd[p + "linear1.bias"])
            lin(q + "l2", p + "linear2")
            norm(q + "n2", p + "norm2")
        w["pd.adapter.w"] = W(sd[pd + "adapter_1.0.weight"].flatten(1))
        w["pd.adapter.b"] = Fp(sd[pd + "adapter_1.0.bias"])
        norm("pd.adapter.gn", pd + "adapter_1.1")
        w["pd.layer.w"] = W(conv_mat(sd[pd + "layer_1.0.weight"]))
        w["pd.layer.b"] = Fp(sd[pd + "layer_1.0.bias"])
        norm("pd.layer.gn", pd + "layer_1.1")
        w["pd.mf.w"] = W(sd[pd + "mask_features.weight"].flatten(1))
        w["pd.mf.b"] = Fp(sd[pd + "mask_features.bias"])

        # ---- predictor
        pr = "predictor."
        D = cfg.md_hidden
        nl, nlev = cfg.md_dec_layers, cfg.md_levels
        for i in range(nl):
            c = f"{pr}transformer_cross_attention_layers.{i}."
            Wi, bi = sd[c + "multihead_attn.in_proj_weight"], sd[c + "multihead_attn.in_proj_bias"]
            w[f"pr{i}.cq.w"], w[f"pr{i}.cq.b"] = self._Ws(Wi[:D]), Fp(bi[:D])
            lin(f"pr{i}.co", c + "multihead_attn.out_proj", small=True)
            norm(f"pr{i}.cn", c + "norm")
            s_ = f"{pr}transformer_self_attention_layers.{i}."
            Ws, bs = sd[s_ + "self_attn.in_proj_weight"], sd[s_ + "self_attn.in_proj_bias"]
            w[f"pr{i}.sqk.w"], w[f"pr{i}.sqk.b"] = self._Ws(Ws[:2 * D]), Fp(bs[:2 * D])
            w[f"pr{i}.sv.w"], w[f"pr{i}.sv.b"] = self._Ws(Ws[2 * D:]), Fp(bs[2 * D:])
            lin(f"pr{i}.so", s_ + "self_attn.out_proj", small=True)
            norm(f"pr{i}.sn", s_ + "norm")
            f_ = f"{pr}transformer_ffn_layers.{i}."
            lin(f"pr{i}.f1", f_ + "linear1", small=True)
            lin(f"pr{i}.f2", f_ + "linear2", small=True)
            norm(f"pr{i}.fn", f_ + "norm")
        # cross-attention K / V projections of all layers that read level l, stacked: one GEMM per level
        for l in range(nlev):
            layers = [i for i in range(nl) if i % nlev == l]
            ks, vs, kb, vb = [], [], [], []
            for i in layers:
                c = f"{pr}transformer_cross_attention_layers.{i}."
                Wi, bi = sd[c + "multihead_attn.in_proj_weight"], sd[c + "multihead_attn.in_proj_bias"]
                ks.append(Wi[D:2 * D]); kb.append(bi[D:2 * D]); vs.append(Wi[2 * D:]); vb.append(bi[2 * D:])
            w[f"pr.lvl{l}.k.w"], w[f"pr.lvl{l}.k.b"] = W(torch.cat(ks, 0)), Fp(torch.cat(kb, 0))
            w[f"pr.lvl{l}.v.w"], w[f"pr.lvl{l}.v.b"] = W(torch.cat(vs, 0)), Fp(torch.cat(vb, 0))
        norm("pr.dn", pr + "decoder_norm")
        w["pr.query_embed"] = Fp(sd[pr + "query_embed.weight"])
        w["pr.level_embed"] = Fp(sd[pr + "level_embed.weight"])
        for name, n in (("mask_embed", 3), ("SEG_proj", 2), ("CLASS_proj", 2), ("REGION_proj", 2)):
            for j in range(n):
                lin(f"pr.{name}{j}", f"{pr}{name}.layers.{j}", small=True)

    # ======================================================================================= small host tables
    def _pos_embed(self, Hh, Ww):
        """PositionEmbeddingSine(normalize=True) as an (H*W, D) fp32 token table (position_encoding.py:29-52).
        Input-independent, so computed once per resolution on the host with the reference's formula."""
        key = ("pos", Hh, Ww)
        if key not in self._cache:
            npf = self.cfg.md_hidden // 2
            y = torch.arange(1, Hh + 1, dtype=torch.float32)[:, None].expand(Hh, Ww)
            x = torch.arange(1, Ww + 1, dtype=torch.float32)[None, :].expand(Hh, Ww)
            eps, scale = 1e-6, 2 * math.pi
            y = y / (y[-1:, :] + eps) * scale
            x = x / (x[:, -1:] + eps) * scale
            dim_t = torch.arange(npf, dtype=torch.float32)
            dim_t = 10000.0 ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / npf)
            px, py = x[:, :, None] / dim_t, y[:, :, None] / dim_t
            px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
            py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
            self._cache[key] = torch.cat((py, px), dim=2).reshape(Hh * Ww, 2 * npf).contiguous().to(self.device)
        return self._cache[key]

    def _rope(self, L):
        """cos/sin (L, rot) fp32, computed as modeling_phi.py:75-88 does (emb = cat(freqs, freqs))."""
        key = ("rope", L)
        if key not in self._cache:
            rd = self.cfg.rotary_dim
            inv = 1.0 / (self.cfg.rope_theta ** (torch.arange(0, rd, 2, dtype=torch.float32) / rd))
            fr = torch.arange(L, dtype=torch.float32)[:, None] * inv[None]
            emb = torch.cat((fr, fr), -1)
            self._cache[key] = (emb.cos().contiguous().to(self.device), emb.sin().contiguous().to(self.device))
        return self._cache[key]

    # ======================================================================================= Swin + projector
    def _swin_dims(self):
        return [self.cfg.swin_embed_dim * (2 ** s) for s in range(len(self.cfg.swin_depths))]

    def swin(self, images):
        """swin_trans.py:608-633.  images (B,3,H,W) fp32 on device -> [(tokens (B*h*w, C) adt, h, w)] x 4 (post norm_i)."""
        o, w, cfg = self.ops, self.w, self.cfg
        B, _, Hi, Wi = images.shape
        ps, ws = cfg.swin_patch, cfg.swin_window
        if (self.c_stages and self.x3 and self.fuse_split and ws == 12 and getattr(o.lib, "records", None) is None and not (H._DEBUG_BOUNDS or o.debug_bounds)
                and all(c % 8 == 0 and c <= 2048 and c == 32 * h for c, h in zip(self._swin_dims(), cfg.swin_heads))
                and all((f"swin{s}.{b}.fc1.bnd" in w) and (cfg.swin_mlp_ratio * c) % 8 == 0 for s, (d_, c) in enumerate(zip(cfg.swin_depths, self._swin_dims())) for b in range(d_))):
            key = ("swin_desc",)                       # stage-level native call (csrc/stages.hip): the tower's ~180 launches from ONE ctypes call
            if key not in self._cache:
                self._cache[key] = o.swin_desc(w, cfg.swin_depths, cfg.swin_heads, self._swin_dims(), ps, ws, self.pe_kpad, cfg.swin_mlp_ratio, self.paired)
            return o.swin_forward(self._cache[key], images)
        cols = o.patch_im2col(images, ps, self.pe_kpad, out_dtype=self.adt)
        Hc, Wc = (Hi + ps - 1) // ps, (Wi + ps - 1) // ps
        x = o.gemm(cols, w["swin.pe.w"], w["swin.pe.b"], out_dtype=torch.float32)
        x = o.layernorm(x, w["swin.pe.ln.g"], w["swin.pe.ln.b"])
        outs = []
        for s, (depth, heads) in enumerate(zip(cfg.swin_depths, cfg.swin_heads)):
            nWh, nWw = (Hc + ws - 1) // ws, (Wc + ws - 1) // ws
            for b in range(depth):
                q = f"swin{s}.{b}."
                shift = 0 if b % 2 == 0 else ws // 2
                x3f = self.x3 and x.shape[-1] % 8 == 0 and x.shape[-1] <= 2048   # f16x3: norm1 / norm2 rows leave in split-f16 form
                if x3f:
                    xw = o.swin_window_gather_split(x, w[q + "n1.g"], w[q + "n1.b"], B, Hc, Wc, ws, shift)
                else:
                    xw = o.swin_window_gather(x, w[q + "n1.g"], w[q + "n1.b"], B, Hc, Wc, ws, shift, out_dtype=self.adt)
                qkv = o.gemm(xw, w[q + "qkv.w"], w[q + "qkv.b"], out_dtype=self.adt)
                if x3f and self.fuse_split and ws == 12 and isinstance(xw, H.SplitF16):      # f16x3: the output leaves as the projection GEMM's split operand
                    aw = o.window_attention_split(qkv, w[q + "rpb"], xw.inv_scale, w[q + "qkv.bnd"], B, nWh, nWw, heads, ws, shift)
                else:
                    aw = o.window_attention(qkv, w[q + "rpb"], B, nWh, nWw, heads, ws, shift)
                pw = o.gemm(aw, w[q + "proj.w"], w[q + "proj.b"], out_dtype=self.adt)
                if x3f:
                    x, h = o.swin_window_merge_ln_split(pw, x, w[q + "n2.g"], w[q + "n2.b"], B, Hc, Wc, ws, shift)
                elif x.shape[-1] % 8 == 0:
                    x, h = o.swin_window_merge_ln(pw, x, w[q + "n2.g"], w[q + "n2.b"], B, Hc, Wc, ws, shift, h_dtype=self.adt)
                else:
                    x = o.swin_window_merge(pw, x, B, Hc, Wc, ws, shift)
                    h = o.layernorm(x, w[q + "n2.g"], w[q + "n2.b"], out_dtype=self.adt)
                hs = self._gemm_act_split(h, q + "fc1", H.ACT_GELU) if x3f else None      # f16x3: gelu(fc1) leaves as fc2's split operand
                h = hs if hs is not None else o.gemm(h, w[q + "fc1.w"], w[q + "fc1.b"], act=H.ACT_GELU, out_dtype=self.adt)
                x = o.gemm(h, w[q + "fc2.w"], w[q + "fc2.b"], residual=x, out_dtype=torch.float32)
            outs.append((o.layernorm(x, w[f"swin.out{s}.g"], w[f"swin.out{s}.b"], out_dtype=self.adt), Hc, Wc))
            if s < len(cfg.swin_depths) - 1:
                xm = o.patch_merge_ln(x, w[f"swin{s}.ds.ln.g"], w[f"swin{s}.ds.ln.b"], B, Hc, Wc, out_dtype=self.adt)
                x = o.gemm(xm, w[f"swin{s}.ds.red.w"], out_dtype=torch.float32)
                Hc, Wc = (Hc + 1) // 2, (Wc + 1) // 2
        return outs

    def projector(self, res5, B, h, w_):
        """multimodal_projector/builder.py:365-375 (+ BasicBlock :85-111, conv2 applied twice). -> (B*n, hidden) fp32."""
        o, w = self.ops, self.w
        if self.adt == torch.bfloat16 and res5.shape[-1] % 64 == 0 and w["proj.c2.w"].shape[0] % 64 == 0:
            # implicit-GEMM convolutions (no im2col matrix in HBM)
            ho, wo = (h + 2 - 3) // 2 + 1, (w_ + 2 - 3) // 2 + 1
            y = o.conv2d_nhwc(res5, B, h, w_, w["proj.c1.w"], 3, 2, 1, bias=w["proj.c1.b"], act=H.ACT_RELU)
            y = o.conv2d_nhwc(y, B, ho, wo, w["proj.c2.w"], 3, 1, 1)
            ds = o.conv2d_nhwc(res5, B, h, w_, w["proj.ds.w"], 1, 2, 0, bias=w["proj.ds.b"])
            y = o.conv2d_nhwc(y, B, ho, wo, w["proj.c2f.w"], 3, 1, 1, bias=w["proj.c2f.b"], residual=ds,
                              act=H.ACT_RELU | H.ACT_POST_RESIDUAL)
            return o.gemm(y, w["proj.fc.w"], w["proj.fc.b"], out_dtype=torch.float32), ho * wo
        if (self.c_stages and self.x3 and res5.dtype == torch.float32 and res5.shape[-1] % 8 == 0 and w["proj.c1.w"].shape[0] % 8 == 0
                and getattr(o.lib, "records", None) is None and all(isinstance(w[k], H.SplitF16) for k in ("proj.c1.w", "proj.c2.w", "proj.c2f.w", "proj.ds.w", "proj.fc.w"))):
            key = ("proj_desc",)                       # stage-level native call (csrc/stages.hip)
            if key not in self._cache:
                self._cache[key] = o.projector_desc(w)
            return o.projector_forward(self._cache[key], res5, B, h, w_)
        im2col = o.im2col_split if (self.x3 and res5.shape[-1] % 8 == 0) else o.im2col_nhwc   # f16x3: patches straight into split form
        c1 = im2col(res5, B, h, w_, 3, 2, 1)
        y = o.gemm(c1, w["proj.c1.w"], w["proj.c1.b"], act=H.ACT_RELU, out_dtype=self.adt)
        ho, wo = (h + 2 - 3) // 2 + 1, (w_ + 2 - 3) // 2 + 1
        y = o.gemm(im2col(y, B, ho, wo, 3, 1, 1), w["proj.c2.w"], out_dtype=self.adt)
        ds = o.gemm(im2col(res5, B, h, w_, 1, 2, 0), w["proj.ds.w"], w["proj.ds.b"], out_dtype=self.adt)
        y = o.gemm(im2col(y, B, ho, wo, 3, 1, 1), w["proj.c2f.w"], w["proj.c2f.b"], residual=ds,
                   act=H.ACT_RELU | H.ACT_POST_RESIDUAL, out_dtype=self.adt)
        return o.gemm(y, w["proj.fc.w"], w["proj.fc.b"], out_dtype=torch.float32), ho * wo

    # ======================================================================================= token splicing (host ints)
    def _splice_plan(self, input_ids, attention_mask, n_img, class_name_ids, cls_indices, token_refer_id, n_regions,
                     want_cls, want_refer):
        """llava_phi.py:767-971 on host integers, VECTORISED (numpy; no per-token Python loop -- the reference walks the prompt token by
        token with `.item()` calls, LP:581-766): where every row of inputs_embeds comes from, plus the row sets used after the LLM (seg
        queries LP:1299-1316, class-name groups LP:552-565, refer span LP:972-978, regions LP:302-307).
        source ids: 0 = embed_tokens row, 1 = image token, 2 = seg_query row, 3 = region feature row.
        `_splice_plan_reference` is the per-token form it is tested against (tests/test_8_collate.py)."""
        ids_all = input_ids.cpu().numpy().astype(np.int64)
        B, T = ids_all.shape
        am_all = np.ones((B, T), bool) if attention_mask is None else attention_mask.cpu().numpy().astype(bool)   # LP accepts attention_mask=None
        NQ = self.cfg.md_queries

        def expand(starts, sizes):                       # concatenation of arange(starts[i], starts[i] + sizes[i])
            sizes = np.asarray(sizes, np.int64)
            tot = int(sizes.sum())
            if tot == 0:
                return np.zeros(0, np.int64)
            off = np.concatenate(([0], np.cumsum(sizes)[:-1]))
            return np.repeat(np.asarray(starts, np.int64) - off, sizes) + np.arange(tot)

        per = []
        reg_base = 0
        known = (IMAGE_TOKEN_INDEX, SEG_TOKEN_INDEX, CLS_TOKEN_INDEX, REGION_TOKEN_INDEX, REFER_TOKEN_INDEX)
        for b in range(B):
            ids = ids_all[b]
            neg = ids < 0
            if neg.any() and not np.isin(ids[neg], known).all():
                raise ValueError(f"unknown sentinel token id {int(ids[neg][~np.isin(ids[neg], known)][0])}")
            is_img, is_seg, is_cls = ids == IMAGE_TOKEN_INDEX, ids == SEG_TOKEN_INDEX, ids == CLS_TOKEN_INDEX
            is_reg, is_ref = ids == REGION_TOKEN_INDEX, ids == REFER_TOKEN_INDEX
            lens = np.ones(T, np.int64)
            lens[is_img] = n_img
            lens[is_seg] = NQ
            ctoks = gsz = None
            if class_name_ids is not None:               # unique_consecutive groups of cls_indices >= 0 (LP:566-574)
                ci = cls_indices[b].cpu().numpy().astype(np.int64)
                cn = class_name_ids[b].cpu().numpy().astype(np.int64)
                valid = ci >= 0
                start = valid & np.concatenate(([True], (ci[1:] != ci[:-1]) | ~valid[:-1]))
                ctoks = cn[valid]
                gsz = np.diff(np.concatenate((np.flatnonzero(start[valid]), [int(valid.sum())])))
                assert int(is_cls.sum()) == gsz.shape[0], "the number of <cls> tokens and class_embed needs to be same"      # LP:590-591
                lens[is_cls] = gsz
            refer_tok = None
            if token_refer_id is not None:
                refer_tok = token_refer_id[b].cpu().numpy().astype(np.int64)
                lens[is_ref] = refer_tok.shape[0]
            elif is_ref.any():
                raise ValueError("<refer> token without token_refer_id")
            n_reg = int(is_reg.sum())
            if n_regions is not None:
                assert n_reg == n_regions[b], "the number of <region> tokens and regions needs to be same"                  # LP:592-594
            pos = np.cumsum(lens) - lens
            Lb = int(lens.sum())
            sid = np.zeros(Lb, np.int32)
            srow = np.zeros(Lb, np.int32)
            txt = ~neg
            srow[pos[txt]] = ids[txt]
            r = expand(pos[is_img], lens[is_img])
            sid[r] = 1
            srow[r] = np.tile(np.arange(b * n_img, (b + 1) * n_img), int(is_img.sum()))
            seg_rows = expand(pos[is_seg], lens[is_seg])
            sid[seg_rows] = 2
            srow[seg_rows] = np.tile(np.arange(NQ), int(is_seg.sum()))
            cls_rows = np.zeros(0, np.int64)
            if ctoks is not None:
                cls_rows = expand(pos[is_cls], gsz)
                srow[cls_rows] = ctoks
            reg_rows = pos[is_reg]
            sid[reg_rows] = 3
            srow[reg_rows] = reg_base + np.arange(n_reg)
            reg_base += n_reg
            ref_rows = np.zeros(0, np.int64)
            if refer_tok is not None:
                ref_rows = expand(pos[is_ref], lens[is_ref])
                srow[ref_rows] = np.tile(refer_tok, int(is_ref.sum()))
            per.append((sid, srow, seg_rows, cls_rows, gsz if gsz is not None else np.zeros(0, np.int64), ref_rows, reg_rows, Lb))
        lens_b = [p[7] for p in per]
        L = max(lens_b)
        sid = np.full((B, L), -1, np.int32)
        srow = np.zeros((B, L), np.int32)
        kmask = np.zeros((B, L), np.uint8)
        for b, p in enumerate(per):
            Lb = lens_b[b]
            sid[b, :Lb] = p[0]
            srow[b, :Lb] = p[1]
            kmask[b, : Lb - T] = 1                                # LP:939-946 / LP:965-968
            kmask[b, Lb - T: Lb] = am_all[b]
        plan = {"L": L, "lens": lens_b, "sid": sid, "srow": srow, "kmask": kmask}

        def csr(rows_per_b, sizes_per_b):                        # CSR over rows of the flattened (B*L, H) hidden-state matrix
            rows = np.concatenate([r + b * L for b, r in enumerate(rows_per_b)]) if rows_per_b else np.zeros(0, np.int64)
            sizes = np.concatenate(sizes_per_b) if sizes_per_b else np.zeros(0, np.int64)
            return np.concatenate(([0], np.cumsum(sizes))).astype(np.int32), rows.astype(np.int32)
        plan["seg"] = csr([p[2] for p in per], [np.ones(p[2].shape[0], np.int64) for p in per])
        plan["cls"] = csr([p[3] for p in per], [p[4] for p in per]) if want_cls else None
        plan["n_cls"] = [int(p[4].shape[0]) for p in per]
        plan["refer"] = csr([p[5] for p in per], [np.asarray([p[5].shape[0]], np.int64) for p in per]) if want_refer else None
        plan["region"] = csr([p[6] for p in per], [np.ones(p[6].shape[0], np.int64) for p in per]) if n_regions is not None else None
        return plan

    def _splice_plan_reference(self, input_ids, attention_mask, n_img, class_name_ids, cls_indices, token_refer_id, n_regions,
                               want_cls, want_refer):
        """Per-token form of `_splice_plan` (the round-1 implementation, a direct restatement of LP:767-971): kept as the checker of the
        vectorised version."""
        ids_all = input_ids.tolist()
        am_all = attention_mask.to(torch.bool).tolist() if attention_mask is not None else [[True] * len(r) for r in ids_all]
        B, T = len(ids_all), len(ids_all[0])
        NQ = self.cfg.md_queries
        per = []
        reg_base = 0
        for b in range(B):
            sid: List[int] = []
            srow: List[int] = []
            seg_rows: List[int] = []
            cls_groups: List[List[int]] = []
            refer_rows: List[int] = []
            region_rows: List[int] = []
            class_tok = None
            if class_name_ids is not None:
                ci = cls_indices[b].tolist()
                cn = class_name_ids[b].tolist()
                class_tok = []
                prev = None
                for idx, tok in zip(ci, cn):                    # unique_consecutive groups with idx >= 0 (LP:566-574)
                    if idx < 0:
                        prev = None
                        continue
                    if idx != prev:
                        class_tok.append([])
                        prev = idx
                    class_tok[-1].append(tok)
            refer_tok = token_refer_id[b].tolist() if token_refer_id is not None else None
            cls_i = reg_i = 0
            for t in ids_all[b]:
                pos = len(sid)
                if t >= 0:
                    sid.append(0); srow.append(t)
                elif t == IMAGE_TOKEN_INDEX:
                    sid += [1] * n_img; srow += list(range(b * n_img, (b + 1) * n_img))
                elif t == SEG_TOKEN_INDEX:
                    sid += [2] * NQ; srow += list(range(NQ)); seg_rows += list(range(pos, pos + NQ))
                elif t == CLS_TOKEN_INDEX:
                    toks = class_tok[cls_i]
                    cls_i += 1
                    sid += [0] * len(toks); srow += toks; cls_groups.append(list(range(pos, pos + len(toks))))
                elif t == REGION_TOKEN_INDEX:
                    sid.append(3); srow.append(reg_base + reg_i); region_rows.append(pos)
                    reg_i += 1
                elif t == REFER_TOKEN_INDEX:
                    sid += [0] * len(refer_tok); srow += refer_tok; refer_rows += list(range(pos, pos + len(refer_tok)))
                else:
                    raise ValueError(f"unknown sentinel token id {t}")
            if n_regions is not None:
                assert reg_i == n_regions[b], "the number of <region> tokens and regions needs to be same"   # LP:592-594
                reg_base += n_regions[b]
            if class_tok is not None:
                assert cls_i == len(class_tok), "the number of <cls> tokens and class_embed needs to be same"  # LP:590-591
            per.append((sid, srow, seg_rows, cls_groups, refer_rows, region_rows))
        lens = [len(p[0]) for p in per]
        L = max(lens)
        sid = np.full((B, L), -1, np.int32)
        srow = np.zeros((B, L), np.int32)
        kmask = np.zeros((B, L), np.uint8)
        for b, p in enumerate(per):
            Lb = lens[b]
            sid[b, :Lb] = p[0]
            srow[b, :Lb] = p[1]
            kmask[b, : Lb - T] = 1                                # LP:939-946 / LP:965-968
            kmask[b, Lb - T: Lb] = am_all[b]
        plan = {"L": L, "lens": lens, "sid": sid, "srow": srow, "kmask": kmask}
        # row sets as CSR over rows of the flattened (B*L, H) hidden-state matrix
        def csr(groups_per_b):
            off, rows = [0], []
            for b, groups in enumerate(groups_per_b):
                for gr in groups:
                    rows += [b * L + r for r in gr]
                    off.append(len(rows))
            return np.asarray(off, np.int32), np.asarray(rows, np.int32)
        plan["seg"] = csr([[[r] for r in p[2]] for p in per])
        plan["cls"] = csr([p[3] for p in per]) if want_cls else None
        plan["n_cls"] = [len(p[3]) for p in per]
        plan["refer"] = csr([[p[4]] for p in per]) if want_refer else None
        plan["region"] = csr([[[r] for r in p[5]] for p in per]) if n_regions is not None else None
        return plan

    def _bucketed(self, plan, B):
        """The splice plan with the sequence length rounded up to a multiple of `len_bucket`: the extra positions are what the shorter
        prompts of a ragged batch already are (LP:939-946: zero embedding rows, key mask 0, behind every real token -- the causal mask
        alone keeps them out of every real row), the row sets are re-based to the new row stride, and the variable-length ones (class-name
        / refer / region rows) are padded to a multiple of `len_bucket` entries behind their last CSR offset (never read).  Nothing a
        kernel launch is sized by then depends on the exact prompt length: one captured graph serves the whole bucket."""
        q = int(self.len_bucket or 0)
        if q <= 1:
            return plan
        hit = plan.get("_bucketed")
        if hit is not None and hit[0] == q:
            return hit[1]
        L = plan["L"]
        Lp = (L + q - 1) // q * q
        out = dict(plan)
        out.pop("_bucketed", None)
        if Lp != L:
            def wide(a, fill):
                w_ = np.full((B, Lp), fill, a.dtype)
                w_[:, :L] = a
                return w_
            out["sid"], out["srow"], out["kmask"] = wide(plan["sid"], -1), wide(plan["srow"], 0), wide(plan["kmask"], 0)
            out["L"] = Lp
        for name in ("seg", "cls", "refer", "region"):
            if plan[name] is None:
                continue
            off, rows = plan[name]
            rows = (rows.astype(np.int64) // L * Lp + rows.astype(np.int64) % L).astype(np.int32)
            if name != "seg":                            # (seg: B * Q rows whatever the prompt)
                n = (rows.shape[0] + q - 1) // q * q
                rows = np.concatenate((rows, np.zeros(max(n, q) - rows.shape[0], np.int32)))
            out[name] = (off, rows)
        plan["_bucketed"] = (q, out)
        return out

    def _dev_i32(self, a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    # ======================================================================================= region pooling
    def region_points(self, region_masks_list, sampler):
        """context_cluster.py:345-356: per region mask, nonzero()/[H,W] then random repeat/subsample to n points.
        Host side (RNG + tiny index work); returns (total_regions, n, 2) fp32 (y,x) in [0,1) and regions per image."""
        pts, counts = [], []
        n = self.cfg.region_points
        for masks in region_masks_list:
            masks = masks.cpu()
            counts.append(int(masks.shape[0]))
            for m in masks:
                nz = _nonzero_2d(m)
                wh = torch.tensor([m.shape[0], m.shape[1]])[None]
                pts.append((nz / wh)[sampler(nz, n)].float())
        return (torch.stack(pts) if pts else torch.zeros(0, n, 2)), counts

    # ======================================================================================= Phi decoder
    def llm(self, embeds, key_mask, B, L):
        """PhiModel.forward (modeling_phi.py:343-396) on inputs_embeds (B*L, hidden) fp32; key_mask (B,L) u8."""
        o, w, cfg = self.ops, self.w, self.cfg
        Hd, I = cfg.hidden_size, cfg.intermediate_size
        cos, sin = self._rope(L)
        x = embeds
        # f16x3: gelu(fc1) (GEMM epilogue) and the attention output leave directly as the split-f16 A operand [attn | gelu(fc1)] of the
        # [dense | fc2] GEMM -- one operand buffer, one scale per row from a magnitude bound (psalm_gemm_x3_split)
        fuse_split = self.fuse_split and (Hd + I) % 64 == 0 and Hd % 8 == 0 and "llm0.bnd" in w and cfg.head_dim == 64 and cfg.rotary_dim == 32
        if fuse_split and self.c_stages and self.x3 and getattr(o.lib, "records", None) is None and not (H._DEBUG_BOUNDS or o.debug_bounds) \
                and key_mask.is_contiguous():
            key = ("phi_desc",)
            if key not in self._cache:                # (pointers into the weight arena: built once per model instance / replica)
                self._cache[key] = o.phi_desc([dict(w1=w[f"llm{i}.w1"], b1=w[f"llm{i}.b1"], w2=w[f"llm{i}.w2"], b2=w[f"llm{i}.b2"],
                                                    ln_g=w[f"llm{i}.ln.g"], ln_b=w[f"llm{i}.ln.b"], bnd=w[f"llm{i}.bnd"],
                                                    paired=self.paired.get(f"llm{i}", False)) for i in range(cfg.num_layers)],
                                              Hd, I, cfg.num_heads, cfg.head_dim, cfg.rotary_dim, cfg.layer_norm_eps, w["llm.final.g"], w["llm.final.b"])
            if self.llm_products != 3:
                o.x3_products(self.llm_products)
            try:
                return o.phi_forward(self._cache[key], embeds, key_mask, cos, sin, B, L)
            finally:
                if self.llm_products != 3:
                    o.x3_products(3)
        big = o.empty(B * L, 3 * Hd if fuse_split else 3 * Hd + I, dtype=self.adt)
        if fuse_split:
            a2 = o.empty(B * L, 2 * (Hd + I), dtype=torch.float16)
            inv2 = o.empty(B * L, dtype=torch.float32)
        fused = self.adt == torch.bfloat16           # residual projection + the NEXT layer's LayerNorm in one call (psalm_gemm_ln)
        if self.paired.get("llm0", False) and not fuse_split:
            raise H.PsalmHipError("the Phi fc1 rows are laid out for paired split-f16 stores but the fused operand hand-over is off "
                                  "(build the model with paired_split_stores=False)")
        if self.x3:                                  # f16x3: LayerNorm emits the [k|v|q|fc1] GEMM's split-f16 A operand directly
            h = o.layernorm_split(x, w["llm0.ln.g"], w["llm0.ln.b"], cfg.layer_norm_eps)[1]
        else:
            h = o.layernorm(x, w["llm0.ln.g"], w["llm0.ln.b"], cfg.layer_norm_eps, out_dtype=self.adt)
        if self.llm_products != 3:
            o.x3_products(self.llm_products)             # thread-local launch-time policy of the split-f16 GEMMs: reset below
        try:
            return self._llm_layers(x, h, big, a2 if fuse_split else None, inv2 if fuse_split else None, fuse_split, fused, cos, sin, key_mask, B, L)
        finally:
            if self.llm_products != 3:
                o.x3_products(3)

    def _llm_layers(self, x, h, big, a2, inv2, fuse_split, fused, cos, sin, key_mask, B, L):
        o, w, cfg = self.ops, self.w, self.cfg
        Hd, I = cfg.hidden_size, cfg.intermediate_size
        for i in range(cfg.num_layers):
            last = i == cfg.num_layers - 1
            ng, nb = (w["llm.final.g"], w["llm.final.b"]) if last else (w[f"llm{i + 1}.ln.g"], w[f"llm{i + 1}.ln.b"])
            if fuse_split:
                o.gemm_x3_split(h, w[f"llm{i}.w1"], w[f"llm{i}.b1"], H.ACT_GELU_NEW, a2, inv2, w[f"llm{i}.bnd"], split_col_off=Hd,
                                split_col_start=3 * Hd, act_col_start=3 * Hd, out=big, global_rows=True,
                                paired=self.paired.get(f"llm{i}", False))
                o.causal_attention_split(big, 2 * Hd, 0, Hd, a2, inv2, 0, cos, sin, key_mask, B, L, cfg.num_heads, cfg.head_dim,
                                         cfg.rotary_dim)
                if last or Hd % 64 != 0 or Hd > 2048:
                    x = o.gemm(H.SplitF16(a2, inv2, Hd + I), w[f"llm{i}.w2"], w[f"llm{i}.b2"], residual=x, out_dtype=torch.float32)
                    h = o.layernorm(x, ng, nb, cfg.layer_norm_eps, out_dtype=torch.float32) if last else \
                        o.layernorm_split(x, ng, nb, cfg.layer_norm_eps)[1]
                else:                                 # residual GEMM + the next layer's LayerNorm + its split: one pass after the K slices
                    x, h, _ = o.gemm_x3_ln_split(H.SplitF16(a2, inv2, Hd + I), w[f"llm{i}.w2"], w[f"llm{i}.b2"], x, ng, nb,
                                                 cfg.layer_norm_eps)
                continue
            o.gemm(h, w[f"llm{i}.w1"], w[f"llm{i}.b1"], act=H.ACT_GELU_NEW, act_col_start=3 * Hd, out=big)
            # columns: [k | v | q | gelu_new(fc1)];  attention output overwrites q in place
            o.causal_attention(big, 2 * Hd, 0, Hd, big, 2 * Hd, cos, sin, key_mask, B, L, cfg.num_heads, cfg.head_dim,
                               cfg.rotary_dim)
            if fused:
                x, h = o.gemm_ln(big[:, 2 * Hd:], w[f"llm{i}.w2"], w[f"llm{i}.b2"], x, ng, nb, cfg.layer_norm_eps,
                                 ln_dtype=torch.float32 if last else self.adt)
            else:
                x = o.gemm(big[:, 2 * Hd:], w[f"llm{i}.w2"], w[f"llm{i}.b2"], residual=x, out_dtype=torch.float32)
                if self.x3 and not last:
                    h = o.layernorm_split(x, ng, nb, cfg.layer_norm_eps)[1]
                else:
                    h = o.layernorm(x, ng, nb, cfg.layer_norm_eps, out_dtype=torch.float32 if last else self.adt)
        return h

    # ======================================================================================= pixel decoder (one image)
    def pixel_decoder(self, feats):
        """msdeformattn.py:268-315 for ONE image.  feats = [(tokens (h*w, C), h, w)] res2..res5.
        Returns mask_features (H2*W2, mask_dim) in the GEMM-weight dtype, [level tokens (h*w, D) fp32] x 3, sizes."""
        o, w, cfg = self.ops, self.w, self.cfg
        D, G, M = cfg.md_hidden, cfg.md_gn_groups, cfg.md_heads
        levels = [feats[3], feats[2], feats[1]]
        shapes = [(h, w_) for _, h, w_ in levels]
        starts = [0]
        for h, w_ in shapes[:-1]:
            starts.append(starts[-1] + h * w_)
        S = starts[-1] + shapes[-1][0] * shapes[-1][1]
        key = ("lvlpos", tuple(shapes))
        if key not in self._cache:
            self._cache[key] = torch.cat([self._pos_embed(h, w_) + w["pd.level_embed"][l][None] for l, (h, w_) in enumerate(shapes)], 0).contiguous()
        lvl_pos = self._cache[key]
        if (self.c_stages and self.x3 and self.fuse_split and D % 8 == 0 and D <= 2048 and D == 32 * M and cfg.md_levels == 3 and cfg.md_points == 4
                and getattr(o.lib, "records", None) is None and not (H._DEBUG_BOUNDS or o.debug_bounds)
                and all(t.dtype == torch.float32 and t.shape[-1] % 8 == 0 for t, _, _ in feats)
                and all(f"pd.enc{i}.l1.bnd" in w for i in range(cfg.md_enc_layers)) and cfg.md_enc_ffn % 8 == 0 and cfg.md_mask_dim % 8 == 0):
            dkey = ("pd_desc",)                        # stage-level native call (csrc/stages.hip): ~60 launches from ONE ctypes call
            if dkey not in self._cache:
                self._cache[dkey] = o.pd_desc(w, D, G, M, cfg.md_enc_layers, cfg.md_enc_ffn, cfg.md_mask_dim, [int(t.shape[-1]) for t, _, _ in feats], self.paired)
            mf, ms_all = o.pixel_decoder_forward(self._cache[dkey], [(t.contiguous(), h, w_) for t, h, w_ in feats], lvl_pos)
            return mf, [ms_all[starts[l]: starts[l] + h * w_] for l, (h, w_) in enumerate(shapes)], shapes, (feats[0][1], feats[0][2])
        src = o.empty(S, D, dtype=torch.float32)
        for i, (tok, h, w_) in enumerate(levels):
            t = o.gemm(tok, w[f"pd.ip{i}.w"], w[f"pd.ip{i}.b"], out_dtype=self.adt)
            # GroupNorm writes straight into this level's rows of the level-concatenated token buffer
            o.groupnorm_nhwc(t, w[f"pd.ip{i}.gn.g"], w[f"pd.ip{i}.gn.b"], 1, h * w_, G, out=src[starts[i]: starts[i] + h * w_])
        dual = self.adt == torch.bfloat16        # keep a bf16 copy of the fp32 token stream as the GEMM A operand
        if dual:                                 # bf16 copy of the GroupNorm output for the first layer's value projection
            key = ("zrow", D)
            if key not in self._cache:
                self._cache[key] = o.zeros(1, D, dtype=torch.float32)
            src_a = o.add_bcast(src, self._cache[key], out_dtype=self.adt)
        else:
            src_a = src
        qin = o.add_bcast(src, lvl_pos, out_dtype=self.adt)
        for i in range(cfg.md_enc_layers):
            q_ = f"pd.enc{i}."
            value = o.gemm(src_a, w[q_ + "value.w"], w[q_ + "value.b"], out_dtype=self.adt)
            ow = o.gemm(qin, w[q_ + "ow.w"], w[q_ + "ow.b"], out_dtype=torch.float32)
            att = o.msda_fused(value.view(1, S, D), shapes, starts, ow.view(1, S, -1), M, out_dtype=self.adt).view(S, D)
            if self.x3 and D % 8 == 0 and D <= 2048:
                # f16x3: both LayerNorms hand the fp32 stream AND the following GEMMs' split-f16 operands over in one pass
                # (linear1 input; next layer's value input = src and offset / weight input = src + pos)
                src, src_s, _ = o.layernorm_split(o.gemm(att, w[q_ + "out.w"], w[q_ + "out.b"], residual=src, out_dtype=torch.float32),
                                                  w[q_ + "n1.g"], w[q_ + "n1.b"], want_y=True)
                hdd = self._gemm_act_split(src_s, q_ + "l1", H.ACT_RELU)
                if hdd is None:
                    hdd = o.gemm(src_s, w[q_ + "l1.w"], w[q_ + "l1.b"], act=H.ACT_RELU, out_dtype=self.adt)
                more = i + 1 < cfg.md_enc_layers
                src, src_a, qin = o.layernorm_split(o.gemm(hdd, w[q_ + "l2.w"], w[q_ + "l2.b"], residual=src, out_dtype=torch.float32),
                                                    w[q_ + "n2.g"], w[q_ + "n2.b"], want_y=True, want_split=more, add=lvl_pos if more else None)
                continue
            mid_a = o.empty(S, D, dtype=self.adt) if dual else None
            src = o.layernorm(o.gemm(att, w[q_ + "out.w"], w[q_ + "out.b"], residual=src, out_dtype=torch.float32),
                              w[q_ + "n1.g"], w[q_ + "n1.b"], out2=mid_a)
            hdd = o.gemm(mid_a if dual else src, w[q_ + "l1.w"], w[q_ + "l1.b"], act=H.ACT_RELU, out_dtype=self.adt)
            src_a = o.empty(S, D, dtype=self.adt) if dual else None
            nxt = dual and i + 1 < cfg.md_enc_layers              # next layer's query input (src + pos) from the same pass
            qin = o.empty(S, D, dtype=self.adt) if nxt else None
            src = o.layernorm(o.gemm(hdd, w[q_ + "l2.w"], w[q_ + "l2.b"], residual=src, out_dtype=torch.float32),
                              w[q_ + "n2.g"], w[q_ + "n2.b"], out2=src_a, add=lvl_pos if nxt else None, out3=qin)
            if not dual:
                src_a = src
                if i + 1 < cfg.md_enc_layers:
                    qin = o.add_bcast(src, lvl_pos, out_dtype=self.adt)
        ms = [src[starts[l]: starts[l] + h * w_] for l, (h, w_) in enumerate(shapes)]
        tok2, H2, W2 = feats[0]
        lat = o.gemm(tok2, w["pd.adapter.w"], w["pd.adapter.b"], out_dtype=self.adt)
        lat = o.groupnorm_nhwc(lat, w["pd.adapter.gn.g"], w["pd.adapter.gn.b"], 1, H2 * W2, G, relu=True, out_dtype=torch.float32)
        hs, ws_ = shapes[-1]
        y = o.upsample_add_nhwc(lat, ms[-1], 1, hs, ws_, H2, W2, out_dtype=self.adt)
        if self.adt == torch.bfloat16 and D % 64 == 0:
            y = o.conv2d_nhwc(y, 1, H2, W2, w["pd.layer.w"], 3, 1, 1, bias=w["pd.layer.b"])
        else:
            cols = o.im2col_split(y, 1, H2, W2, 3, 1, 1) if (self.x3 and D % 8 == 0) else o.im2col_nhwc(y, 1, H2, W2, 3, 1, 1)
            y = o.gemm(cols, w["pd.layer.w"], w["pd.layer.b"], out_dtype=self.adt)
        y = o.groupnorm_nhwc(y, w["pd.layer.gn.g"], w["pd.layer.gn.b"], 1, H2 * W2, G, relu=True, out_dtype=self.adt)
        mf = o.gemm(y, w["pd.mf.w"], w["pd.mf.b"], out_dtype=self.wdt)
        return mf, ms, shapes, (H2, W2)

    # ======================================================================================= predictor (one image)
    def _predictor_native_ok(self, mf):
        """the part of the native-predictor condition that is known before the LLM has run (psalm_predictor_kv / psalm_predictor_forward, csrc/stages.hip)"""
        o, cfg = self.ops, self.cfg
        D, nh, Q, nlev = cfg.md_hidden, cfg.md_heads, cfg.md_queries, cfg.md_levels
        return (self.c_stages and self.x3 and Q <= 128 and D == 32 * nh and D % 8 == 0 and nlev <= 3 and cfg.md_mask_dim % 8 == 0 and cfg.md_dim_ff % 8 == 0
                and getattr(o.lib, "records", None) is None and mf.dtype == torch.float32)

    def _predictor_desc(self, shapes):
        o, w, cfg = self.ops, self.w, self.cfg
        dkey = ("pr_desc",)                            # stage-level native call (csrc/stages.hip): ~200 launches from ONE ctypes call
        if dkey not in self._cache:
            self._cache[dkey] = o.pr_desc(w, cfg.md_hidden, cfg.md_heads, cfg.md_queries, cfg.md_dec_layers, cfg.md_levels, cfg.md_dim_ff, cfg.md_mask_dim)
        prpos = []
        for l in range(cfg.md_levels):
            h, w_ = shapes[l]
            key = ("prpos", l, h, w_)
            if key not in self._cache:
                self._cache[key] = (self._pos_embed(h, w_) + w["pr.level_embed"][l][None]).contiguous()
            prpos.append(self._cache[key])
        return self._cache[dkey], prpos

    def predictor_kv(self, ms, shapes, mf, mf_size, n_reg=0, slot=0):
        """The LLM-independent front of `predictor` for one image -- the level K / V projections and the mask-feature split -- as its own native call
        (psalm_predictor_kv), so that it can run on the side stream behind the pixel decoder, beside the LLM.  Returns a handle for `predictor(kv=...)`, or
        None when the op-by-op path will run."""
        if not self._predictor_native_ok(mf):
            return None
        desc, prpos = self._predictor_desc(shapes)
        cont = lambda t: t if t.is_contiguous() else t.contiguous()       # noqa: E731
        ms_c, mf_c = [cont(t) for t in ms], cont(mf)
        return (self.ops.predictor_kv(desc, ms_c, shapes, prpos, mf_c, mf_size, n_reg, slot), ms_c, mf_c)

    def predictor(self, ms, shapes, mf, mf_size, seg_query, SEG_emb=None, class_emb=None, region_emb=None, kv=None):
        """mask2former_transformer_decoder.py:596-693 (+ heads :695-762) for ONE image.
        seg_query (Q, D) fp32; *_emb (n, D) in the GEMM-weight dtype.  kv: `predictor_kv`'s handle (the K / V front already issued)."""
        o, w, cfg = self.ops, self.w, self.cfg
        D, nh, Q = cfg.md_hidden, cfg.md_heads, cfg.md_queries
        nl, nlev = cfg.md_dec_layers, cfg.md_levels
        H2, W2 = mf_size
        qe = w["pr.query_embed"]
        # bf16 mode: attention on the matrix cores (split-KV kernel).  It takes V transposed, which the value projection
        # produces directly by swapping the GEMM operands (V^T = W_v . X^T, bias along rows).
        mfma = self.adt == torch.bfloat16 and all((h * w_) % 8 == 0 for h, w_ in shapes)
        if (self._predictor_native_ok(mf) and seg_query.dtype == torch.float32
                and all(e is None or (e.dtype == torch.float32 and e.data_ptr() % 16 == 0) for e in (SEG_emb, class_emb, region_emb))):
            desc, prpos = self._predictor_desc(shapes)
            cont = lambda t: t if t is None or t.is_contiguous() else t.contiguous()       # noqa: E731
            n_reg = int(region_emb.shape[0]) if region_emb is not None else 0
            if kv is not None and kv[0][3] == n_reg:
                handle, ms_c, mf_c = kv
            else:
                handle, ms_c, mf_c = None, [cont(t) for t in ms], cont(mf)
            masks, cls_l, seg_l, reg_l = o.predictor_forward(desc, ms_c, shapes, prpos, mf_c, mf_size, cont(seg_query),
                                                             cont(class_emb), cont(SEG_emb), cont(region_emb), kv=handle)
            return {"pred_masks": masks.view(Q, H2, W2), "pred_class_name_logits": cls_l, "pred_SEG_logits": seg_l, "pred_region_logits": reg_l}
        Kl, Vl = [], []
        for l in range(nlev):
            h, w_ = shapes[l]
            key = ("prpos", l, h, w_)
            if key not in self._cache:
                self._cache[key] = (self._pos_embed(h, w_) + w["pr.level_embed"][l][None]).contiguous()
            kin = o.add_bcast(ms[l], self._cache[key], out_dtype=self.adt)
            vin = o.add_bcast(ms[l], w["pr.level_embed"][l:l + 1], out_dtype=self.adt)
            Kl.append(o.gemm(kin, w[f"pr.lvl{l}.k.w"], w[f"pr.lvl{l}.k.b"], out_dtype=self.adt))
            if mfma:
                Vl.append(o.gemm(w[f"pr.lvl{l}.v.w"], vin, w[f"pr.lvl{l}.v.b"], act=H.ACT_BIAS_ROW, out_dtype=self.adt))   # (n*D, HW)
            else:
                Vl.append(o.gemm(vin, w[f"pr.lvl{l}.v.w"], w[f"pr.lvl{l}.v.b"], out_dtype=self.adt))

        def mlp(x, name, n, out_dtype):
            for j in range(n):
                last = j == n - 1
                x = o.gemm(x, w[f"pr.{name}{j}.w"], w[f"pr.{name}{j}.b"], act=H.ACT_NONE if last else H.ACT_RELU,
                           out_dtype=out_dtype if last else self.adt)
            return x

        mf_w = self._wop(mf)                                        # f16x3: the mask features are split once for the 10 head passes

        def mask_head(out):
            dec = o.layernorm(out, w["pr.dn.g"], w["pr.dn.b"], out_dtype=self.adt)
            me = mlp(dec, "mask_embed", 3, self.adt)
            return dec, o.gemm(me, mf_w, out_dtype=torch.float32)   # (Q, H2*W2) fp32 mask logits

        if mfma:
            Qp = (Q + 7) // 8 * 8
            key = ("pr.selfvt", Q)
            if key not in self._cache:                              # V^T of the self-attention: (D, Qp), pad columns stay zero
                self._cache[key] = o.zeros(D, Qp, dtype=self.adt)
            self_vt = self._cache[key]
        out = seg_query
        dec, masks = mask_head(out)
        out_q = o.add_bcast(out, qe, out_dtype=self.adt)                      # out + query_embed (bf16 operand of the q projections)
        for i in range(nl):
            l = i % nlev
            h, w_ = shapes[l]
            amask, flags = o.attn_mask(masks.view(1, Q, H2, W2), h, w_)
            j = i // nlev
            qp = o.gemm(out_q, w[f"pr{i}.cq.w"], w[f"pr{i}.cq.b"], out_dtype=self.adt)
            if mfma:
                a = o.mha_attention_t(qp, Kl[l][:, j * D:(j + 1) * D], Vl[l][j * D:(j + 1) * D], 1, Q, h * w_, nh, amask, flags)
            else:
                a = o.mha_attention(qp, Kl[l][:, j * D:(j + 1) * D], Vl[l][:, j * D:(j + 1) * D], 1, Q, h * w_, nh, amask, flags)
            out_a = o.empty(Q, D, dtype=self.adt) if mfma else None          # bf16 copy of the stream = next GEMM operand
            out_q = o.empty(Q, D, dtype=self.adt) if mfma else None          # ... and stream + query_embed, from the same LayerNorm pass
            out = o.layernorm(o.gemm(a, w[f"pr{i}.co.w"], w[f"pr{i}.co.b"], residual=out, out_dtype=torch.float32),
                              w[f"pr{i}.cn.g"], w[f"pr{i}.cn.b"], out2=out_a, add=qe if mfma else None, out3=out_q)
            if not mfma:
                out_q = o.add_bcast(out, qe, out_dtype=self.adt)
            qk = o.gemm(out_q, w[f"pr{i}.sqk.w"], w[f"pr{i}.sqk.b"], out_dtype=self.adt)
            if mfma:
                o.gemm(w[f"pr{i}.sv.w"], out_a, w[f"pr{i}.sv.b"], act=H.ACT_BIAS_ROW, out=self_vt[:, :Q])
                a = o.mha_attention_t(qk[:, :D], qk[:, D:], self_vt, 1, Q, Q, nh)
            else:
                v = o.gemm(out, w[f"pr{i}.sv.w"], w[f"pr{i}.sv.b"], out_dtype=self.adt)
                a = o.mha_attention(qk[:, :D], qk[:, D:], v, 1, Q, Q, nh)
            out = o.layernorm(o.gemm(a, w[f"pr{i}.so.w"], w[f"pr{i}.so.b"], residual=out, out_dtype=torch.float32),
                              w[f"pr{i}.sn.g"], w[f"pr{i}.sn.b"], out2=out_a)
            hdd = o.gemm(out_a if mfma else out, w[f"pr{i}.f1.w"], w[f"pr{i}.f1.b"], act=H.ACT_RELU, out_dtype=self.adt)
            out_q = o.empty(Q, D, dtype=self.adt) if mfma else None
            out = o.layernorm(o.gemm(hdd, w[f"pr{i}.f2.w"], w[f"pr{i}.f2.b"], residual=out, out_dtype=torch.float32),
                              w[f"pr{i}.fn.g"], w[f"pr{i}.fn.b"], add=qe if mfma else None, out3=out_q)
            if not mfma:
                out_q = o.add_bcast(out, qe, out_dtype=self.adt)
            dec, masks = mask_head(out)
        res = {"pred_masks": masks.view(Q, H2, W2), "pred_class_name_logits": None, "pred_SEG_logits": None,
               "pred_region_logits": None}
        if class_emb is not None:
            res["pred_class_name_logits"] = o.gemm(mlp(dec, "CLASS_proj", 2, self.adt), self._wop(class_emb), out_dtype=torch.float32)
        if SEG_emb is not None:
            res["pred_SEG_logits"] = o.gemm(mlp(dec, "SEG_proj", 2, self.adt), self._wop(SEG_emb), out_dtype=torch.float32)
        if region_emb is not None:     # einsum 'kd,ld->kl' (TD:744): (k, Q)
            res["pred_region_logits"] = o.gemm(region_emb.to(self.adt) if region_emb.dtype != self.adt else region_emb,
                                               self._wop(mlp(dec, "REGION_proj", 2, self.wdt)), out_dtype=torch.float32)
        return res

    # ======================================================================================= host preparation
    def _prepare(self, input_ids, attention_mask, images, seg_info, class_name_ids, class_name_embedding_indices, cls_indices,
                 token_refer_id, refer_embedding_indices, region_point_sampler, video: bool = False):
        """Everything of one call that is host integer / RNG work (llava_phi.py:767-971 token splicing, region point
        sampling context_cluster.py:345-356, crop boxes LP:1418-1423), packed into ONE byte blob so that a call costs a
        single host->device copy.  Returns (blob uint8 ndarray, layout {name: (offset, count, dtype)}, meta)."""
        cfg = self.cfg
        B, _, Hi, Wi = images.shape
        # Fast path: the SAME prompt / padding-mask tensor objects as in an earlier call, unmodified (object identity + torch's version
        # counter; the cache entry keeps them alive, so an id cannot be recycled) -> the packed blob is that call's.  An evaluation loop
        # over one prompt pays the hashing / mask scan below once, not per image.  Not for region prompts (their points are re-drawn).
        def ident(t):
            return None if t is None else ((id(t), t._version) if torch.is_tensor(t) else id(t))
        pms = [info.get("padding_mask") if isinstance(info, dict) else None for info in (seg_info or [])]
        extra = tuple((info.get("height"), info.get("width")) if isinstance(info, dict) else None for info in (seg_info or []))
        ikey = (tuple(images.shape), video, class_name_embedding_indices is not None, refer_embedding_indices is not None, extra) + tuple(
            ident(t) for t in [input_ids, attention_mask, class_name_ids, cls_indices] + list(token_refer_id or []) + pms)
        # (only torch tensors carry a version counter: a list / ndarray prompt edited in place would go unseen -> no fast path for those.
        #  Edits that bypass the counter -- `.data`, storage shared with numpy through torch.from_numpy -- are the caller's to avoid.)
        cacheable = all(t is None or torch.is_tensor(t) for t in
                        [input_ids, attention_mask, class_name_ids, cls_indices] + list(token_refer_id or []) + pms)
        hit = self._prep_cache.get(ikey) if cacheable else None
        if hit is not None:
            return hit[0]
        ps = cfg.swin_patch
        h5, w5 = (Hi + ps - 1) // ps, (Wi + ps - 1) // ps
        for _ in range(len(cfg.swin_depths) - 1):
            h5, w5 = (h5 + 1) // 2, (w5 + 1) // 2
        n_img = ((h5 + 2 - 3) // 2 + 1) * ((w5 + 2 - 3) // 2 + 1)                 # projector conv stride 2 (PJ:334-336)
        arrays = {}
        n_regions = None
        if bool((input_ids == REGION_TOKEN_INDEX).any()):
            pts, n_regions = self.region_points([(s["instances"].vp_region_masks if video else s["instances"].region_masks).tensor
                                                 for s in seg_info], region_point_sampler)                # LP:792 / LP:1664
            arrays["region_img"] = np.asarray([b for b, k in enumerate(n_regions) for _ in range(k)], np.int32)
            arrays["region_pts"] = np.ascontiguousarray(pts.numpy(), np.float32)
        # the splice plan depends only on the (small) integer prompt tensors: identical prompts -- every image of a panoptic /
        # semantic evaluation uses the same one -- reuse it
        pkey = (n_img, tuple(n_regions) if n_regions is not None else None, class_name_embedding_indices is not None,
                refer_embedding_indices is not None) + tuple(
            None if t is None else (tuple(t.shape), t.cpu().numpy().tobytes())
            for t in (input_ids, attention_mask, class_name_ids, cls_indices)) + (
            tuple(t.cpu().numpy().tobytes() for t in token_refer_id) if token_refer_id is not None else None,)
        plan = self._plan_cache.get(pkey)
        if plan is None:
            if len(self._plan_cache) >= 64:
                self._plan_cache.clear()
            plan = self._plan_cache[pkey] = self._splice_plan(
                input_ids, attention_mask, n_img, class_name_ids, cls_indices, token_refer_id, n_regions,
                class_name_embedding_indices is not None, refer_embedding_indices is not None)
        plan = self._bucketed(plan, B)
        arrays["sid"] = plan["sid"].reshape(-1)
        arrays["srow"] = plan["srow"].reshape(-1)
        arrays["kmask"] = plan["kmask"].reshape(-1)
        for name in ("seg", "cls", "refer", "region"):
            if plan[name] is not None:
                arrays[name + "_off"], arrays[name + "_rows"] = plan[name]
        post = []
        if seg_info is not None:
            div = cfg.size_divisibility
            Hpad, Wpad = (Hi + div - 1) // div * div, (Wi + div - 1) // div * div     # ImageList.from_tensors(images, 32), LP:1400
            for info in seg_info:
                pm = info.get("padding_mask")
                if pm is None:
                    oh, ow = Hi, Wi
                else:                                     # extent of the un-padded box (LP:1418-1423), via row / column projections
                    valid = ~np.asarray(pm.cpu() if torch.is_tensor(pm) else pm, dtype=bool)
                    rows, cols = np.flatnonzero(valid.any(1)), np.flatnonzero(valid.any(0))
                    oh, ow = int(rows[-1] - rows[0] + 1), int(cols[-1] - cols[0] + 1)
                post.append((Hpad, Wpad, oh, ow, int(info.get("height", Hi)), int(info.get("width", Wi))))
        layout, off = {}, 0
        for k, a in arrays.items():
            a = np.ascontiguousarray(a)
            arrays[k] = a
            layout[k] = (off, a.size, a.dtype.str)
            off = (off + a.nbytes + 15) // 16 * 16
        blob = np.zeros(max(off, 16), np.uint8)
        for k, a in arrays.items():
            o0 = layout[k][0]
            blob[o0:o0 + a.nbytes] = a.view(np.uint8).reshape(-1)
        meta = {"B": B, "L": plan["L"], "lens": plan["lens"], "n_img": n_img, "n_cls": tuple(plan["n_cls"]),
                "n_regions": tuple(n_regions) if n_regions is not None else None, "post": tuple(post),
                "img_shape": tuple(images.shape), "layout": tuple(sorted(layout.items())), "video": video}
        if n_regions is None and cacheable:
            if len(self._prep_cache) >= 16:
                self._prep_cache.clear()
            self._prep_cache[ikey] = ((blob, layout, meta), [input_ids, attention_mask, class_name_ids, cls_indices, token_refer_id, pms])
        return blob, layout, meta

    @staticmethod
    def _views(dev_blob, layout):
        _T = {"<i4": torch.int32, "|u1": torch.uint8, "<f4": torch.float32}
        out = {}
        for k, (off, n, dt) in layout.items():
            t = _T[dt]
            nb = n * torch.empty((), dtype=t).element_size()
            out[k] = dev_blob[off:off + nb].view(t)
        return out

    def _side_stream(self):
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    # ======================================================================================= device forward
    def _forward_device(self, images, dv, meta, stages: Optional[dict] = None, postprocess: bool = True, vp_images=None):
        """All device work of eval_seg (LP:1350-1466): only kernel launches on the current stream, no host round trip
        (so the whole call can be captured into one hipGraph).  images (B,3,H,W) fp32 on device; dv: device views of
        the prepared blob.  Returns per-image predictor outputs (postprocess=False) or result dicts with `_pending`."""
        o, w, cfg = self.ops, self.w, self.cfg
        B, L, n_img = meta["B"], meta["L"], meta["n_img"]
        feats = self.swin(images)
        res5, h5, w5 = feats[3]
        img_tok, n_img2 = self.projector(res5, B, h5, w5)
        assert n_img2 == n_img
        if stages is not None:
            stages.update(feats=feats, image_tokens=img_tok)
        region_feats, n_regions = None, meta["n_regions"]
        if n_regions is not None:                                                      # region pooling (LP:791-797)
            side = int(math.sqrt(n_img))
            R = sum(n_regions)
            pool_tok = img_tok
            if vp_images is not None:            # eval_video: pool from the previous frame's projector tokens (LP:1663-1670)
                vf = self.swin(vp_images)
                pool_tok, _ = self.projector(vf[3][0], B, vf[3][1], vf[3][2])
            region_feats = o.region_pool(pool_tok, dv["region_img"], dv["region_pts"].view(R, -1, 2), side, side, n_img)
        # ---- the pixel decoder needs only the Swin features, the LLM only the projector tokens: run them concurrently on two
        # HIP streams (the decoder's ~150 small, latency-bound kernels fill the gaps of the LLM's large GEMMs); joined
        # before the predictor.  Captured as a fork/join inside the hipGraph in graph mode.
        pd_out = [None] * B
        side = None
        if self.overlap_streams and not o.is_emu:
            side = self._side_stream()
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for b in range(B):
                    pd_out[b] = self.pixel_decoder([(tok[b * h * w_:(b + 1) * h * w_], h, w_) for tok, h, w_ in feats])
                    mf_, ms_, shapes_, mfs_ = pd_out[b]
                    # r06: the predictor's K / V front needs nothing from the LLM either: on the side stream too (~0.2 ms of launches off the critical path)
                    pd_out[b] = pd_out[b] + ((self.predictor_kv(ms_, shapes_, mf_, mfs_, n_regions[b] if n_regions else 0, slot=b) if self.kv_side else None),)
        embeds = o.gather_rows([w["embed"], img_tok, w["seg_query"], region_feats], dv["sid"], dv["srow"], cfg.hidden_size,
                               out_dtype=torch.float32)
        hidden = self.llm(embeds, dv["kmask"].view(B, L), B, L)
        if stages is not None:
            Lr = max(meta["lens"])                    # (L is the bucketed length: hand the stage tensors out at the real one)
            stages.update(inputs_embeds=embeds.view(B, L, -1)[:, :Lr], hidden_states=hidden.view(B, L, -1)[:, :Lr], lengths=meta["lens"])
        # ---- LLM states -> decoder embeddings (LP:1366-1390)
        Q = cfg.md_queries
        # (row-set means come out in the GEMM operand dtype: bf16 -> the skinny MFMA kernel instead of the converting path)
        seg_q = o.gemm(o.segment_mean(hidden, dv["seg_off"], dv["seg_rows"], out_dtype=self.adt), w["seg_query_projector.w"],
                       w["seg_query_projector.b"], out_dtype=torch.float32)
        cls_emb = seg_emb = reg_emb = None
        if "cls_off" in dv:
            cls_emb = o.gemm(o.segment_mean(hidden, dv["cls_off"], dv["cls_rows"], out_dtype=self.adt), w["class_name_projector.w"],
                             w["class_name_projector.b"], out_dtype=self.wdt)
        if "refer_off" in dv:
            seg_emb = o.gemm(o.segment_mean(hidden, dv["refer_off"], dv["refer_rows"], out_dtype=self.adt), w["SEG_token_projector.w"],
                             w["SEG_token_projector.b"], out_dtype=self.wdt)
        if "region_off" in dv:
            reg_emb = o.gemm(o.segment_mean(hidden, dv["region_off"], dv["region_rows"], out_dtype=self.adt), w["region_projector.w"],
                             w["region_projector.b"], out_dtype=self.adt)
        if stages is not None:
            stages.update(seg_query=seg_q.view(B, Q, -1), class_name_embedding=cls_emb, SEG_embedding=seg_emb,
                          region_embedding=reg_emb)
        # ---- per image: pixel decoder + predictor (+ post-processing)
        outs = []
        c0 = r0 = 0
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
            if not torch.cuda.is_current_stream_capturing():
                # the pixel decoder's outputs were allocated on the side stream and are read by the predictor on this one: tell the caching
                # allocator.  (Their blocks could only be re-issued to the NEXT image's side-stream work, which starts behind
                # `side.wait_stream(main)` -- ordered already; this makes it hold without that argument.)
                for po in pd_out:
                    if po is not None:
                        for t in [po[0]] + list(po[1]) + ([po[4][0][0]] if len(po) > 4 and po[4] is not None else []):
                            if torch.is_tensor(t):
                                t.record_stream(torch.cuda.current_stream())
        for b in range(B):
            if pd_out[b] is None:
                pd_out[b] = self.pixel_decoder([(tok[b * h * w_:(b + 1) * h * w_], h, w_) for tok, h, w_ in feats])
                mf_, ms_, shapes_, mfs_ = pd_out[b]
                pd_out[b] = pd_out[b] + (self.predictor_kv(ms_, shapes_, mf_, mfs_, n_regions[b] if n_regions else 0, slot=b),)
            mf, ms, shapes, mf_size, kv = pd_out[b]
            nc = meta["n_cls"][b]
            ce = cls_emb[c0:c0 + nc] if cls_emb is not None else None
            c0 += nc
            se = seg_emb[b:b + 1] if seg_emb is not None else None
            re = None
            if reg_emb is not None:
                re = reg_emb[r0:r0 + n_regions[b]]
                r0 += n_regions[b]
            r = self.predictor(ms, shapes, mf, mf_size, seg_q[b * Q:(b + 1) * Q], se, ce, re, kv=kv)
            if stages is not None:
                stages.setdefault("mask_features", []).append(mf)
                stages.setdefault("multi_scale_features", []).append(ms)
            if postprocess == "head":                    # graph capture: stop where the image's own geometry starts (see len_bucket)
                outs.append(self._post_head(r, meta["post"][b][0], meta["post"][b][1]))
            else:
                outs.append(self._postprocess(r, meta["post"][b]) if postprocess else r)
        return outs

    def forward_logits(self, input_ids, attention_mask, images, seg_info=None, class_name_ids=None,
                       class_name_embedding_indices=None, cls_indices=None, token_refer_id=None,
                       refer_embedding_indices=None, labels=None, region_point_sampler: Callable = default_region_point_sampler,
                       stages: Optional[dict] = None, vp_images=None):
        """Everything of eval_seg up to the predictor outputs (LP:1350-1398).  Returns a list (one per image) of
        dicts with pred_masks (Q,h,w) fp32, pred_class_name_logits (Q,C+1) / pred_SEG_logits (Q,1) / pred_region_logits (k,Q)."""
        images = images.to(self.device, torch.float32).contiguous()
        if vp_images is not None:
            vp_images = vp_images.to(self.device, torch.float32).contiguous()
        blob, layout, meta = self._prepare(input_ids, attention_mask, images, seg_info, class_name_ids, class_name_embedding_indices,
                                           cls_indices, token_refer_id, refer_embedding_indices, region_point_sampler,
                                           video=vp_images is not None)
        dv = self._views(torch.from_numpy(blob).to(self.device), layout)
        return self._forward_device(images, dv, meta, stages=stages, postprocess=False, vp_images=vp_images)

    # ======================================================================================= post-processing + eval_seg
    def _semantic(self, mflat, probsT, Kpad, want_mask_score=False):
        """class_name_semantic_inference (LP:402-406): (C, HW) = probs^T . sigmoid(masks) [+ the per-query mask scores, LP:443-444].
        bf16 mode: one fused pass over the mask logits (sigmoid -> LDS -> MFMA, score sums on the side); exact mode (or shapes
        outside the fused kernel's limits): fp32 sigmoid^T + fp32 GEMM + the separate mask-score reduction."""
        o = self.ops
        if (probsT.dtype == torch.bfloat16 or self.x3) and Kpad == 128 and probsT.shape[0] <= 160:
            return o.semantic_from_masks(mflat, probsT, want_mask_score=want_mask_score)    # bf16 MFMA / split-f16 (f16x3) fused pass
        sem = o.gemm(probsT, self._wop(o.sigmoid_transpose(mflat, Kpad, self.wdt)), out_dtype=torch.float32)
        return (sem, o.mask_scores(mflat)) if want_mask_score else sem

    def _post_head(self, r, Hpad, Wpad):
        """The part of llava_phi.py:1401-1466 that does not depend on the image's own geometry: the mask logits at the padded image size
        (LP:1401-1406) and the class softmax (LP:328,403,410).  Inside the captured graph."""
        o, cfg = self.ops, self.cfg
        Q = cfg.md_queries
        h = {"r": r, "mp0": o.resize_planes(r["pred_masks"], Hpad, Wpad)}                 # LP:1401-1406
        if self.seg_task in ("semantic", "instance", "panoptic"):
            Kpad = (Q + 63) // 64 * 64                                   # K of the semantic GEMM (direct-to-LDS path: K % 64 == 0)
            h["Kpad"] = Kpad
            h["soft"] = o.class_softmax(r["pred_class_name_logits"], Kpad, probsT_dtype=self.wdt)
        return h

    def _post_tail(self, h, sizes):
        """llava_phi.py:1418-1466 for one image, device side: crop to the un-padded box, resize to the original size, the task's
        inference function.  h: `_post_head`'s dict; sizes: (Hpad, Wpad, crop_h, crop_w, out_h, out_w) from _prepare.  Every launch here
        is sized by the image's own geometry, so this runs outside the captured graph (see `len_bucket` / `graph_tail`).
        With `c_stages` (default) the dozen launches are ONE native call, psalm_postprocess_<task> (csrc/stages.hip: the same op-level entries in
        the same order -- `_post_tail_ops` is that sequence issued from Python, kept as the test's reference and for the cases the native entry
        leaves out: bf16 / exact-fp32 class maps, vocabularies above 160 classes)."""
        if self._post_native_ok(h):
            return self._post_tail_native(h, sizes)
        return self._post_tail_ops(h, sizes)

    def _post_native_ok(self, h):
        if not (self.c_stages and self.precision in ("f16x3", "fp32") and self.cfg.md_queries <= 128 and getattr(self.ops.lib, "records", None) is None
                and h["mp0"].dtype == torch.float32):
            return False                               # (bench's per-launch event records attribute time to op-level entries: the op sequence then)
        if self.seg_task in ("semantic", "panoptic"):
            return self.x3 and h.get("Kpad") == 128 and h["r"]["pred_class_name_logits"].shape[1] - 1 <= 160
        return True

    def _post_tail_native(self, h, sizes):
        o, cfg, task, r = self.ops, self.cfg, self.seg_task, h["r"]
        Hpad, Wpad, oh, ow, height, width = sizes
        C1 = r["pred_class_name_logits"].shape[1] if task in ("semantic", "instance", "panoptic") else 0
        out = o.postprocess(task, sizes, pred_masks=None, mask_up=h["mp0"],
                            cls_logits=r["pred_class_name_logits"] if C1 else None,
                            seg_logits=r["pred_SEG_logits"] if task == "referring" else None,
                            region_logits=r["pred_region_logits"] if task == "region" else None,
                            is_thing=self._thing_dev(C1 - 1) if task == "panoptic" else None,
                            obj_thr=cfg.object_mask_threshold, overlap_thr=cfg.overlap_threshold)
        res = {"_hw": (height, width), "_crop": (oh, ow), "mask_pred": out["mask_pred"]}
        if task == "semantic":
            res["sem_seg"] = out["sem_seg"]
            res["_pending"] = ("semantic",)
        elif task == "instance":
            res["_pending"] = ("instance", out["scores"], out["classes"], out["query"], out["counts"][0:1], out["inst_masks"], out["boxes"])
        elif task == "panoptic":
            res["sem_seg"] = out["sem_seg"]
            res["_pending"] = ("panoptic", out["scores"], out["classes"], out["query"], out["counts"], out["inst_masks"], out["pan"], out["boxes"])
        elif task == "referring":
            res["_pending"] = ("referring", out["scores"], out["query"], out["inst_masks"], out["boxes"])
        else:
            res["_pending"] = ("region", out["scores"], out["inst_masks"], out["boxes"])
        return res

    def _post_tail_ops(self, h, sizes):
        """`_post_tail` as op-level calls issued from Python (see there)."""
        o, cfg = self.ops, self.cfg
        Q = cfg.md_queries
        Hpad, Wpad, oh, ow, height, width = sizes
        r, mp = h["r"], h["mp0"]
        task = self.seg_task
        resize_after = (oh, ow, height, width) != (Hpad, Wpad, Hpad, Wpad)
        if resize_after and task != "semantic":                                       # sem_seg_postprocess_before_inference, LP:301,1427
            mp = o.resize_planes(mp, height, width, crop=(oh, ow))                    # sem_seg_postprocess, LP:1427-1429
        mh, mw = int(mp.shape[1]), int(mp.shape[2])
        HW = mh * mw
        mflat = mp.view(Q, HW)
        res = {"_hw": (height, width), "_crop": (oh, ow)}
        if task == "semantic":                                                        # semantic only, post-processed AFTER inference
            C1 = r["pred_class_name_logits"].shape[1]
            probs, probsT, score, label = h["soft"]
            sem = self._semantic(mflat, probsT, h["Kpad"]).view(C1 - 1, mh, mw)                                    # LP:402-406
            res["sem_seg"] = o.resize_planes(sem, height, width, crop=(oh, ow)) if resize_after else sem           # LP:1437-1440
            res["_pending"] = ("semantic",)
            res["mask_pred"] = mp
            return res
        if task == "instance":                                                        # top-k instances, no thing filter (LP:428)
            C1 = r["pred_class_name_logits"].shape[1]
            probs = h["soft"][0]
            mscore = o.mask_scores(mflat)
            sc, cl, qq, cnt = o.topk_select(probs, C1 - 1, Q, None, mscore)
            res["_pending"] = ("instance", sc, o.to_i64(cl), o.to_i64(qq), cnt, o.binarize_gather(mp, Q, qq, cnt), o.zeros(sc.shape[0], 4))
            res["mask_pred"] = mp
            return res
        if task == "panoptic":
            C1 = r["pred_class_name_logits"].shape[1]
            probs, probsT, score, label = h["soft"]
            sem, mscore = self._semantic(mflat, probsT, h["Kpad"], want_mask_score=True)                     # LP:402-406, 443-444
            res["sem_seg"] = sem.view(C1 - 1, height, width)
            thing = self._thing_dev(C1 - 1)
            counts = o.zeros(2 + 3 * Q, dtype=torch.int32)      # [instances kept, segments, segments_info (Q,3)]: ONE device-to-host copy
            sc, cl, qq, cnt = o.topk_select(probs, C1 - 1, Q, thing, mscore, count_out=counts[0:1])          # LP:407-447
            inst_masks = o.binarize_gather(mp, Q, qq, cnt)
            pan, pinfo, ninfo = o.panoptic(mp, score, label, thing, C1 - 1, cfg.object_mask_threshold, cfg.overlap_threshold,
                                           info_out=counts[2:].view(Q, 3), ninfo_out=counts[1:2])
            # (LongTensor labels / indices and the all-zero pred_boxes of `Instances` are made HERE, by this library's cast / memset:
            #  _finalize only slices -- no framework kernel per image)
            res["_pending"] = ("panoptic", sc, o.to_i64(cl), o.to_i64(qq), counts, inst_masks, pan, o.zeros(sc.shape[0], 4))
        elif task == "referring":
            mscore = o.mask_scores(mflat)
            sc, cl, qq, cnt = o.topk_select(r["pred_SEG_logits"], 1, Q, None, mscore, apply_sigmoid=True)    # LP:308-324
            inst_masks = o.binarize_gather(mp, Q, qq, cnt)
            res["_pending"] = ("referring", sc, o.to_i64(qq), inst_masks, o.zeros(inst_masks.shape[0], 4))
        elif task == "region":
            mscore = o.mask_scores(mflat)
            scores = o.region_scores(r["pred_region_logits"], mscore)                                        # LP:387-400
            inst_masks = o.binarize_gather(mp, Q)
            res["_pending"] = ("region", scores, inst_masks, o.zeros(Q, 4))
        else:
            raise NotImplementedError(f"seg_task {task}")
        res["mask_pred"] = mp
        return res

    def _postprocess(self, r, sizes):
        """llava_phi.py:1401-1466 for one image, device side.  r: predictor outputs; sizes: (Hpad, Wpad, crop_h, crop_w,
        out_h, out_w) from _prepare."""
        return self._post_tail(self._post_head(r, sizes[0], sizes[1]), sizes)

    def _thing_dev(self, C):
        key = ("thing", C, tuple(int(bool(x)) for x in self.is_thing_list))
        if key not in self._cache:
            t = list(key[2]) + [0] * max(0, C - len(key[2]))
            self._cache[key] = torch.tensor(t[:C], dtype=torch.int32, device=self.device)
        return self._cache[key]

    def _finalize(self, res, info):
        """One host round trip per image: fetch the data-dependent counts and slice the padded result buffers."""
        res = dict(res)
        pend = res.pop("_pending")
        hw = res.pop("_hw")
        oh, ow = res.pop("_crop")
        if pend[0] == "semantic":
            return res
        if pend[0] == "instance":
            _, sc, cl, qq, cnt, inst_masks, boxes = pend
            n = int(cnt.item())
            res["instances"] = Instances(hw, pred_masks=inst_masks[:n], scores=sc[:n], pred_classes=cl[:n], query_index=qq[:n], pred_boxes=boxes[:n])
            return res
        if pend[0] == "panoptic":
            _, sc, cl, qq, counts, inst_masks, pan, boxes = pend
            hc = counts.cpu().tolist()                            # the image's one host round trip: both counts + the segment table
            n, ni = hc[0], hc[1]
            res["instances"] = Instances(hw, pred_masks=inst_masks[:n], scores=sc[:n], pred_classes=cl[:n], query_index=qq[:n], pred_boxes=boxes[:n])
            rows = [hc[2 + 3 * i: 5 + 3 * i] for i in range(ni)]
            res["panoptic_seg"] = (pan, [{"id": a, "isthing": bool(b), "category_id": c} for a, b, c in rows])
        elif pend[0] == "referring":
            _, sc, qq, inst_masks, boxes = pend
            res["instances"] = Instances(hw, pred_masks=inst_masks, scores=sc, query_index=qq, pred_boxes=boxes)
        else:
            _, scores, inst_masks, boxes = pend
            gt = info["instances"].gt_masks
            gt = gt.tensor if hasattr(gt, "tensor") else gt
            gt = gt.to(self.device, torch.float32).contiguous()
            res["gt"] = self.ops.resize_planes(gt, hw[0], hw[1], crop=(oh, ow))                              # LP:1458-1461
            res["instances"] = Instances(hw, pred_masks=inst_masks, scores=scores, pred_boxes=boxes)
        return res

    # ---- hipGraph execution: the ~600 launches of one call are captured once per input signature and replayed
    def _graph_key(self, meta):
        """What a captured launch sequence is specific to.  NOT in it (unless graph_tail): the per-image geometry `meta["post"]` beyond the
        padded image size (itself a function of img_shape); the sequence length and the row-set sizes enter bucketed (`_bucketed`)."""
        return (self.seg_task, self.precision, self.llm_products, meta["video"], meta["img_shape"], meta["L"], meta["n_cls"], meta["n_regions"],
                meta["post"] if self.graph_tail else len(meta["post"]),
                meta["layout"], tuple(int(bool(x)) for x in self.is_thing_list) if self.is_thing_list is not None else None)

    def _run_graphed(self, images, blob, layout, meta, vp_images=None):
        key = self._graph_key(meta)
        ent = self._graphs.get(key)
        host = torch.from_numpy(blob)
        st = self.graph_stats
        st["calls"] += 1
        if ent is None:                                    # first sighting: eager run (fills caches / workspaces, warms the allocator)
            while len(self._graphs) >= self.max_graphs:    # every captured graph owns its intermediates (GBs at 1024^2): bound them
                self._graphs.pop(next(iter(self._graphs)))  # (insertion order = oldest signature first)
            self._graphs[key] = {"seen": 1}
            st["eager"] += 1
            dv = self._views(host.to(self.device), layout)
            return self._forward_device(images, dv, meta, vp_images=vp_images)
        if "graph" not in ent:                             # second sighting: capture
            ent["images"] = images.clone()
            ent["vp"] = vp_images.clone() if vp_images is not None else None
            ent["blob"] = host.to(self.device)
            dv = self._views(ent["blob"], layout)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g):
                    ent["outs"] = self._forward_device(ent["images"], dv, meta, vp_images=ent["vp"],
                                                       postprocess=True if self.graph_tail else "head")
            except Exception:
                self._graphs.pop(key, None)                # a failed capture must not leave a half-built entry behind
                raise
            ent["graph"] = g
            st["captures"] += 1
        self.ops.copy_(ent["images"], images)
        if vp_images is not None:
            self.ops.copy_(ent["vp"], vp_images)
        ent["blob"].copy_(host)
        ent["graph"].replay()
        st["replays"] += 1
        copy = self.graph_outputs != "alias"
        if not self.graph_tail:
            # the tail's launches are queued behind the replay while the GPU is still inside the graph; its results are fresh buffers --
            # only `mask_pred` can still be the graph's own (an image that needs no crop / resize)
            outs = []
            for b, h in enumerate(ent["outs"]):
                res = self._post_tail(h, meta["post"][b])
                if copy and res["mask_pred"].data_ptr() == h["mp0"].data_ptr():
                    res["mask_pred"] = res["mask_pred"].clone()
                outs.append(res)
            return outs
        if not copy:
            return ent["outs"]

        def own(v):
            if torch.is_tensor(v):
                return v.clone()
            if isinstance(v, tuple):
                return tuple(own(x) for x in v)
            if isinstance(v, list):
                return [own(x) for x in v]
            if isinstance(v, dict):
                return {k: own(x) for k, x in v.items()}
            return v
        return own(ent["outs"])

    @torch.no_grad()
    def eval_seg(self, input_ids=None, attention_mask=None, past_key_values=None, inputs_embeds=None, labels=None,
                 use_cache=None, output_attentions=None, output_hidden_states=None, images=None, return_dict=None,
                 seg_info=None, class_name_ids=None, class_name_embedding_indices=None, cls_indices=None,
                 token_refer_id=None, refer_embedding_indices=None, is_thing_list=None,
                 region_point_sampler: Callable = default_region_point_sampler, vp_images=None):
        """Same keyword signature as the reference's PSALM.eval_seg (llava_phi.py:1317-1336).  Returns list[dict] with
        `sem_seg`, `instances`, `panoptic_seg` (panoptic) / `instances` (referring) / `instances`,`gt` (region), one entry
        per image (the reference stops after image 0, LP:1472).
        With use_graphs=True the results are private copies of the graph's output buffers unless `graph_outputs = "alias"` (then they
        are overwritten by the next eval_seg call with the same input signature)."""
        if self.seg_task == "panoptic":
            assert is_thing_list is not None, "is_thing_list need to be given"        # LP:1337-1339
            self.is_thing_list = is_thing_list
        images = images.to(self.device, torch.float32).contiguous()
        if vp_images is not None:
            vp_images = vp_images.to(self.device, torch.float32).contiguous()
        blob, layout, meta = self._prepare(input_ids, attention_mask, images, seg_info, class_name_ids,
                                           class_name_embedding_indices, cls_indices, token_refer_id, refer_embedding_indices,
                                           region_point_sampler, video=vp_images is not None)
        self._last_meta = meta                          # (bench.py: the bucketed vs real sequence length of the last call)
        if self.use_graphs and not self.ops.is_emu:
            results = self._run_graphed(images, blob, layout, meta, vp_images)
        else:
            dv = self._views(torch.from_numpy(blob).to(self.device), layout)
            results = self._forward_device(images, dv, meta, vp_images=vp_images)
        return [self._finalize(r, seg_info[b]) for b, r in enumerate(results)]

    def eval_video(self, input_ids=None, attention_mask=None, past_key_values=None, inputs_embeds=None, labels=None,
                   use_cache=None, output_attentions=None, output_hidden_states=None, images=None, vp_images=None,
                   return_dict=None, seg_info=None, class_name_ids=None, class_name_embedding_indices=None, cls_indices=None,
                   token_refer_id=None, refer_embedding_indices=None, is_thing_list=None,
                   region_point_sampler: Callable = default_region_point_sampler):
        """PSALMForDAVISEval.eval_video (llava_phi.py:1845-1998), same keyword signature: eval_seg whose <region> features are
        pooled from the previous frame `vp_images` at `seg_info[i]['instances'].vp_region_masks` (LP:1663-1670)."""
        return self.eval_seg(input_ids=input_ids, attention_mask=attention_mask, labels=labels, images=images, seg_info=seg_info,
                             class_name_ids=class_name_ids, class_name_embedding_indices=class_name_embedding_indices,
                             cls_indices=cls_indices, token_refer_id=token_refer_id, refer_embedding_indices=refer_embedding_indices,
                             is_thing_list=is_thing_list, region_point_sampler=region_point_sampler, vp_images=vp_images)
