"""End-to-end host orchestration (psalm_amd.model.PSALM) on a TINY architecture, kernels running in the host
emulation, against the CPU oracle on the same seeded weights/inputs.  Validates layouts, weight fusion/folding,
token splicing and the stage wiring without a GPU.  (The full-size model is checked on the GPU: test_9_e2e_gpu.py.)"""
import dataclasses

import pytest
import torch

from ops_backend import make_ops
from oracle import psalm_oracle as O
from psalm_amd.config import PsalmConfig
from psalm_amd.model import PSALM
from psalm_amd.synthetic import make_inputs, make_state_dict


def _rel(a, b):
    return ((a.float().cpu() - b.float()).abs().max() / b.float().abs().max().clamp(min=1e-6)).item()


@pytest.mark.parametrize("task,batch,size", [("panoptic", 1, 96), ("referring", 2, 96), ("region", 2, 96)])
def test_tiny_forward_logits_fp32(task, batch, size):
    cfg = PsalmConfig.tiny(task)
    sd = make_state_dict(cfg, seed=11)
    inputs = make_inputs(cfg, task, size=size, batch=batch, seed=3, num_classes=9)
    model = PSALM(cfg, sd, ops=make_ops("emu"), precision="fp32")
    torch.manual_seed(77)
    _, st = O.eval_seg(sd, cfg, return_stages=True, postprocess=False, **inputs)
    torch.manual_seed(77)
    stages = {}
    kw = {k: v for k, v in inputs.items() if k != "is_thing_list"}
    outs = model.forward_logits(stages=stages, **kw)
    B = batch
    for i, name in enumerate(("res2", "res3", "res4", "res5")):
        tok, h, w = stages["feats"][i]
        want = st[name].permute(0, 2, 3, 1).reshape(-1, st[name].shape[1])
        assert _rel(tok, want) < 2e-4, name
    assert _rel(stages["image_tokens"].view(B, -1, cfg.hidden_size), st["image_tokens"]) < 2e-4
    assert _rel(stages["inputs_embeds"], st["inputs_embeds"]) < 1e-5
    for b in range(B):
        Lb = st["lengths"][b]
        assert _rel(stages["hidden_states"][b, :Lb], st["hidden_states"][b, :Lb]) < 5e-4
    assert _rel(stages["seg_query"], st["seg_query"]) < 5e-4
    for b in range(B):
        mfw = st["mask_features"][b].permute(1, 2, 0).reshape(-1, cfg.md_mask_dim)
        assert _rel(stages["mask_features"][b], mfw) < 5e-4
        for l in range(3):
            want = st["multi_scale_features"][l][b].permute(1, 2, 0).reshape(-1, cfg.md_hidden)
            assert _rel(stages["multi_scale_features"][b][l], want) < 5e-4
        assert _rel(outs[b]["pred_masks"], st["pred_masks"][b]) < 2e-3
        if task == "panoptic":
            assert _rel(outs[b]["pred_class_name_logits"], st["pred_class_name_logits"][b]) < 2e-3
        if task == "referring":
            assert _rel(outs[b]["pred_SEG_logits"], st["pred_SEG_logits"][b]) < 2e-3
        if task == "region":
            assert _rel(outs[b]["pred_region_logits"], st["pred_region_logits"][b]) < 2e-3


@pytest.mark.parametrize("task,batch,pad", [("panoptic", 1, 0), ("referring", 2, 32), ("region", 1, 0)])
def test_tiny_eval_seg_postprocess_fp32(task, batch, pad):
    """Full eval_seg incl. post-processing kernels vs the oracle's post-processing of the oracle's logits."""
    cfg = PsalmConfig.tiny(task)
    size = 96
    sd = make_state_dict(cfg, seed=12)
    inputs = make_inputs(cfg, task, size=size, batch=batch, seed=4, num_classes=9, pad=pad)
    model = PSALM(cfg, sd, ops=make_ops("emu"), precision="fp32")
    torch.manual_seed(5)
    want = O.eval_seg(sd, cfg, **inputs)
    torch.manual_seed(5)
    got = model.eval_seg(**inputs)
    assert len(got) == batch
    for b in range(batch):
        g, w = got[b], want[b]
        assert _rel(g["mask_pred"], w["mask_pred"]) < 2e-3
        gi, wi = g["instances"], w["instances"]
        if task == "panoptic":
            assert (g["sem_seg"].argmax(0).cpu() == w["sem_seg"].argmax(0)).float().mean() > 0.999
            assert _rel(g["sem_seg"], w["sem_seg"]) < 2e-3
            gp, ginfo = g["panoptic_seg"]
            wp, winfo = w["panoptic_seg"]
            assert ginfo == winfo
            assert (gp.cpu() == wp).float().mean() > 0.999
            og = sorted(zip((-gi.scores.cpu()).tolist(), gi.pred_classes.cpu().tolist()))
            ow = sorted(zip((-wi.scores).tolist(), wi.pred_classes.tolist()))
            assert len(og) == len(ow)
            for (a, c1), (b_, c2) in zip(og, ow):
                assert abs(a - b_) < 1e-4 and c1 == c2
        elif task == "referring":
            assert (torch.sort(gi.scores.cpu()).values - torch.sort(wi.scores).values).abs().max() < 1e-4
            # masks: compare per query through query_index
            gm = torch.zeros_like(wi.pred_masks)
            gm[gi.query_index.cpu()] = gi.pred_masks.cpu()
            wm = torch.zeros_like(wi.pred_masks)
            wm[wi.query_index] = wi.pred_masks
            assert (gm != wm).float().mean() < 1e-3
        else:
            assert _rel(gi.scores, wi.scores) < 2e-3
            assert (gi.pred_masks.cpu() != wi.pred_masks).float().mean() < 1e-3
            assert _rel(g["gt"], w["gt"]) < 1e-5


def test_tiny_eval_seg_bf16_mode_fused_paths():
    """precision="bf16" end to end on the emulator (the mode bench.py measures): bf16 GEMM operands on the direct-to-LDS kernels,
    MFMA attention kernels, split-K + fused LayerNorm and -- with 72 queries, i.e. a 128-wide padded K -- the fused sigmoid / semantic / mask-score pass.  Tolerances are bf16-operand level against the fp32 oracle."""
    cfg = dataclasses.replace(PsalmConfig.tiny("panoptic"), md_queries=72)
    sd = make_state_dict(cfg, seed=12)
    inputs = make_inputs(cfg, "panoptic", size=128, batch=1, seed=4, num_classes=9)     # 4x4 / 8x8 / 16x16 levels: MFMA attention path
    model = PSALM(cfg, sd, ops=make_ops("emu"), precision="bf16")
    torch.manual_seed(5)
    w = O.eval_seg(sd, cfg, **inputs)[0]
    torch.manual_seed(5)
    g = model.eval_seg(**inputs)[0]
    assert _rel(g["mask_pred"], w["mask_pred"]) < 2e-2
    assert _rel(g["sem_seg"], w["sem_seg"]) < 0.1
    assert (g["sem_seg"].argmax(0).cpu() == w["sem_seg"].argmax(0)).float().mean() > 0.95
    assert (g["panoptic_seg"][0].cpu() == w["panoptic_seg"][0]).float().mean() > 0.95
    gi, wi = g["instances"], w["instances"]
    assert len(gi.scores) == len(wi.scores)
    assert (torch.sort(gi.scores.cpu()).values - torch.sort(wi.scores).values).abs().max() < 2e-2


def test_tiny_eval_video_vs_oracle():
    """eval_video (PSALMForDAVISEval, LP:1845-1998): region features pooled from the previous frame; fp32 mode vs the oracle.
    (The post-processing around it is eval_seg's; the full call is checked on the GPU against the reference-generated golden.)"""
    cfg = PsalmConfig.tiny("region")
    sd = make_state_dict(cfg, seed=12)
    inputs = make_inputs(cfg, "region", size=96, batch=1, seed=4, video=True)
    model = PSALM(cfg, sd, ops=make_ops("emu"), precision="fp32")
    kw = {k: v for k, v in inputs.items() if k != "is_thing_list"}
    torch.manual_seed(5)
    _, st = O.eval_seg(sd, cfg, return_stages=True, postprocess=False, **inputs)
    torch.manual_seed(5)
    a = model.forward_logits(**kw)[0]
    assert _rel(a["pred_region_logits"], st["pred_region_logits"][0]) < 2e-3
    assert _rel(a["pred_masks"], st["pred_masks"][0]) < 2e-3
    # the previous frame really is what the region tokens are pooled from: region logits change when it is dropped
    torch.manual_seed(5)
    b_ = model.forward_logits(**{k: v for k, v in kw.items() if k != "vp_images"})[0]
    assert (a["pred_region_logits"] - b_["pred_region_logits"]).abs().max() > 1e-3 * a["pred_region_logits"].abs().max()


@pytest.mark.parametrize("queries,size", [(12, 96), (72, 96)])
def test_tiny_eval_seg_f16x3_mode(queries, size):
    """precision="f16x3" (the qualifying fast mode: every GEMM in split-f16 arithmetic, everything else as the exact-fp32 mode) end to
    end on the emulator: fp32-class agreement with the oracle, i.e. the tolerances of the fp32-mode tests, not the bf16 mode's.
    72 queries (a 128-wide padded K) also takes the fused split-f16 semantic / mask-score pass and the MFMA-tiled attention sizes."""
    cfg = dataclasses.replace(PsalmConfig.tiny("panoptic"), md_queries=queries)
    sd = make_state_dict(cfg, seed=12)
    inputs = make_inputs(cfg, "panoptic", size=size, batch=1, seed=4, num_classes=9)
    model = PSALM(cfg, sd, ops=make_ops("emu"), precision="f16x3")
    torch.manual_seed(5)
    w = O.eval_seg(sd, cfg, **inputs)[0]
    torch.manual_seed(5)
    g = model.eval_seg(**inputs)[0]
    assert _rel(g["mask_pred"], w["mask_pred"]) < 1e-4
    assert _rel(g["sem_seg"], w["sem_seg"]) < 1e-4
    # labels: identical wherever the oracle's decision is not an exact tie (this tiny random model has pixels whose two best classes
    # differ by ~1e-11 at a scale of 2e-4, i.e. below one fp32 ulp of the sum: any summation order may pick either)
    top2 = w["sem_seg"].topk(2, 0).values
    decided = (top2[0] - top2[1]) > 1e-6 * w["sem_seg"].abs().max()
    same = g["sem_seg"].argmax(0).cpu() == w["sem_seg"].argmax(0)
    assert bool(same[decided].all()) and same.float().mean() >= 0.98
    assert torch.equal(g["panoptic_seg"][0].cpu(), w["panoptic_seg"][0])
    assert g["panoptic_seg"][1] == w["panoptic_seg"][1]
    gi, wi = g["instances"], w["instances"]
    assert len(gi.scores) == len(wi.scores)
    assert (torch.sort(gi.scores.cpu()).values - torch.sort(wi.scores).values).abs().max() < 1e-4


# (the panoptic / batch-1 data flow through the same fused kernels is compared with the oracle by test_tiny_eval_seg_f16x3_mode)
@pytest.mark.parametrize("task,batch,keys", [("referring", 2, ("pred_masks", "pred_SEG_logits"))])
def test_tiny_f16x3_fused_split_outputs_match_unfused(task, batch, keys):
    """The f16x3 mode's fused operand hand-over (GEMM epilogue / attention kernels emit the next GEMM's split-f16 operand under a
    bound-derived scale) against the same model with fuse_split = False (fp32 tensors + psalm_split_f16 passes, exact row-max scales):
    the two only differ in where hi + lo hits the f16 subnormal floor and in the split-K / LayerNorm summation order -> fp32 round-off."""
    cfg = PsalmConfig.tiny(task)
    sd = make_state_dict(cfg, seed=21)
    inputs = make_inputs(cfg, task, size=96, batch=batch, seed=6, num_classes=7)            # batch 2: ragged prompts, padded key mask
    model = PSALM(cfg, sd, ops=make_ops("emu"), precision="f16x3")
    assert model.fuse_split and model.so_paired and any(model.paired.values())
    kw = {k: v for k, v in inputs.items() if k != "is_thing_list"}
    torch.manual_seed(5)
    model_out_a = model.forward_logits(**kw)
    model.fuse_split = False                              # its fc1 / linear1 rows are permuted for the paired stores of the hand-over: refused
    with pytest.raises(Exception, match="paired"):
        model.forward_logits(**kw)
    model = PSALM(cfg, sd, ops=make_ops("emu"), precision="f16x3", paired_split_stores=False)
    assert not any(model.paired.values())
    model.fuse_split = False
    torch.manual_seed(5)
    model_out_b = model.forward_logits(**kw)
    a, b = model_out_a[0], model_out_b[0]
    for i in range(batch):
        for k in keys:
            assert _rel(a[k], b[k]) < 2e-5, (i, k)
        if i + 1 < batch:
            a, b = model_out_a[i + 1], model_out_b[i + 1]


def test_replica_shares_weights_owns_state_and_agrees():
    """PSALM.replica(): the second instance for another stream / host thread (bench.py `two_in_flight`) holds the SAME weight tensors, its
    own binding (workspaces), caches and graphs, and returns the first instance's results bit for bit."""
    cfg = PsalmConfig.tiny("panoptic")
    sd = make_state_dict(cfg, seed=12)
    inputs = make_inputs(cfg, "panoptic", size=64, batch=1, seed=4, num_classes=9)
    m = PSALM(cfg, sd, ops=make_ops("emu"), precision="f16x3")
    a = m.eval_seg(**inputs)[0]
    r = m.replica()
    assert r.w is m.w and r.paired is m.paired and r.ops is not m.ops and r.ops._ws is not m.ops._ws
    assert r._cache == {} and r._graphs == {} and r._prep_cache == {} and m._cache
    b = r.eval_seg(**inputs)[0]
    assert torch.equal(a["mask_pred"], b["mask_pred"]) and torch.equal(a["sem_seg"], b["sem_seg"])
    assert torch.equal(a["panoptic_seg"][0], b["panoptic_seg"][0]) and a["panoptic_seg"][1] == b["panoptic_seg"][1]


def test_llm_single_product_side_mode_changes_the_llm_only_and_leaves_the_default_alone():
    """PSALM(llm_products=1) -- BASELINE.json configs[4]'s reduced-precision LLM path as an opt-in side mode: the Phi GEMMs form one f16 product
    (hi.hi) instead of three.  On the tiny model: the vision tower's outputs are bit for bit those of the default (only the LLM's GEMMs
    change), the LLM's hidden states move by an f16-operand-sized amount (between 1e-5 and 2e-2 relative L2 -- stated, looser tolerance), and
    a default model evaluated afterwards on the same binding gives its own bits again (the setting is reset after the LLM)."""
    cfg = PsalmConfig.tiny("panoptic")
    sd = make_state_dict(cfg, seed=11)
    inputs = make_inputs(cfg, "panoptic", size=96, batch=1, seed=3, num_classes=9)
    kw = {k: v for k, v in inputs.items() if k != "is_thing_list"}
    ops = make_ops("emu")
    m3 = PSALM(cfg, sd, ops=ops, precision="f16x3")
    m1 = PSALM(cfg, sd, ops=ops, precision="f16x3", llm_products=1)
    s3, s1, s3b = {}, {}, {}
    o3 = m3.forward_logits(stages=s3, **kw)
    o1 = m1.forward_logits(stages=s1, **kw)
    o3b = m3.forward_logits(stages=s3b, **kw)
    assert torch.equal(s3["image_tokens"], s1["image_tokens"]) and torch.equal(s3["inputs_embeds"], s1["inputs_embeds"])
    h3, h1 = s3["hidden_states"].double(), s1["hidden_states"].double()
    rel = float((h3 - h1).norm() / h3.norm())
    assert 1e-5 < rel < 2e-2, rel
    assert torch.equal(s3["hidden_states"], s3b["hidden_states"]) and torch.equal(o3[0]["pred_masks"], o3b[0]["pred_masks"])
    rm = float((o3[0]["pred_masks"].double() - o1[0]["pred_masks"].double()).abs().max() / o3[0]["pred_masks"].abs().max())
    assert rm < 0.2, rm
    with pytest.raises(ValueError):
        PSALM(cfg, sd, ops=ops, precision="fp32", llm_products=1)


@pytest.mark.parametrize("task,batch", [("panoptic", 1), ("referring", 2), ("region", 1)])      # (ragged batch: referring; the GPU test runs full size)
def test_stage_level_calls_are_bitwise_the_op_by_op_sequence(task, batch):
    """psalm_swin_forward / psalm_phi_forward (csrc/stages.hip; SURVEY section 8(b): the stage-level C ABI behind the model API) issues the Phi decoder's launch
    sequence from native code -- ONE ctypes call instead of ~4 per layer.  Same launches, same order: the hidden states, and everything
    downstream, are bit for bit those of PSALM.llm's op-by-op Python sequence (c_stages = False), on a ragged batch too; the one-product side
    mode goes through it as well."""
    cfg = PsalmConfig.tiny(task)
    sd = make_state_dict(cfg, seed=11)
    inputs = make_inputs(cfg, task, size=96, batch=batch, seed=3, num_classes=9)
    kw = {k: v for k, v in inputs.items() if k != "is_thing_list"}
    ops = make_ops("emu")
    for prods in ((3, 1) if task == "panoptic" else (3,)):       # (the one-product side mode differs in the LLM stage only: one task covers it)
        m = PSALM(cfg, sd, ops=ops, precision="f16x3", llm_products=prods)
        assert m.c_stages
        sa, sb = {}, {}
        torch.manual_seed(77)                                     # (region prompts: the point sampling draws from the global RNG)
        oa = m.forward_logits(stages=sa, **kw)
        assert ("phi_desc",) in m._cache                          # the stage-level call ran
        m.c_stages = False
        torch.manual_seed(77)
        ob = m.forward_logits(stages=sb, **kw)
        assert torch.equal(sa["hidden_states"], sb["hidden_states"]), prods
        assert ("swin_desc",) in m._cache and ("proj_desc",) in m._cache        # ... the Swin tower's (psalm_swin_forward) and the projector's
        for (ta, ha, wa), (tb, hb, wb) in zip(sa["feats"], sb["feats"]):
            assert (ha, wa) == (hb, wb) and torch.equal(ta, tb)
        assert torch.equal(sa["image_tokens"], sb["image_tokens"])
        assert ("pd_desc",) in m._cache                              # ... the pixel decoder's (psalm_pixel_decoder_forward, one call per image)
        for ma, mb in zip(sa["mask_features"], sb["mask_features"]):
            assert torch.equal(ma, mb)
        for la, lb in zip(sa["multi_scale_features"], sb["multi_scale_features"]):
            assert all(torch.equal(x_, y_) for x_, y_ in zip(la, lb))
        assert ("pr_desc",) in m._cache                              # ... and the masked-attention decoder's (psalm_predictor_forward)
        for a, b in zip(oa, ob):
            assert torch.equal(a["pred_masks"], b["pred_masks"])
            for k in ("pred_class_name_logits", "pred_SEG_logits", "pred_region_logits"):
                assert (a[k] is None) == (b[k] is None) and (a[k] is None or torch.equal(a[k], b[k])), k
    # the library refuses a workspace that is too small instead of writing past it
    import ctypes
    from ctypes import c_long, c_void_p
    d = m._cache[("phi_desc",)]
    ops.lib.psalm_phi_forward_workspace.restype = c_long
    need = ops.lib.psalm_phi_forward_workspace(ctypes.byref(d), 1, 64)
    assert need > 0
    x = torch.zeros(64, cfg.hidden_size)
    km = torch.ones(1, 64, dtype=torch.uint8)
    cos, sin = m._rope(64)
    ws = torch.zeros(need + 256, dtype=torch.uint8)
    off = (-ws.data_ptr()) % 256
    rc = ops.lib.psalm_phi_forward(ctypes.byref(d), c_void_p(x.data_ptr()), c_void_p(km.data_ptr()), c_void_p(cos.data_ptr()), c_void_p(sin.data_ptr()),
                                   1, 64, c_void_p(x.data_ptr()), c_void_p(ws.data_ptr() + off), c_long(need - 1), None, c_long(0), None)
    assert rc != 0 and b"workspace" in ops.lib.psalm_last_error()


def _flatten_result(r):
    """every tensor of one eval_seg result, by name (Instances fields, the panoptic pair, plain tensors)"""
    out = {}
    for k, v in r.items():
        if torch.is_tensor(v):
            out[k] = v
        elif isinstance(v, tuple):                                   # panoptic_seg: (id map, segments_info)
            out[k + ".map"] = v[0]
            out[k + ".info"] = v[1]
        elif hasattr(v, "_fields") or hasattr(v, "__dict__"):
            for f, t in vars(v).items():
                if torch.is_tensor(t):
                    out[f"{k}.{f}"] = t
                elif isinstance(t, dict):
                    for f2, t2 in t.items():
                        if torch.is_tensor(t2):
                            out[f"{k}.{f2}"] = t2
    return out


@pytest.mark.parametrize("task,pad", [("panoptic", 0), ("panoptic", 11), ("semantic", 0), ("semantic", 11), ("instance", 11), ("referring", 0), ("referring", 11),
                                      ("region", 11)])
def test_native_postprocess_is_bitwise_the_op_sequence(task, pad):
    """psalm_postprocess_<task> (csrc/stages.hip): llava_phi.py:1401-1466 for one image as ONE native call -- the results of
    eval_seg are word for word those of PSALM._post_tail_ops, the same op-level entries issued from Python; with and without a crop / resize to the
    original size.  72 queries: the class-map tasks take the fused split-f16 pass (Q in (64, 128]), as the 100-query model does."""
    import dataclasses
    cfg = dataclasses.replace(PsalmConfig.tiny(task), md_queries=72)
    sd = make_state_dict(cfg, seed=21)
    inputs = make_inputs(cfg, task, size=96, batch=1, seed=6, num_classes=9, pad=pad)
    ops = make_ops("emu")
    m = PSALM(cfg, sd, ops=ops, precision="f16x3")
    calls = []
    real = ops.postprocess
    ops.postprocess = lambda *a, **k: (calls.append(a[0]), real(*a, **k))[1]
    try:
        torch.manual_seed(3)
        got = m.eval_seg(**inputs)
        assert calls == [task], calls                                    # the native entry ran
        m._post_native_ok = lambda h: False
        torch.manual_seed(3)
        want = m.eval_seg(**inputs)
        assert calls == [task]
    finally:
        ops.postprocess = real
    fa, fb = _flatten_result(got[0]), _flatten_result(want[0])
    assert set(fa) == set(fb) and len(fa) >= 1, (sorted(fa), sorted(fb))
    for k in fa:
        if torch.is_tensor(fa[k]):
            assert fa[k].shape == fb[k].shape and fa[k].dtype == fb[k].dtype and torch.equal(fa[k], fb[k]), k
        else:
            assert fa[k] == fb[k], k
