"""GPU parity tests proper: the HIP path (through the C ABI) vs
  (1) the golden vectors generated from the REFERENCE code (tests/golden/*.npz), full-size architecture;
  (2) the CPU oracle on the same seeded inputs (tiny architecture, all three tasks).
  (3) the CPU oracle at BASELINE.json's configurations 2 / 3 / 5 (full 24-layer model: 1024^2 panoptic, 640^2 x4 ragged referring,
      1024^2 x2 region prompts) in the headline mode "f16x3", at the north star's bar: mask IoU >= 0.999, labels >= 99.9 % identical.
Tolerances are stated per mode:
  fp32 mode : stage tensors within 1e-3 * absmax (fp32 round-off through ~100 layers), labels/masks >= 99.9 % identical
  f16x3 mode: the fp32 mode's tolerances (split-f16 GEMMs carry 22-bit operands; measured 2e-6 on the 1024^2 mask logits)
  bf16 mode : stage tensors within 6e-2 * absmax, mask IoU and label agreement reported and bounded below.
"""
import functools
import json
import os

import numpy as np
import pytest
import torch

from golden_util import RNG_SEED_AT_CALL, check_signature, load_case
from oracle import psalm_oracle as O
from psalm_amd.config import PsalmConfig
from psalm_amd.synthetic import make_inputs, make_state_dict

pytestmark = pytest.mark.gpu
EXACT = ("fp32", "f16x3")      # fp32-class modes: asserted at the fp32 tolerances (f16x3 = split-f16 GEMMs, 22-bit operands)
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_report.jsonl")


def _report(**kw):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(json.dumps(kw) + "\n")


def _nchw(tok, B, h, w):
    return tok.view(B, h, w, -1).permute(0, 3, 1, 2)


def _run_golden(name, precision):
    from psalm_amd.model import PSALM
    case, z = load_case(name)
    cfg = PsalmConfig(num_layers=case["layers"], seg_task=case["task"])
    sd = make_state_dict(cfg, seed=case["seed"])
    inputs = make_inputs(cfg, task=case["task"], size=case["size"], batch=case["batch"], seed=case["seed"], pad=case["pad"],
                         video=case.get("video", False), **({"geometry": [case["geometry"]]} if case.get("geometry") else {}))
    model = PSALM(cfg, sd, precision=precision)
    del sd
    torch.manual_seed(RNG_SEED_AT_CALL)
    stages = {}
    kw = {k: v for k, v in inputs.items() if k != "is_thing_list"}
    outs = model.forward_logits(stages=stages, **kw)
    torch.manual_seed(RNG_SEED_AT_CALL)
    results = model.eval_video(**inputs) if case.get("video") else model.eval_seg(**inputs)
    torch.cuda.synchronize()
    return case, z, cfg, stages, outs, results


def _stage_checks(z, case, cfg, stages, outs, rtol, tag):
    B = case["batch"]
    errs = {}
    for i, k in enumerate(("res2", "res3", "res4", "res5")):
        tok, h, w = stages["feats"][i]
        errs[k] = check_signature(z, k, _nchw(tok, B, h, w).contiguous(), rtol, what=tag)
    if not case.get("video"):          # (eval_video: the golden's projector signature belongs to the previous frame, LP:1665)
        errs["image_tokens"] = check_signature(z, "image_tokens", stages["image_tokens"], rtol, what=tag)
    # the LLM's own output (a Phi regression must not surface only as a downstream mask error).  Un-padded batches
    # only: behind a ragged batch's padding the reference's hidden states are whatever its masked attention leaves there.
    hs = stages["hidden_states"]
    if all(int(n) == hs.shape[1] for n in stages["lengths"]):
        errs["hidden_states"] = check_signature(z, "hidden_states", hs, rtol, what=tag)
    pm = torch.stack([o["pred_masks"] for o in outs])
    # mask logits sit behind the 24-layer LLM and the 9 discontinuous mask -> attention-mask feedback steps: in bf16 on random
    # weights they move 2-3x more than the feed-forward stages (measured 1.4e-2 .. 7e-2 of absmax over the five golden cases)
    errs["pred_masks"] = check_signature(z, "pred_masks", pm, rtol if rtol < 1e-2 else 2 * rtol, what=tag)
    mf = torch.stack([m.view(pm.shape[-2], pm.shape[-1], -1).permute(2, 0, 1) for m in stages["mask_features"]])
    errs["mask_features"] = check_signature(z, "mask_features", mf.contiguous(), rtol, what=tag)
    return errs, pm


@pytest.mark.parametrize("precision,rtol", [("fp32", 1e-3), ("f16x3", 1e-3), ("bf16", 6e-2)])
def test_golden_panoptic_512(precision, rtol):
    case, z, cfg, stages, outs, results = _run_golden("panoptic_512", precision)
    errs, pm = _stage_checks(z, case, cfg, stages, outs, rtol, precision + ":")
    cls = outs[0]["pred_class_name_logits"].cpu().numpy()
    gcls = z["pred_class_name_logits"][0]
    cls_err = float(np.abs(cls - gcls).max() / np.abs(gcls).max())
    r = results[0]
    sem_agree = float((r["sem_seg"].argmax(0).to(torch.uint8).cpu().numpy() == z["sem_seg_argmax"]).mean())
    pan, info = r["panoptic_seg"]
    pan_agree = float((pan.to(torch.uint8).cpu().numpy() == z["panoptic_ids"]).mean())
    info_same = [[s["id"], int(s["isthing"]), s["category_id"]] for s in info] == z["panoptic_info"].tolist()
    # per-query mask IoU of (pred_masks > 0) at stride 4 against the golden logits
    g = torch.from_numpy(z["pred_masks_s4"])[0] > 0
    c = pm[0, :, ::4, ::4].cpu() > 0
    inter = (g & c).flatten(1).sum(1).float()
    union = (g | c).flatten(1).sum(1).float()
    iou = torch.where(union > 0, inter / union.clamp(min=1), torch.ones_like(union))
    pix_agree = float((g == c).float().mean())
    _report(test="panoptic_512", precision=precision, stage_err=errs, cls_err=cls_err, sem_argmax_agree=sem_agree,
            panoptic_agree=pan_agree, panoptic_info_identical=info_same, mask_iou_mean=float(iou.mean()), mask_iou_min=float(iou.min()),
            mask_pixel_agree=pix_agree, n_instances=len(r["instances"]), n_segments=len(info))
    if precision in EXACT:
        assert cls_err < 1e-3 and sem_agree > 0.999 and pan_agree > 0.999 and info_same
        assert float(iou.mean()) > 0.999 and pix_agree > 0.9999
        gi = r["instances"]
        og = np.lexsort((gi.pred_classes.cpu().numpy(), -gi.scores.cpu().numpy()))
        ow = np.lexsort((z["inst_classes"], -z["inst_scores"]))
        assert len(og) == len(ow)
        np.testing.assert_allclose(gi.scores.cpu().numpy()[og], z["inst_scores"][ow], atol=2e-3)
        assert (gi.pred_classes.cpu().numpy()[og] == z["inst_classes"][ow]).all()
    else:
        # bf16 storage + bf16 MFMA on RANDOM weights: a 24-layer random transformer amplifies rounding noise, so class
        # logits move by several % and near-tied semantic labels flip (measured r1: cls 8e-2, labels 82 %, IoU 0.970).
        assert cls_err < 0.15 and sem_agree > 0.75 and float(iou.mean()) > 0.95 and pix_agree > 0.995


@pytest.mark.parametrize("precision,rtol", [("fp32", 1e-3), ("f16x3", 1e-3), ("bf16", 6e-2)])
def test_golden_referring_384_b2(precision, rtol):
    case, z, cfg, stages, outs, results = _run_golden("referring_384_b2", precision)
    errs, pm = _stage_checks(z, case, cfg, stages, outs, rtol, precision + ":")
    seg = torch.stack([o["pred_SEG_logits"] for o in outs]).cpu().numpy()
    seg_err = float(np.abs(seg - z["pred_SEG_logits"]).max() / np.abs(z["pred_SEG_logits"]).max())
    sc = np.sort(results[0]["instances"].scores.cpu().numpy())
    sc_err = float(np.abs(sc - np.sort(z["inst_scores"])).max())
    sc_mean = float(np.abs(sc - np.sort(z["inst_scores"])).mean())
    _report(test="referring_384_b2", precision=precision, stage_err=errs, seg_err=seg_err, score_err=sc_err, score_err_mean=sc_mean)
    assert len(results) == 2
    # bf16 (the mode that does NOT meet the north-star bar): single outputs move chaotically with every change of rounding -- the
    # score of one instance sat at 0.05 from the reference in r03 and at 0.2005 in r04 (a GELU expression evaluated with one explicit fma).  A bound
    # on the WORST of 100 scores therefore has to be loose (0.3 on [0, 1]); what stays meaningful is the distribution: the mean distance between
    # the sorted score lists (the previous `max < 0.5` alone was close to vacuous)
    assert seg_err < (rtol if precision in EXACT else 0.15) and sc_err < (2e-3 if precision in EXACT else 0.3)
    assert precision in EXACT or sc_mean < 0.03, sc_mean


@pytest.mark.parametrize("precision,rtol", [("fp32", 1e-3), ("f16x3", 1e-3), ("bf16", 6e-2)])
def test_golden_region_384(precision, rtol):
    case, z, cfg, stages, outs, results = _run_golden("region_384", precision)
    errs, pm = _stage_checks(z, case, cfg, stages, outs, rtol, precision + ":")
    rl = torch.cat([o["pred_region_logits"].reshape(-1) for o in outs]).cpu().numpy()
    rl_err = float(np.abs(rl - z["pred_region_logits"]).max() / np.abs(z["pred_region_logits"]).max())
    sc_err = float(np.abs(results[0]["instances"].scores.cpu().numpy() - z["inst_scores"]).max())
    _report(test="region_384", precision=precision, stage_err=errs, region_logit_err=rl_err, score_err=sc_err)
    assert rl_err < (rtol if precision in EXACT else 0.15)
    if precision in EXACT:
        assert sc_err < 2e-3
    check_signature(z, "gt", results[0]["gt"], 1e-5)


@pytest.mark.parametrize("task,batch", [("panoptic", 1), ("referring", 2), ("region", 2)])
def test_tiny_vs_oracle_on_gpu(task, batch):
    from psalm_amd.model import PSALM
    cfg = PsalmConfig.tiny(task)
    sd = make_state_dict(cfg, seed=12)
    inputs = make_inputs(cfg, task, size=96, batch=batch, seed=4, num_classes=9)
    torch.manual_seed(5)
    want = O.eval_seg(sd, cfg, **inputs)
    for precision, tol in (("fp32", 2e-3), ("f16x3", 2e-3), ("bf16", 8e-2)):
        model = PSALM(cfg, sd, precision=precision)
        torch.manual_seed(5)
        got = model.eval_seg(**inputs)
        for b in range(batch):
            a, w = got[b]["mask_pred"].cpu(), want[b]["mask_pred"]
            err = float((a - w).abs().max() / w.abs().max())
            _report(test=f"tiny_{task}", precision=precision, image=b, mask_pred_err=err)
            assert err < tol, (precision, err)


@pytest.mark.parametrize("precision", ["f16x3", "fp32", "bf16"])
@pytest.mark.parametrize("task,batch", [("panoptic", 1), ("referring", 2)])
def test_graph_replay_is_bitwise_eager(task, batch, precision):
    """use_graphs=True: 1st call eager, 2nd call captures the launch sequence into a hipGraph, later calls replay it with new inputs copied
    into the graph's static buffers.  Every mode -- the headline one with its side-stream overlap of pixel decoder and LLM included -- gives
    bit-identical results.  r05: the calls also differ in what the graph key no longer holds -- crop box, original size, and (referring)
    sentence lengths inside one length bucket -- and still replay ONE graph."""
    from psalm_amd.model import PSALM
    cfg = PsalmConfig.tiny(task)
    sd = make_state_dict(cfg, seed=12)
    eager = PSALM(cfg, sd, precision=precision)
    graphed = PSALM(cfg, sd, precision=precision, use_graphs=True)
    geos = [None, None, [(96, 96, 96, 96)], [(64, 96, 40, 61), (96, 72, 120, 90)], [(80, 96, 50, 60)], None]
    lens = [None, None, None, [7, 12], [5, 14], [9, 9]]     # prompts 4 / 5 tokens + image + refer + seg: all inside one 32-token bucket
    for call, seed in enumerate((4, 5, 6, 7, 8, 4)):        # different pixels / token ids each call
        kw = {"geometry": geos[call]} if geos[call] else {}
        if task == "referring" and lens[call]:
            kw["refer_lens"] = lens[call]
        inputs = make_inputs(cfg, task, size=96, batch=batch, seed=seed, num_classes=9, **kw)
        want = eager.eval_seg(**inputs)
        got = graphed.eval_seg(**inputs)
        torch.cuda.synchronize()
        for b in range(batch):
            assert got[b]["mask_pred"].shape == want[b]["mask_pred"].shape
            assert torch.equal(got[b]["mask_pred"], want[b]["mask_pred"]), (call, b)
            assert torch.equal(got[b]["instances"].scores, want[b]["instances"].scores)
            assert torch.equal(got[b]["instances"].pred_masks, want[b]["instances"].pred_masks)
            if task == "panoptic":
                assert torch.equal(got[b]["sem_seg"], want[b]["sem_seg"])
                assert torch.equal(got[b]["panoptic_seg"][0], want[b]["panoptic_seg"][0])
                assert got[b]["panoptic_seg"][1] == want[b]["panoptic_seg"][1]
    st = graphed.graph_stats
    assert len(graphed._graphs) == 1 and st["captures"] == 1 and st["eager"] == 1 and st["replays"] == 5, (st, len(graphed._graphs))
    # default graph_outputs="copy": a kept result is NOT overwritten by the next call; "alias" returns the graph's own buffers
    inputs_a = make_inputs(cfg, task, size=96, batch=batch, seed=4, num_classes=9)
    inputs_b = make_inputs(cfg, task, size=96, batch=batch, seed=5, num_classes=9)
    kept = graphed.eval_seg(**inputs_a)[0]["mask_pred"]
    snap = kept.clone()
    graphed.eval_seg(**inputs_b)
    torch.cuda.synchronize()
    assert torch.equal(kept, snap)
    graphed.graph_outputs = "alias"
    kept = graphed.eval_seg(**inputs_a)[0]["mask_pred"]
    snap = kept.clone()
    graphed.eval_seg(**inputs_b)
    torch.cuda.synchronize()
    assert not torch.equal(kept, snap)
    # graph_tail = True (the r01-r04 form: tail inside the graph, geometry in its key): same bits
    tailed = PSALM(cfg, sd, precision=precision, use_graphs=True)
    tailed.graph_tail = True
    for _ in range(3):
        got = tailed.eval_seg(**inputs_a)
    want = eager.eval_seg(**inputs_a)
    torch.cuda.synchronize()
    assert all(torch.equal(got[b]["mask_pred"], want[b]["mask_pred"]) and torch.equal(got[b]["instances"].pred_masks, want[b]["instances"].pred_masks)
               for b in range(batch))


def test_golden_panoptic_1024_box_full_model_vs_the_reference_itself():
    """The headline configuration against the REFERENCE's own outputs, not through the oracle: full 24-layer model,
    1024 x 1024 canvas holding the 768 x 1024 box of a 480 x 640 original, results at 480 x 640 (tests/golden/make_golden.py
    panoptic_1024_box: the reference's `eval_seg` run on the CPU).  Headline arithmetic; stage signatures at 1e-3 of their range, the label maps
    and segments in full."""
    case, z, cfg, stages, outs, results = _run_golden("panoptic_1024_box", "f16x3")
    errs, pm = _stage_checks(z, case, cfg, stages, outs, 1e-3, "f16x3:")
    r = results[0]
    assert tuple(r["sem_seg"].shape[-2:]) == (480, 640) == tuple(r["panoptic_seg"][0].shape)
    sem_agree = float((r["sem_seg"].argmax(0).to(torch.uint8).cpu().numpy() == z["sem_seg_argmax"]).mean())
    pan, info = r["panoptic_seg"]
    pan_agree = float((pan.to(torch.uint8).cpu().numpy() == z["panoptic_ids"]).mean())
    info_same = [[s["id"], int(s["isthing"]), s["category_id"]] for s in info] == z["panoptic_info"].tolist()
    g = torch.from_numpy(z["pred_masks_s8"])[0]
    c = pm[0, :, ::8, ::8].cpu()
    rel = float((g - c).abs().max() / g.abs().max())
    flips = int(((g > 0) != (c > 0)).sum())
    cls_err = float(np.abs(outs[0]["pred_class_name_logits"].cpu().numpy() - z["pred_class_name_logits"][0]).max() / np.abs(z["pred_class_name_logits"]).max())
    _report(test="panoptic_1024_box", precision="f16x3", stage_err=errs, cls_err=cls_err, sem_argmax_agree=sem_agree, panoptic_agree=pan_agree,
            panoptic_info_identical=info_same, mask_logit_rel_err_stride8=rel, flipped_stride8=flips, n_segments=len(info))
    # (the reference's own path -- MSDA through grid_sample, torch's operator order -- and the oracle agree to ~1e-5 .. 2e-4 of a stage's range:
    #  the tolerance the oracle itself is pinned at, tests/test_4_oracle_golden.py; r05e on the MI355X: 5.8e-5, 0 flipped signs, labels identical)
    assert rel < 2e-4 and flips <= 2 and cls_err < 2e-4
    assert sem_agree > 0.9999 and pan_agree > 0.9999 and info_same
    gi = r["instances"]
    og = np.lexsort((gi.pred_classes.cpu().numpy(), -gi.scores.cpu().numpy()))
    ow = np.lexsort((z["inst_classes"], -z["inst_scores"]))
    assert len(og) == len(ow) and (gi.pred_classes.cpu().numpy()[og] == z["inst_classes"][ow]).all()
    np.testing.assert_allclose(gi.scores.cpu().numpy()[og], z["inst_scores"][ow], atol=2e-4)
    assert tuple(gi.pred_masks.shape[-2:]) == (480, 640)
    # (pixel counts of the instance masks: within 2 pixels -- the bound the oracle itself is pinned at against the reference, test_4; r05f: 1 pixel in 2 of 43)
    assert np.abs(gi.pred_masks.flatten(1).sum(1).cpu().numpy()[og] - z["inst_mask_area"][ow]).max() <= 2


@pytest.mark.parametrize("precision,rtol", [("fp32", 1e-3), ("f16x3", 1e-3), ("bf16", 6e-2)])
def test_golden_semantic_384(precision, rtol):
    case, z, cfg, stages, outs, results = _run_golden("semantic_384", precision)
    errs, pm = _stage_checks(z, case, cfg, stages, outs, rtol, precision + ":")
    sem = results[0]["sem_seg"]
    agree = float((sem.argmax(0).to(torch.uint8).cpu().numpy() == z["sem_seg_argmax"]).mean())
    _report(test="semantic_384", precision=precision, stage_err=errs, sem_argmax_agree=agree)
    assert tuple(sem.shape[-2:]) == (case["size"] - case["pad"],) * 2
    if precision in EXACT:
        check_signature(z, "sem_seg", sem, 1e-3)
        assert agree > 0.999
    else:
        assert agree > 0.75


@pytest.mark.parametrize("precision,rtol", [("fp32", 1e-3), ("f16x3", 1e-3), ("bf16", 6e-2)])
def test_golden_instance_384(precision, rtol):
    case, z, cfg, stages, outs, results = _run_golden("instance_384", precision)
    errs, pm = _stage_checks(z, case, cfg, stages, outs, rtol, precision + ":")
    inst = results[0]["instances"]
    assert len(inst) == cfg.md_queries
    sc = np.sort(inst.scores.cpu().numpy())
    sc_err = float(np.abs(sc - np.sort(z["inst_scores"])).max())
    _report(test="instance_384", precision=precision, stage_err=errs, score_err=sc_err)
    if precision in EXACT:
        og = np.lexsort((inst.pred_classes.cpu().numpy(), -inst.scores.cpu().numpy()))
        ow = np.lexsort((z["inst_classes"], -z["inst_scores"]))
        np.testing.assert_allclose(inst.scores.cpu().numpy()[og], z["inst_scores"][ow], atol=2e-3)
        assert (inst.pred_classes.cpu().numpy()[og] == z["inst_classes"][ow]).all()
    else:
        # random-weight bf16 (see test_golden_panoptic_512): individual near-tied candidates swap, the score distribution holds
        assert float(np.abs(sc - np.sort(z["inst_scores"])).mean()) < 0.05 and sc_err < 0.6


@pytest.mark.parametrize("precision,rtol", [("fp32", 1e-3), ("f16x3", 1e-3), ("bf16", 6e-2)])
def test_golden_video_region_384(precision, rtol):
    """eval_video (PSALMForDAVISEval, LP:1845-1998) against the golden generated by the reference class."""
    case, z, cfg, stages, outs, results = _run_golden("video_region_384", precision)
    errs, pm = _stage_checks(z, case, cfg, stages, outs, rtol, precision + ":")
    rl = torch.cat([o["pred_region_logits"].reshape(-1) for o in outs]).cpu().numpy()
    rl_err = float(np.abs(rl - z["pred_region_logits"]).max() / np.abs(z["pred_region_logits"]).max())
    _report(test="video_region_384", precision=precision, stage_err=errs, region_logit_err=rl_err)
    assert rl_err < (rtol if precision in EXACT else 0.15)
    if precision in EXACT:
        assert float(np.abs(results[0]["instances"].scores.cpu().numpy() - z["inst_scores"]).max()) < 2e-3


# ------------------------------------------------------------------------------------------- BASELINE.json configurations, full model
def _mask_iou(g, w):
    gm, wm = g > 0, w > 0
    inter = (gm & wm).flatten(1).sum(1).float()
    union = (gm | wm).flatten(1).sum(1).float()
    return torch.where(union > 0, inter / union.clamp(min=1), torch.ones_like(union)), float((gm == wm).float().mean())


@functools.lru_cache(maxsize=1)
def _full_model(task):
    """(cfg, state dict) of the full-size model.  Kept for the NEXT test of the same task (the config-2 / -3 tests sit together): generating
    1.5 G seeded parameters takes the host ~15 s, and the driver runs this file inside a fixed budget."""
    cfg = PsalmConfig(seg_task=task)
    return cfg, make_state_dict(cfg, seed=0)


@functools.lru_cache(maxsize=2)
def _full_psalm(task, precision):
    """The full-size model on the GPU, shared by consecutive tests of one task (eager launches; the object keeps no state between calls
    that a test could observe: tests/test_6_model_emu.py::test_replica_shares_weights_owns_state_and_agrees)."""
    from psalm_amd.model import PSALM
    cfg, sd = _full_model(task)
    return PSALM(cfg, sd, precision=precision)


def test_config2_panoptic_1024_f16x3_meets_north_star_bar():
    """BASELINE.json configs[1] (the bench workload): COCO-panoptic 1024x1024 batch 1, full 24-layer model, headline mode, vs the CPU
    oracle on the same seeded weights / inputs.  Bar = north star: mask IoU within 1e-3, argmax-identical labels (>= 99.9 % of pixels)."""
    from psalm_amd.model import PSALM
    cfg, sd = _full_model("panoptic")
    inputs = make_inputs(cfg, "panoptic", size=1024, batch=1, seed=0)
    want = O.eval_seg(sd, cfg, **inputs)[0]
    got = _full_psalm("panoptic", "f16x3").eval_seg(**inputs)[0]
    torch.cuda.synchronize()
    iou, pix = _mask_iou(got["mask_pred"].cpu(), want["mask_pred"])
    sem = float((got["sem_seg"].argmax(0).cpu() == want["sem_seg"].argmax(0)).float().mean())
    pan = float((got["panoptic_seg"][0].cpu() == want["panoptic_seg"][0]).float().mean())
    rel = float((got["mask_pred"].cpu() - want["mask_pred"]).abs().max() / want["mask_pred"].abs().max())
    _report(test="config2_panoptic_1024", precision="f16x3", mask_iou_mean=float(iou.mean()), mask_iou_min=float(iou.min()), mask_pixel_agree=pix,
            sem_argmax_agree=sem, panoptic_agree=pan, mask_logit_rel_err=rel, segments=[len(got["panoptic_seg"][1]), len(want["panoptic_seg"][1])])
    assert float(iou.mean()) >= 0.999 and sem >= 0.999 and pan >= 0.999
    assert got["panoptic_seg"][1] == want["panoptic_seg"][1]
    assert got["instances"].scores.numel() == want["instances"].scores.numel()
    assert float((torch.sort(got["instances"].scores.cpu()).values - torch.sort(want["instances"].scores).values).abs().max()) < 2e-3


def test_config2_padding_tiles_left_out_is_bitwise_the_plain_slice_kernel():
    """The library's default K loop for the Phi GEMMs (policy 2582: the phased slice kernel with the matrix instructions of all-padding m-tiles
    left out -- M = 899 on 256-row tiles) against the same kernel with them (2581, the form every parity record of r04 before r04p was taken
    with): the whole 1024 x 1024 panoptic evaluation, every output tensor, bit for bit."""
    from psalm_amd import hip_ops as H
    cfg, sd = _full_model("panoptic")
    inputs = make_inputs(cfg, "panoptic", size=1024, batch=1, seed=0)
    model = _full_psalm("panoptic", "f16x3")
    ops = model.ops
    outs = {}
    try:
        for pol in (2581, 2582):
            ops.gemm_tile_policy(pol)
            r = model.eval_seg(**inputs)[0]
            torch.cuda.synchronize()
            outs[pol] = (r["mask_pred"].clone(), r["sem_seg"].clone(), r["panoptic_seg"][0].clone(), r["instances"].scores.clone(),
                         r["instances"].pred_masks.clone(), list(r["panoptic_seg"][1]))
    finally:
        ops.gemm_tile_policy(H.Ops.GEMM_X3_256_DEFAULT)
    for a, b in zip(outs[2581][:5], outs[2582][:5]):
        assert torch.equal(a, b)
    assert outs[2581][5] == outs[2582][5]
    _report(test="config2_padding_tiles_left_out_bitwise", identical=True)


def test_config2_stage_level_calls_are_bitwise_the_op_by_op_sequence():
    """The stage-level C ABI (csrc/stages.hip: psalm_swin_forward, psalm_projector_forward, psalm_phi_forward, psalm_pixel_decoder_forward,
    psalm_predictor_forward -- one native call issues a stage's launches: ~560 of the image's ~600) against the op-by-op Python sequence it
    replaces (PSALM.c_stages = False): the whole 1024 x 1024 panoptic evaluation, every output tensor, bit for bit -- eagerly and through
    hipGraph capture / replay."""
    from psalm_amd.model import PSALM
    cfg, sd = _full_model("panoptic")
    inputs = make_inputs(cfg, "panoptic", size=1024, batch=1, seed=0)
    model = _full_psalm("panoptic", "f16x3")
    assert model.c_stages
    outs = {}
    try:
        for flag in (True, False):
            model.c_stages = flag
            r = model.eval_seg(**inputs)[0]
            torch.cuda.synchronize()
            outs[flag] = (r["mask_pred"].clone(), r["sem_seg"].clone(), r["panoptic_seg"][0].clone(), r["instances"].scores.clone(), list(r["panoptic_seg"][1]))
    finally:
        model.c_stages = True
    assert all((k,) in model._cache for k in ("swin_desc", "proj_desc", "phi_desc", "pd_desc", "pr_desc"))
    for a, b in zip(outs[True][:4], outs[False][:4]):
        assert torch.equal(a, b)
    assert outs[True][4] == outs[False][4]
    graphed = PSALM(cfg, sd, precision="f16x3", use_graphs=True)
    for _ in range(3):
        g = graphed.eval_seg(**inputs)[0]
    torch.cuda.synchronize()
    assert graphed.graph_stats["captures"] == 1 and torch.equal(g["mask_pred"], outs[False][0]) and torch.equal(g["panoptic_seg"][0], outs[False][2])
    _report(test="config2_stage_level_calls_bitwise", identical=True)
    del graphed
    torch.cuda.empty_cache()


def test_config2_panoptic_1024_multi_seed_default_and_fp32_control():
    """one image is a noisy gate (0.3 % positive pixels, ~10 empty reference masks, masks of a few pixels whose IoU
    moves in steps of 1/area).  Four more seeded inputs (seed 0 is the test above), same weights, the headline mode (three f16 products
    everywhere) AND the exact-fp32 GPU mode (information: the oracle's arithmetic width in another summation order) -- against one oracle
    run per input.  Asserted per input for the product, gate version 4 (oracle/parity_gate.py): every pixel whose sign differs from the
    oracle's has an oracle |logit| within 1e-5 of the logit range (a flip only where the oracle itself is within rounding of the threshold),
    pooled mask IoU >= 0.9995, mean IoU over the reference masks of >= 64 pixels >= 0.999, semantic / panoptic agreement >= 0.999.  (The plain
    mean over all 100 queries is reported and loosely bounded: one such pixel in a 4-pixel mask is IoU 0.75 for that query.)"""
    from oracle import parity_gate as PG
    cfg, sd = _full_model("panoptic")
    models = {"f16x3": _full_psalm("panoptic", "f16x3"), "fp32": _full_psalm("panoptic", "fp32")}
    for seed in (1, 2, 3, 4):
        inputs = make_inputs(cfg, "panoptic", size=1024, batch=1, seed=seed)
        want = O.eval_seg(sd, cfg, **inputs)[0]
        for mode, m in models.items():
            got = m.eval_seg(**inputs)[0]
            torch.cuda.synchronize()
            p = PG.parity_of(got, want)
            _report(test="config2_panoptic_1024_multi_seed", mode=mode, **dict(p, inputs_seed=seed))
            if mode == "f16x3":                               # the product's arithmetic: asserted per input
                assert p["flips_within_margin"], (seed, p["flip_margin_rel_max"], p["flipped_mask_pixels"])
                assert p["meets_bar_pooled"] and p["mask_iou_pooled"] >= 0.9995 and p["panoptic_id_agreement"] >= 0.999, (seed, p)
                assert p["mask_iou_mean"] >= 0.99 and p["mask_logit_rel_err"] < 1e-5, (seed, p)
    # The exact-fp32 mode is REPORTED only: r04a (profiles/r04_parity_wide.jsonl) found inputs seed 4 -- one of these -- 9e-3 of the logit
    # range / 266 pixels away from the oracle in that mode (the reference's own arithmetic in another summation order), while the
    # three-product arithmetic sits at 1.6e-6 on it.


def test_config2_seed11_knife_edge_input():
    """The one panoptic input of the 16-seed set on the committed knife-edge list (tests/golden/knife_edge_inputs.json): 1024x1024, inputs
    seed 11.  The fp32 oracle (= the reference itself on this input: 0 flipped pixels between the two, profiles/r04a_reference_vs_oracle_*.log)
    sits on a knife edge of the thresholded attention-mask feedback (TD:754-760) there: with EVERY linear layer of the oracle evaluated in
    float64 -- more exact than the reference -- it moves by 9.2e-4 of the logit range to 558 other pixels (tests/golden/make_seed11_control.py
    -> panoptic_1024_seed11_float64_control.npz).  The r04a product landed 2 pixels from that float64 result, the r04 HEAD product 4 pixels from
    the fp32 oracle.  Gate version 4: the test PASSES only on the oracle's side -- the flip-margin property and the
    plain-mean bar against the fp32 oracle; landing within 16 pixels of the float64 control is reported as XFAIL with the measured numbers
    (the more exact evaluation, but not the reference's result); anywhere else FAILS."""
    from oracle import parity_gate as PG
    cfg, sd = _full_model("panoptic")
    inputs = make_inputs(cfg, "panoptic", size=1024, batch=1, seed=11)
    want = O.eval_seg(sd, cfg, **inputs)[0]
    got = _full_psalm("panoptic", "f16x3").eval_seg(**inputs)[0]
    torch.cuda.synchronize()
    entry = PG.knife_edge_entry("panoptic", 1024, 11, 0)
    assert entry is not None
    p = PG.parity_of(got, want)
    ok, side = PG.judge(p, got, want, entry)
    _report(test="config2_seed11_knife_edge", **p)
    if side == "float64_control":
        pytest.xfail(f"knife-edge input landed on its float64 control's side: {p['flipped_mask_pixels']} pixels / {p['mask_logit_rel_err']:.1e} from the fp32 "
                     f"oracle, {p['knife_edge_symmetric_difference_vs_control']} pixels from the control")
    assert ok and side == "oracle", p
    assert p["mask_logit_rel_err"] < 1e-5 and p["flipped_mask_pixels"] <= 16


def test_config2_panoptic_1024_padded_box_and_original_size():
    """BASELINE configs[1] with what the reference's loop really feeds (coco_panoptic_mapper.py:81-89, LP:1418-1429): a
    480 x 640 original -> resized box 768 x 1024 inside the 1024^2 canvas (padding_mask set below it), results cropped to the box and resized
    to 480 x 640; and a portrait 640 x 427 one.  Through hipGraph replay (ONE graph for both and for the square input), vs the CPU oracle at
    the north star's bar + the flip-margin property; `sem_seg`, `panoptic_seg`, instance masks come out at the ORIGINAL size."""
    from oracle import parity_gate as PG
    from psalm_amd.model import PSALM
    from psalm_amd.synthetic import resized_box
    cfg, sd = _full_model("panoptic")
    model = PSALM(cfg, sd, precision="f16x3", use_graphs=True)
    model.eval_seg(**make_inputs(cfg, "panoptic", size=1024, batch=1, seed=0))            # eager sighting
    model.eval_seg(**make_inputs(cfg, "panoptic", size=1024, batch=1, seed=0))            # capture, on the square input
    for seed, (h, w) in ((21, (480, 640)), (22, (640, 427))):
        oh, ow = resized_box(h, w, 1024)
        inputs = make_inputs(cfg, "panoptic", size=1024, batch=1, seed=seed, geometry=[(oh, ow, h, w)])
        want = O.eval_seg(sd, cfg, **inputs)[0]
        got = model.eval_seg(**inputs)[0]
        torch.cuda.synchronize()
        assert tuple(got["mask_pred"].shape[-2:]) == (h, w) == tuple(want["mask_pred"].shape[-2:])
        assert tuple(got["sem_seg"].shape[-2:]) == (h, w) and tuple(got["panoptic_seg"][0].shape) == (h, w)
        assert got["instances"].image_size == (h, w) and tuple(got["instances"].pred_masks.shape[-2:]) == (h, w)
        p = PG.parity_of(got, want)
        _report(test="config2_panoptic_1024_padded_box", inputs_seed=seed, original=[h, w], box=[oh, ow], **p)
        assert p["flips_within_margin"] and p["meets_bar_pooled"] and p["mask_logit_rel_err"] < 1e-5, p
        assert p["panoptic_id_agreement"] >= 0.999 and got["panoptic_seg"][1] == want["panoptic_seg"][1]
    assert len(model._graphs) == 1 and model.graph_stats["captures"] == 1 and model.graph_stats["replays"] >= 3
    del model
    torch.cuda.empty_cache()


def test_config3_referring_640_batch4_ragged():
    """BASELINE.json configs[2]: RefCOCO-shaped 640x640 batch 4, four referring sentences of different lengths (6 / 9 / 13 / 21 tokens)
    and different prompt lengths -> the ragged-batch padding path (llava_phi.py:874-948); every image checked (the reference returns
    after image 0, LP:1472)."""
    from psalm_amd.model import PSALM
    cfg, sd = _full_model("referring")
    inputs = make_inputs(cfg, "referring", size=640, batch=4, seed=0)
    assert len({int(t.numel()) for t in inputs["token_refer_id"]}) == 4 and len(set(inputs["attention_mask"].sum(1).tolist())) == 4
    want = O.eval_seg(sd, cfg, **inputs)
    got = _full_psalm("referring", "f16x3").eval_seg(**inputs)
    torch.cuda.synchronize()
    assert len(got) == 4
    for b in range(4):
        iou, pix = _mask_iou(got[b]["mask_pred"].cpu(), want[b]["mask_pred"])
        gi, wi = got[b]["instances"], want[b]["instances"]
        sc = float((torch.sort(gi.scores.cpu()).values - torch.sort(wi.scores).values).abs().max())
        assert int(gi.query_index[gi.scores.argmax()]) == int(wi.query_index[wi.scores.argmax()])      # same top-1 query (what the evaluator keeps)
        bm = float((gi.pred_masks.cpu()[gi.scores.argmax()] != wi.pred_masks[wi.scores.argmax()]).float().mean())   # the evaluator's top-1 mask
        _report(test="config3_referring_640_b4", image=b, mask_iou_mean=float(iou.mean()), mask_iou_min=float(iou.min()), mask_pixel_agree=pix,
                score_err=sc, top1_mask_pixel_diff=bm)
        assert float(iou.mean()) >= 0.999 and pix >= 0.9999 and sc < 2e-3 and bm < 1e-3


def test_config3_referring_640_input_that_moved_the_r03_fast_form():
    """Regression input: referring 640x640 batch 4, inputs seed 4, image 0 -- the input on which r03's opt-in e4m3-cross-term form of the Phi
    GEMMs left the bar (mask logits off by 4e-3 of their range, mean IoU 0.9986; DESIGN.md section 0, profiles/r03n_*, r03o_*, r03s_*; the
    form was removed in r04).  The default arithmetic (three f16 products everywhere) agrees with the oracle on it like on every other
    input (1.6e-6)."""
    from psalm_amd.model import PSALM
    cfg, sd = _full_model("referring")
    inputs = make_inputs(cfg, "referring", size=640, batch=4, seed=4)
    want = O.eval_seg(sd, cfg, **inputs)
    got = _full_psalm("referring", "f16x3").eval_seg(**inputs)
    torch.cuda.synchronize()
    for b in range(4):
        iou, pix = _mask_iou(got[b]["mask_pred"].cpu(), want[b]["mask_pred"])
        rel = float((got[b]["mask_pred"].cpu() - want[b]["mask_pred"]).abs().max() / want[b]["mask_pred"].abs().max())
        _report(test="config3_referring_640_seed4", image=b, mask_iou_mean=float(iou.mean()), mask_pixel_agree=pix, mask_logit_rel_err=rel)
        assert float(iou.mean()) >= 0.999 and rel < 1e-4, (b, float(iou.mean()), rel)


@pytest.mark.parametrize("seed,image,sides", [(11, 1, ("oracle", "one_thread_control")), (10, 2, ("oracle", "float64_groupnorm_control"))])
def test_config3_referring_640_images_on_the_knife_edge_list(seed, image, sides):
    """The two referring inputs r04 / r05's wide runs left outside the flip margin (640x640 batch 4; seed 11 image 1: 214 pixels / 7.6e-3 in BOTH GPU
    arithmetics; seed 10 image 2: 58 pixels / 2.2e-3).  r06 closed them with controls of the CPU oracle itself (tools/exp_referring_controls.py,
    profiles/r06_referring_controls_seeds_10_11.jsonl): the fp32 oracle tips AGAINST ITSELF to the product's pixels when it merely runs on ONE host
    thread or with float64 attention (seed 11), resp. with its GroupNorms in float64 (seed 10) -- evaluations at least as exact as the reference's --
    and returns to its fp32 side with ALL arithmetic in float64; for seed 10 the oracle's own predictor, fed the product's stage outputs, returns the
    product's result (profiles/r06_referring_seed10_stage_bisect.jsonl).  The reference's fp32 result is within rounding of a decision of the
    thresholded attention-mask feedback (TD:754-760) on these images: they are on the committed knife-edge list with their control's flipped set
    as fixture (tests/golden/make_referring_seed1{0,1}_control.py).  The product must land on the oracle-as-run's side (flip-margin property) or
    within 16 pixels of the control's side -- which of the two the oracle itself takes depends on the host (thread count, BLAS) -- and nowhere else;
    the other three images of the batch must meet the plain property."""
    from oracle import parity_gate as PG
    cfg, sd = _full_model("referring")
    inputs = make_inputs(cfg, "referring", size=640, batch=4, seed=seed)
    want = O.eval_seg(sd, cfg, **inputs)
    got = _full_psalm("referring", "f16x3").eval_seg(**inputs)
    torch.cuda.synchronize()
    for b in range(4):
        entry = PG.knife_edge_entry("referring", 640, seed, 0, batch=4, image=b)
        assert (entry is not None) == (b == image)
        p = PG.parity_of(got[b], want[b])
        ok, side = PG.judge(p, got[b], want[b], entry)
        _report(test="config3_referring_640_knife_edge", seed=seed, image=b, oracle_threads=torch.get_num_threads(), **p)
        assert ok and side in (sides if b == image else ("oracle",)), (b, p)


def test_config5_region_1024_batch2():
    """BASELINE.json configs[4]: interactive (point-prompt discs) 1024x1024 batch 2 with 1 and 3 <region> prompts.  f16x3 at the north-star
    bar vs the oracle.  (The configuration's "fp8 MFMA LLM path": no fp8 form meets the parity bar on this network -- whole-operand e4m3 and
    e4m3 cross terms were both built, measured and removed, DESIGN.md section 0 -- so the configuration runs in the default arithmetic.)"""
    from psalm_amd.model import PSALM
    cfg, sd = _full_model("region")
    inputs = make_inputs(cfg, "region", size=1024, batch=2, seed=0)
    torch.manual_seed(RNG_SEED_AT_CALL)
    want = O.eval_seg(sd, cfg, **inputs)
    torch.manual_seed(RNG_SEED_AT_CALL)
    got = _full_psalm("region", "f16x3").eval_seg(**inputs)
    torch.cuda.synchronize()
    for b in range(2):
        gmask, wmask = got[b]["mask_pred"].cpu(), want[b]["mask_pred"]
        iou, pix = _mask_iou(gmask, wmask)
        # Random weights + point prompts give many few-pixel masks at 1024^2 (here: a 6-pixel mask with ONE pixel whose logit sits within
        # fp32 summation-order noise of 0 -> IoU 5/6, and its mask score moves with it).  Such a flip is below what ANY fp32 implementation
        # with a different accumulation order can reproduce, so the per-query statistics are taken over reference masks of >= 64 pixels;
        # smaller ones are bounded in absolute flipped pixels, and the pooled IoU / pixel agreement cover everything.
        area = (wmask > 0).flatten(1).sum(1)
        big = area >= 64
        flips = ((gmask > 0) != (wmask > 0)).flatten(1).sum(1)
        inter = ((gmask > 0) & (wmask > 0)).sum().float()
        union = ((gmask > 0) | (wmask > 0)).sum().float()
        pooled = float(inter / union.clamp(min=1))
        ds = (got[b]["instances"].scores.cpu() - want[b]["instances"].scores).abs()       # (Q, k)
        sc_big = float(ds[big].max()) if bool(big.any()) else 0.0
        _report(test="config5_region_1024_b2", precision="f16x3", image=b, mask_iou_mean=float(iou.mean()), mask_iou_min=float(iou.min()),
                mask_iou_mean_area_ge_64=float(iou[big].mean()) if bool(big.any()) else None, pooled_iou=pooled, mask_pixel_agree=pix,
                small_masks=int((~big).sum()), max_flips_small=int(flips[~big].max()) if bool((~big).any()) else 0,
                score_err_area_ge_64=sc_big, score_err_all=float(ds.max()))
        assert pooled >= 0.999 and pix >= 0.99999
        assert (not bool(big.any())) or (float(iou[big].mean()) >= 0.999 and sc_big < 2e-3)
        assert (not bool((~big).any())) or int(flips[~big].max()) <= 2
        assert tuple(got[b]["instances"].scores.shape) == tuple(want[b]["instances"].scores.shape)


@pytest.mark.parametrize("task", ["semantic", "instance", "panoptic"])
def test_open_vocab_class_count_459(task):
    """Open-vocabulary evaluations (psalm/eval/semantic_segmentation.py:418-507: ADE-150, PC-459, A-847 class lists) change only the
    number of class prompts: 459 classes + background = 460 class groups, L ~ 1.7 k tokens, 45 900 top-k candidates (beyond the fused
    semantic kernel's 160-class limit and the old 16 384-candidate top-k limit).  Full-width architecture with a 2-layer LLM at 384^2,
    headline mode vs the CPU oracle at the fp32 tolerances."""
    from psalm_amd.model import PSALM
    cfg = PsalmConfig(num_layers=2, seg_task=task)
    sd = make_state_dict(cfg, seed=3)
    inputs = make_inputs(cfg, task, size=384, batch=1, seed=3, num_classes=459)
    want = O.eval_seg(sd, cfg, **inputs)[0]
    got = PSALM(cfg, sd, precision="f16x3").eval_seg(**inputs)[0]
    torch.cuda.synchronize()
    rel = float((got["mask_pred"].cpu() - want["mask_pred"]).abs().max() / want["mask_pred"].abs().max())
    rep = {"test": "open_vocab_459", "task": task, "mask_logit_rel_err": rel}
    assert rel < 1e-4
    if task in ("semantic", "panoptic"):
        assert tuple(got["sem_seg"].shape) == (459, 384, 384)
        top2 = want["sem_seg"].topk(2, 0).values
        decided = (top2[0] - top2[1]) > 1e-5 * want["sem_seg"].abs().max()
        same = got["sem_seg"].argmax(0).cpu() == want["sem_seg"].argmax(0)
        rep["sem_argmax_agree"] = float(same.float().mean())
        assert bool(same[decided].all()) and float(same.float().mean()) >= 0.999
    if task in ("instance", "panoptic"):
        gi, wi = got["instances"], want["instances"]
        assert gi.scores.numel() == wi.scores.numel()
        og = np.lexsort((gi.pred_classes.cpu().numpy(), -gi.scores.cpu().numpy()))
        ow = np.lexsort((wi.pred_classes.numpy(), -wi.scores.numpy()))
        np.testing.assert_allclose(gi.scores.cpu().numpy()[og], wi.scores.numpy()[ow], atol=2e-3)
        rep["instances"] = int(gi.scores.numel())
    if task == "panoptic":
        assert float((got["panoptic_seg"][0].cpu() == want["panoptic_seg"][0]).float().mean()) >= 0.999
    _report(**rep)


def test_config5_region_1024_batch2_reduced_precision_llm():
    """BASELINE.json configs[4] names an "fp8 MFMA LLM path".  No fp8 form met the parity bar on this network (whole-operand e4m3, r02; e4m3 cross
    terms, r03: both removed); what the product offers for that configuration is a REDUCED-PRECISION LLM SIDE MODE, PSALM(llm_products=1): the
    two fused GEMMs of every Phi layer form ONE f16 product (operands = the hi halves of the split-f16 operands: 11-bit mantissas under the
    same per-row scales) instead of three -- a third of the LLM's matrix work; Swin, attention, pixel / mask decoder as in "f16x3".
    Stated tolerance of the side mode -- LOOSER than the north star's, which it does not meet (config 2, r05c: mask logits 1.0e-2 of their range,
    pooled IoU 0.9981, semantic argmax 99.2 %; on a contractive weight set 1.3e-3 / 0.9995 / 99.95 %): against the CPU oracle on this
    configuration: mask logits within 5e-2 of their range, pooled mask IoU >= 0.97, mask pixel agreement >= 0.9999; against the default
    arithmetic on the same GPU: LLM hidden states within 2e-2 relative L2, vision-tower outputs bit for bit equal."""
    from psalm_amd.model import PSALM
    cfg, sd = _full_model("region")
    inputs = make_inputs(cfg, "region", size=1024, batch=2, seed=0)
    torch.manual_seed(RNG_SEED_AT_CALL)
    want = O.eval_seg(sd, cfg, **inputs)
    m1 = PSALM(cfg, sd, precision="f16x3", llm_products=1)
    torch.manual_seed(RNG_SEED_AT_CALL)
    got = m1.eval_seg(**inputs)
    torch.cuda.synchronize()
    kw = {k: v for k, v in inputs.items() if k != "is_thing_list"}
    s1, s3 = {}, {}
    torch.manual_seed(RNG_SEED_AT_CALL)
    m1.forward_logits(stages=s1, **kw)
    torch.manual_seed(RNG_SEED_AT_CALL)
    _full_psalm("region", "f16x3").forward_logits(stages=s3, **kw)
    torch.cuda.synchronize()
    assert torch.equal(s1["image_tokens"], s3["image_tokens"])
    h1, h3 = s1["hidden_states"].double(), s3["hidden_states"].double()
    rel_h = float((h1 - h3).norm() / h3.norm())
    for b in range(2):
        gmask, wmask = got[b]["mask_pred"].cpu(), want[b]["mask_pred"]
        gm, wm = gmask > 0, wmask > 0
        pooled = float((gm & wm).sum().float() / (gm | wm).sum().float().clamp(min=1))
        pix = float((gm == wm).float().mean())
        rel = float((gmask - wmask).abs().max() / wmask.abs().max())
        _report(test="config5_region_1024_b2_llm_products_1", image=b, pooled_iou=pooled, mask_pixel_agree=pix, mask_logit_rel_err=rel,
                flipped_pixels=int((gm != wm).sum()), llm_hidden_rel_l2_vs_default=rel_h)
        assert rel <= 5e-2 and pooled >= 0.97 and pix >= 0.9999, (b, rel, pooled, pix)
    assert 1e-6 < rel_h <= 2e-2, rel_h
    del m1
    torch.cuda.empty_cache()
