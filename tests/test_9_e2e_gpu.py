"""GPU parity tests proper: the HIP path (through the C ABI) vs
  (1) the golden vectors generated from the REFERENCE code (tests/golden/*.npz), full-size architecture;
  (2) the CPU oracle on the same seeded inputs (tiny architecture, all three tasks).
Tolerances are stated per mode:
  fp32 mode : stage tensors within 1e-3 * absmax (fp32 round-off through ~100 layers), labels/masks >= 99.9 % identical
  bf16 mode : stage tensors within 6e-2 * absmax, mask IoU and label agreement reported and bounded below.
"""
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
                         video=case.get("video", False))
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
    pm = torch.stack([o["pred_masks"] for o in outs])
    # mask logits sit behind the 24-layer LLM and the 9 discontinuous mask -> attention-mask feedback steps: in bf16 on random
    # weights they move 2-3x more than the feed-forward stages (measured 1.4e-2 .. 7e-2 of absmax over the five golden cases)
    errs["pred_masks"] = check_signature(z, "pred_masks", pm, rtol if rtol < 1e-2 else 2 * rtol, what=tag)
    mf = torch.stack([m.view(pm.shape[-2], pm.shape[-1], -1).permute(2, 0, 1) for m in stages["mask_features"]])
    errs["mask_features"] = check_signature(z, "mask_features", mf.contiguous(), rtol, what=tag)
    return errs, pm


@pytest.mark.parametrize("precision,rtol", [("fp32", 1e-3), ("bf16", 6e-2)])
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
    if precision == "fp32":
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


@pytest.mark.parametrize("precision,rtol", [("fp32", 1e-3), ("bf16", 6e-2)])
def test_golden_referring_384_b2(precision, rtol):
    case, z, cfg, stages, outs, results = _run_golden("referring_384_b2", precision)
    errs, pm = _stage_checks(z, case, cfg, stages, outs, rtol, precision + ":")
    seg = torch.stack([o["pred_SEG_logits"] for o in outs]).cpu().numpy()
    seg_err = float(np.abs(seg - z["pred_SEG_logits"]).max() / np.abs(z["pred_SEG_logits"]).max())
    sc = np.sort(results[0]["instances"].scores.cpu().numpy())
    sc_err = float(np.abs(sc - np.sort(z["inst_scores"])).max())
    _report(test="referring_384_b2", precision=precision, stage_err=errs, seg_err=seg_err, score_err=sc_err)
    assert len(results) == 2
    assert seg_err < (rtol if precision == "fp32" else 0.15) and sc_err < (2e-3 if precision == "fp32" else 0.2)


@pytest.mark.parametrize("precision,rtol", [("fp32", 1e-3), ("bf16", 6e-2)])
def test_golden_region_384(precision, rtol):
    case, z, cfg, stages, outs, results = _run_golden("region_384", precision)
    errs, pm = _stage_checks(z, case, cfg, stages, outs, rtol, precision + ":")
    rl = torch.cat([o["pred_region_logits"].reshape(-1) for o in outs]).cpu().numpy()
    rl_err = float(np.abs(rl - z["pred_region_logits"]).max() / np.abs(z["pred_region_logits"]).max())
    sc_err = float(np.abs(results[0]["instances"].scores.cpu().numpy() - z["inst_scores"]).max())
    _report(test="region_384", precision=precision, stage_err=errs, region_logit_err=rl_err, score_err=sc_err)
    assert rl_err < (rtol if precision == "fp32" else 0.15)
    if precision == "fp32":
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
    for precision, tol in (("fp32", 2e-3), ("bf16", 8e-2)):
        model = PSALM(cfg, sd, precision=precision)
        torch.manual_seed(5)
        got = model.eval_seg(**inputs)
        for b in range(batch):
            a, w = got[b]["mask_pred"].cpu(), want[b]["mask_pred"]
            err = float((a - w).abs().max() / w.abs().max())
            _report(test=f"tiny_{task}", precision=precision, image=b, mask_pred_err=err)
            assert err < tol, (precision, err)


def test_tiny_fused_decoder_heads_on_gpu():
    """PSALM.fuse_heads (single-launch LayerNorm + mask_embed MLP and out-projection + residual + LayerNorm; off by default): same
    results as the separate launches up to bf16 rounding of the intermediate activations, and within the bf16 tolerance of the oracle."""
    from psalm_amd.model import PSALM
    cfg = PsalmConfig.tiny("panoptic")
    sd = make_state_dict(cfg, seed=12)
    inputs = make_inputs(cfg, "panoptic", size=128, batch=1, seed=4, num_classes=9)      # 4x4 / 8x8 / 16x16 levels: MFMA attention path
    torch.manual_seed(5)
    want = O.eval_seg(sd, cfg, **inputs)[0]["mask_pred"]
    outs = {}
    for fused in (False, True):
        model = PSALM(cfg, sd, precision="bf16")
        model.fuse_heads = fused
        torch.manual_seed(5)
        outs[fused] = model.eval_seg(**inputs)[0]["mask_pred"].cpu()
        err = float((outs[fused] - want).abs().max() / want.abs().max())
        _report(test="tiny_fused_heads", fused=fused, mask_pred_err=err)
        assert err < 8e-2, (fused, err)
    assert float((outs[True] - outs[False]).abs().max() / want.abs().max()) < 4e-2


@pytest.mark.parametrize("task,batch", [("panoptic", 1), ("referring", 2)])
def test_graph_replay_is_bitwise_eager(task, batch):
    """use_graphs=True: 1st call eager, 2nd call captures the launch sequence into a hipGraph, later calls replay it with
    new inputs copied into the graph's static buffers.  Every mode must give bit-identical results."""
    from psalm_amd.model import PSALM
    cfg = PsalmConfig.tiny(task)
    sd = make_state_dict(cfg, seed=12)
    eager = PSALM(cfg, sd, precision="bf16")
    graphed = PSALM(cfg, sd, precision="bf16", use_graphs=True)
    for call, seed in enumerate((4, 5, 6, 4)):             # same shapes, different pixels / token ids each call
        inputs = make_inputs(cfg, task, size=96, batch=batch, seed=seed, num_classes=9)
        want = eager.eval_seg(**inputs)
        got = graphed.eval_seg(**inputs)
        torch.cuda.synchronize()
        for b in range(batch):
            assert torch.equal(got[b]["mask_pred"], want[b]["mask_pred"]), (call, b)
            assert torch.equal(got[b]["instances"].scores, want[b]["instances"].scores)
            assert torch.equal(got[b]["instances"].pred_masks, want[b]["instances"].pred_masks)
            if task == "panoptic":
                assert torch.equal(got[b]["sem_seg"], want[b]["sem_seg"])
                assert torch.equal(got[b]["panoptic_seg"][0], want[b]["panoptic_seg"][0])
                assert got[b]["panoptic_seg"][1] == want[b]["panoptic_seg"][1]
    assert any("graph" in e for e in graphed._graphs.values())


@pytest.mark.parametrize("precision,rtol", [("fp32", 1e-3), ("bf16", 6e-2)])
def test_golden_semantic_384(precision, rtol):
    case, z, cfg, stages, outs, results = _run_golden("semantic_384", precision)
    errs, pm = _stage_checks(z, case, cfg, stages, outs, rtol, precision + ":")
    sem = results[0]["sem_seg"]
    agree = float((sem.argmax(0).to(torch.uint8).cpu().numpy() == z["sem_seg_argmax"]).mean())
    _report(test="semantic_384", precision=precision, stage_err=errs, sem_argmax_agree=agree)
    assert tuple(sem.shape[-2:]) == (case["size"] - case["pad"],) * 2
    if precision == "fp32":
        check_signature(z, "sem_seg", sem, 1e-3)
        assert agree > 0.999
    else:
        assert agree > 0.75


@pytest.mark.parametrize("precision,rtol", [("fp32", 1e-3), ("bf16", 6e-2)])
def test_golden_instance_384(precision, rtol):
    case, z, cfg, stages, outs, results = _run_golden("instance_384", precision)
    errs, pm = _stage_checks(z, case, cfg, stages, outs, rtol, precision + ":")
    inst = results[0]["instances"]
    assert len(inst) == cfg.md_queries
    sc = np.sort(inst.scores.cpu().numpy())
    sc_err = float(np.abs(sc - np.sort(z["inst_scores"])).max())
    _report(test="instance_384", precision=precision, stage_err=errs, score_err=sc_err)
    if precision == "fp32":
        og = np.lexsort((inst.pred_classes.cpu().numpy(), -inst.scores.cpu().numpy()))
        ow = np.lexsort((z["inst_classes"], -z["inst_scores"]))
        np.testing.assert_allclose(inst.scores.cpu().numpy()[og], z["inst_scores"][ow], atol=2e-3)
        assert (inst.pred_classes.cpu().numpy()[og] == z["inst_classes"][ow]).all()
    else:
        # random-weight bf16 (see test_golden_panoptic_512): individual near-tied candidates swap, the score distribution holds
        assert float(np.abs(sc - np.sort(z["inst_scores"])).mean()) < 0.05 and sc_err < 0.6


def test_fp8_llm_path_on_gpu():
    """precision="fp8" (BASELINE.json configs[4]: interactive / region prompts with the fp8 MFMA LLM path): full-size architecture
    with a 2-layer LLM, 384x384, vs the oracle evaluated with the same e4m3 fake-quantisation (tolerance = bf16 mode's), plus the
    distance to the reference-generated golden vectors (reported; e4m3 has 3 mantissa bits)."""
    from psalm_amd.model import PSALM
    case, z = load_case("region_384")
    cfg = PsalmConfig(num_layers=case["layers"], seg_task=case["task"])
    sd = make_state_dict(cfg, seed=case["seed"])
    inputs = make_inputs(cfg, task=case["task"], size=case["size"], batch=case["batch"], seed=case["seed"], pad=case["pad"])
    kw = {k: v for k, v in inputs.items() if k != "is_thing_list"}
    torch.manual_seed(RNG_SEED_AT_CALL)
    _, st8 = O.eval_seg(sd, cfg, return_stages=True, postprocess=False, llm_fp8=True, **inputs)
    model = PSALM(cfg, sd, precision="fp8")
    stages = {}
    torch.manual_seed(RNG_SEED_AT_CALL)
    outs = model.forward_logits(stages=stages, **kw)
    torch.manual_seed(RNG_SEED_AT_CALL)
    res = model.eval_seg(**inputs)
    torch.cuda.synchronize()
    Lb = st8["lengths"][0]
    hs = stages["hidden_states"][0, :Lb].float().cpu()
    e_h = float((hs - st8["hidden_states"][0, :Lb]).abs().max() / st8["hidden_states"][0, :Lb].abs().max())
    pm = outs[0]["pred_masks"].float().cpu()
    e_m = float((pm - st8["pred_masks"][0]).abs().max() / st8["pred_masks"][0].abs().max())
    g = torch.from_numpy(z["pred_masks_s4"])[0]
    e_gold = float((pm[:, ::4, ::4] - g).abs().max() / g.abs().max())
    _report(test="fp8_region_384", hidden_err_vs_fp8_oracle=e_h, mask_err_vs_fp8_oracle=e_m, mask_err_vs_reference_golden=e_gold)
    assert e_h < 3e-2 and e_m < 0.12
    assert res[0]["instances"].pred_masks.shape[0] == cfg.md_queries


@pytest.mark.parametrize("precision,rtol", [("fp32", 1e-3), ("bf16", 6e-2)])
def test_golden_video_region_384(precision, rtol):
    """eval_video (PSALMForDAVISEval, LP:1845-1998) against the golden generated by the reference class."""
    case, z, cfg, stages, outs, results = _run_golden("video_region_384", precision)
    errs, pm = _stage_checks(z, case, cfg, stages, outs, rtol, precision + ":")
    rl = torch.cat([o["pred_region_logits"].reshape(-1) for o in outs]).cpu().numpy()
    rl_err = float(np.abs(rl - z["pred_region_logits"]).max() / np.abs(z["pred_region_logits"]).max())
    _report(test="video_region_384", precision=precision, stage_err=errs, region_logit_err=rl_err)
    assert rl_err < (rtol if precision == "fp32" else 0.15)
    if precision == "fp32":
        assert float(np.abs(results[0]["instances"].scores.cpu().numpy() - z["inst_scores"]).max()) < 2e-3
