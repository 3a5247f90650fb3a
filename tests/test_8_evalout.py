"""SURVEY §8 f2 -- evaluator-facing outputs on the device vs the host arithmetic of the reference's evaluators (oracle/evalout_ref.py).
Integer / byte work: every comparison is EXACT."""
import numpy as np
import pytest
import torch

from ops_backend import ops  # noqa: F401
from oracle import evalout_ref as R
from psalm_amd import evalout as E


def test_rle_codec_known_answers_and_round_trip():
    m = np.array([[0, 1], [1, 1]], np.uint8)                    # column-major: 0 1 1 1
    assert R.rle_encode(m) == [1, 3]
    assert R.rle_encode(np.ones((2, 3), np.uint8)) == [0, 6]    # starts with foreground -> leading zero-length run
    assert R.rle_encode(np.zeros((2, 3), np.uint8)) == [6]
    m2 = np.array([[1, 0, 0], [0, 0, 1]], np.uint8)             # column-major: 1 0 | 0 0 | 0 1
    assert R.rle_encode(m2) == [0, 1, 4, 1]
    rng = np.random.default_rng(0)
    for h, w, p in [(7, 5, 0.5), (40, 33, 0.1), (64, 64, 0.9), (3, 200, 0.3)]:
        m = (rng.random((h, w)) < p).astype(np.uint8)
        c = R.rle_encode(m)
        assert sum(c) == h * w
        s = R.rle_to_string(c)
        assert all(48 <= ch < 112 for ch in s)
        assert R.rle_from_string(s) == c and np.array_equal(R.rle_decode(c, h, w), m)
        assert E.rle_counts_to_string(c) == s
    big = [0, 5, 100000, 3, 7, 2 ** 20, 1]                      # multi-char counts, negative differences
    assert R.rle_from_string(R.rle_to_string(big)) == big


@pytest.mark.parametrize("C,H,W", [(9, 33, 47), (133, 64, 64), (1, 5, 5)])
def test_semantic_labels_and_confusion_matrix(ops, C, H, W):
    g = torch.Generator().manual_seed(C + H)
    sem = torch.randn(C, H, W, generator=g)
    sem[:, 0, 0] = 0.25                                          # an exact tie: first class wins (torch.argmax / np.argmax)
    gt = torch.randint(0, C, (H, W), generator=g)
    gt[torch.rand(H, W, generator=g) < 0.1] = 255
    pred_ref, conf_ref = R.semantic_confusion(sem.numpy(), gt.numpy(), C, 255)
    lab = E.semantic_labels(sem.to(ops.device), ops=ops)
    assert lab.dtype == torch.int32 and np.array_equal(lab.cpu().numpy(), pred_ref)
    cm = E.ConfusionMatrix(C, 255, ops=ops)
    cm.update(lab, gt)
    cm.update(lab, gt)                                           # accumulates over images
    assert np.array_equal(cm.conf.cpu().numpy(), 2 * conf_ref)
    iou, miou = cm.miou()
    assert iou.shape == (C,) and 0.0 <= miou <= 1.0


def test_panoptic_png_rgb(ops):
    ids = torch.tensor([[0, 1, 255, 256], [65535, 65536, 16777215, 70000]], dtype=torch.int32)
    got = E.panoptic_png_rgb(ids.to(ops.device), ops=ops).cpu().numpy()
    assert np.array_equal(got, R.id2rgb(ids.numpy()))


@pytest.mark.parametrize("n,H,W,dtype", [(3, 40, 33, torch.float32), (5, 64, 300, torch.uint8), (2, 7, 5, torch.bool), (1, 1, 1, torch.float32)])
def test_masks_to_rle(ops, n, H, W, dtype):
    g = torch.Generator().manual_seed(n * H + W)
    m = (torch.rand(n, H, W, generator=g) < torch.tensor([0.5, 0.05, 0.95, 0.0, 1.0])[:n, None, None])
    m[0, :, 0] = True                                            # first mask starts with foreground
    masks = m.to(dtype).to(ops.device)
    got = E.masks_to_rle(masks, ops=ops)
    assert len(got) == n
    for i in range(n):
        want = R.rle_encode(m[i].numpy())
        assert got[i]["size"] == [H, W] and got[i]["counts"] == R.rle_to_string(want)
        assert np.array_equal(R.rle_decode(R.rle_from_string(got[i]["counts"]), H, W), m[i].numpy().astype(np.uint8))


def test_iou_counts_and_meters(ops):
    g = torch.Generator().manual_seed(3)
    n, m, H, W = 6, 3, 50, 70
    pred = (torch.rand(n, H, W, generator=g) < 0.4).float()
    gt = (torch.rand(m, H, W, generator=g) < 0.3).to(torch.uint8)
    gt[0][torch.rand(H, W, generator=g) < 0.2] = 255             # ignore region
    gt[2] = 0                                                    # no-object target
    pred[5] = 0
    pairs = [(0, 0), (3, 1), (5, 2), (1, 2)]
    inter, union, tgt = E.iou_counts(pred.to(ops.device), gt, pairs, ops=ops)
    meters = E.IoUMeters()
    ref = {"I": np.zeros(2), "U": np.zeros(2), "acc": np.zeros(2), "n": 0}
    for k, (p, t) in enumerate(pairs):
        ai, au, at = R.intersection_and_union(pred[p].numpy().astype(np.uint8), gt[t].numpy())
        assert np.array_equal(inter[k].cpu().numpy(), ai) and np.array_equal(union[k].cpu().numpy(), au) and np.array_equal(tgt[k].cpu().numpy(), at)
        R.compute_metric_update(ref, pred[p].numpy().astype(np.uint8), gt[t].numpy())
    meters.update(inter, union)
    res = meters.results()
    assert res["n"] == 4
    assert abs(res["ciou"] - ref["I"][1] / (ref["U"][1] + 1e-10)) < 1e-12
    assert abs(res["giou"] - ref["acc"][1] / 4) < 1e-9
    meters.all_reduce()                                          # no process group: identity
    assert meters.results()["n"] == 4


def test_compact_results_equal_host_evaluator_formulas_on_model_outputs():
    """End of the path: a tiny-model eval_seg result (kernels in the emulator) -> psalm_amd.evalout.compact_results (device) == the
    reference evaluators' host arithmetic (oracle/evalout_ref.py) applied to the SAME tensors, exactly; and the oracle's own outputs
    through the host formulas agree with it up to the model's fp32 round-off (labels >= 99 %, identical panoptic segments)."""
    from ops_backend import make_ops
    from oracle import psalm_oracle as O
    from psalm_amd.config import PsalmConfig
    from psalm_amd.model import PSALM
    from psalm_amd.synthetic import make_inputs, make_state_dict
    ops = make_ops("emu")
    cfg = PsalmConfig.tiny("panoptic")
    sd = make_state_dict(cfg, seed=12)
    inputs = make_inputs(cfg, "panoptic", size=96, batch=1, seed=4, num_classes=9)
    torch.manual_seed(5)
    r = PSALM(cfg, sd, ops=ops, precision="f16x3").eval_seg(**inputs)[0]
    torch.manual_seed(5)
    w = O.eval_seg(sd, cfg, **inputs)[0]
    gt = torch.randint(0, 9, (96, 96), generator=torch.Generator().manual_seed(1))
    gt[:10] = 255
    cm = E.ConfusionMatrix(9, 255, ops=ops)
    c = E.compact_results(r, gt_sem=gt, conf=cm, ops=ops)
    pred_ref, conf_ref = R.semantic_confusion(r["sem_seg"].cpu().numpy(), gt.numpy(), 9, 255)
    assert np.array_equal(c["sem_labels"].cpu().numpy(), pred_ref) and np.array_equal(cm.conf.cpu().numpy(), conf_ref)
    assert np.array_equal(c["panoptic_rgb"].cpu().numpy(), R.id2rgb(r["panoptic_seg"][0].cpu().numpy()))
    assert c["segments_info"] == r["panoptic_seg"][1]
    masks = r["instances"].pred_masks.cpu().numpy()
    assert len(c["instances"]["rle"]) == masks.shape[0]
    for i, rle in enumerate(c["instances"]["rle"]):
        assert rle["counts"] == R.rle_to_string(R.rle_encode(masks[i] != 0))
    # the oracle's outputs through the host formulas: same decisions up to fp32 round-off of the model
    pred_o, _ = R.semantic_confusion(w["sem_seg"].numpy(), gt.numpy(), 9, 255)
    assert (pred_o == pred_ref).mean() >= 0.99
    assert np.array_equal(R.id2rgb(w["panoptic_seg"][0].numpy()), c["panoptic_rgb"].cpu().numpy())
    assert w["panoptic_seg"][1] == c["segments_info"]


# ---- the REFERENCE's evaluator arithmetic run on seeded outputs (tests/golden/make_evalout_golden.py -> evalout.npz) --------------------
def _golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "evalout.npz"))


def test_oracle_matches_reference_evaluator_golden():
    """oracle/evalout_ref.py vs what the reference's own intersectionAndUnionGPU / compute_metric / my_SemSegEvaluator.process produced."""
    G = _golden()
    ref = {"I": np.zeros(2), "U": np.zeros(2), "acc": np.zeros(2), "n": 0}
    for i in range(4):
        pred, gt, sc = G[f"s{i}/pred"], G[f"s{i}/gt"], G[f"s{i}/scores"]
        top = int(np.argmax(sc))
        assert np.array_equal(G[f"s{i}/kept_pred"], pred[top])
        ai, au, at = R.intersection_and_union(pred[top], gt)
        assert np.array_equal(ai, G[f"s{i}/intersection"].astype(np.int64)) and np.array_equal(au, G[f"s{i}/union"].astype(np.int64))
        assert np.array_equal(at, G[f"s{i}/target"].astype(np.int64))
        R.compute_metric_update(ref, pred[top], gt)
    assert np.array_equal(ref["I"], G["meters/intersection_sum"]) and np.array_equal(ref["U"], G["meters/union_sum"])
    assert np.allclose(ref["acc"], G["meters/acc_iou_sum"], rtol=0, atol=1e-12) and ref["n"] == int(G["meters/count"])
    assert abs(ref["I"][1] / (ref["U"][1] + 1e-10) - float(G["meters/ciou"])) < 1e-15
    C = int(G["sem/num_classes"])
    _, conf = R.semantic_confusion(G["sem/logits"], G["sem/gt"], C, 255)
    assert np.array_equal(conf, G["sem/conf_matrix"])


def test_device_evalout_matches_reference_evaluator_golden(ops):
    """psalm_amd.evalout (kernels: emulator here, the GPU under -m gpu) vs the same golden: counts exact, cIoU / gIoU to float64 round-off."""
    G = _golden()
    meters = E.IoUMeters()
    for i in range(4):
        pred, gt, sc = torch.from_numpy(G[f"s{i}/pred"]), torch.from_numpy(G[f"s{i}/gt"]), G[f"s{i}/scores"]
        top = int(np.argmax(sc))
        inter, union, tgt = E.iou_counts(pred.to(ops.device), gt[None], [(top, 0)], ops=ops)
        assert np.array_equal(inter[0].cpu().numpy(), G[f"s{i}/intersection"].astype(np.int64))
        assert np.array_equal(union[0].cpu().numpy(), G[f"s{i}/union"].astype(np.int64))
        assert np.array_equal(tgt[0].cpu().numpy(), G[f"s{i}/target"].astype(np.int64))
        meters.update(inter, union)
    res = meters.results()
    assert res["n"] == int(G["meters/count"])
    assert abs(res["ciou"] - float(G["meters/ciou"])) < 1e-12 and abs(res["giou"] - float(G["meters/giou"])) < 1e-12
    C = int(G["sem/num_classes"])
    cm = E.ConfusionMatrix(C, 255, ops=ops)
    cm.update(E.semantic_labels(torch.from_numpy(G["sem/logits"]).to(ops.device), ops=ops), torch.from_numpy(G["sem/gt"]))
    assert np.array_equal(cm.conf.cpu().numpy(), G["sem/conf_matrix"])


# ---- gRefCOCO: the reference's compute_metric / fuse_masks (psalm/eval/eval_grefcoco.py) run on seeded candidates -> grefcoco.npz -----------
def _gref():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "grefcoco.npz"))


def test_oracle_matches_reference_grefcoco_fusion_golden():
    """oracle/evalout_ref.py::grefcoco_fused_prediction vs what the reference's compute_metric kept and counted: several candidates above the
    threshold, one, none (top-1 fall-back), a no-object target, a tie."""
    G = _gref()
    ref = {"I": np.zeros(2), "U": np.zeros(2), "acc": np.zeros(2), "n": 0}
    for i in range(5):
        fused = R.grefcoco_fused_prediction(G[f"s{i}/pred"], G[f"s{i}/scores"], 0.6)
        assert np.array_equal(fused, G[f"s{i}/fused"]), i
        ai, au, _ = R.intersection_and_union(fused, G[f"s{i}/gt"])
        assert np.array_equal(ai, G[f"s{i}/intersection"].astype(np.int64)) and np.array_equal(au, G[f"s{i}/union"].astype(np.int64))
        R.compute_metric_update(ref, fused, G[f"s{i}/gt"])
    assert np.array_equal(ref["I"], G["meters/intersection_sum"]) and np.array_equal(ref["U"], G["meters/union_sum"])
    assert np.allclose(ref["acc"], G["meters/acc_iou_sum"], rtol=0, atol=1e-12) and ref["n"] == int(G["meters/count"])


@pytest.mark.parametrize("dtype", [torch.uint8, torch.float32])
def test_device_grefcoco_fusion_matches_reference_golden(ops, dtype):
    """psalm_amd.evalout.fuse_masks_by_score + iou_counts + IoUMeters (kernels: emulator here, the GPU under -m gpu) vs the same golden."""
    G = _gref()
    meters = E.IoUMeters()
    for i in range(5):
        pred = torch.from_numpy(G[f"s{i}/pred"]).to(dtype)
        fused = E.fuse_masks_by_score(pred.to(ops.device), torch.from_numpy(G[f"s{i}/scores"]).to(ops.device), 0.6, ops=ops)
        assert fused.dtype == torch.uint8 and np.array_equal(fused.cpu().numpy(), G[f"s{i}/fused"]), i
        inter, union, _ = E.iou_counts(fused[None], torch.from_numpy(G[f"s{i}/gt"])[None], [(0, 0)], ops=ops)
        assert np.array_equal(inter[0].cpu().numpy(), G[f"s{i}/intersection"].astype(np.int64))
        assert np.array_equal(union[0].cpu().numpy(), G[f"s{i}/union"].astype(np.int64))
        meters.update(inter, union)
    assert np.array_equal(meters.sum[0:2].numpy(), G["meters/intersection_sum"]) and np.array_equal(meters.sum[2:4].numpy(), G["meters/union_sum"])
    assert np.allclose(meters.sum[4:6].numpy(), G["meters/acc_iou_sum"], rtol=0, atol=1e-12) and int(meters.sum[6]) == int(G["meters/count"])
    with pytest.raises(Exception):
        E.fuse_masks_by_score(torch.zeros(2, 4, 4), torch.zeros(3), ops=ops)


def test_device_fusion_truncates_float_masks_like_astype_uint8(ops):
    """compute_metric casts the candidate masks with `preds.astype(np.uint8)` before fusing (eval_grefcoco.py:116): a float element counts as set
    after truncation toward zero -- 0.5 and -0.9 do not, 1.7 and -1.0 (-> 255) do; the fall-back without a confident candidate takes the FIRST
    maximal score."""
    vals = torch.tensor([0.0, 0.5, 0.999, 1.0, 1.7, -0.9, -1.0, 2.0])
    masks = torch.zeros(3, 2, 4)
    masks[0] = vals.view(2, 4)
    scores = torch.tensor([0.9, 0.1, 0.1])
    fused = E.fuse_masks_by_score(masks.to(ops.device), scores.to(ops.device), 0.6, ops=ops).cpu().numpy().reshape(-1)
    want = (vals.numpy().astype(np.int64).astype(np.uint8) != 0).astype(np.uint8)
    assert np.array_equal(fused, want) and want.tolist() == [0, 0, 0, 1, 1, 0, 1, 1]
    masks[1, 0, 0], masks[2, 0, 1] = 1.0, 1.0
    tie = E.fuse_masks_by_score(masks.to(ops.device), torch.tensor([0.2, 0.5, 0.5]).to(ops.device), 0.6, ops=ops).cpu().numpy().reshape(-1)
    assert tie[0] == 1 and tie[1] == 0                                   # candidate 1 (the first of the two maximal scores), not candidate 2
