"""TEST INFRASTRUCTURE (like everything under oracle/): the parity gate between the product's results and the CPU oracle's -- one
definition shared by bench.py's parity leg and tests/test_9_e2e_gpu.py.  Never imported by psalm_amd/.

Gate version 4 (VERDICT r04 "Next" #4).  North star: "mask IoU within 1e-3 of the CPU reference, argmax-identical pixel
labels on fixed seeds".  The network has one discontinuity on the way to its mask logits -- the thresholded attention mask of
mask2former_transformer_decoder.py:754-760 (`sigmoid(resized mask) < 0.5` decides which keys a query may see in the next layer) -- so an
input either reproduces the reference to fp32 round-off (typically 1.6e-6 of the mask-logit range, 0 - 5 pixels whose oracle logit is
within that distance of 0) or, when one of those thresholded pixels sits within round-off of 0 in the REFERENCE's own evaluation, tips as
a whole (hundreds of pixels, 1e-3 .. 1e-2 of the range; DESIGN.md section 0 of r04: the exact-fp32 GPU mode tips 3 of 78 inputs, the
reference's own arithmetic in float64 tips panoptic seed 11).  Rounds 3 / 4 gated on a COUNT ("no more inputs below the bar than the
exact-fp32 control"); this version gates on a PROPERTY, per input:

    flips_within_margin   every mask pixel whose sign differs from the oracle's has an oracle |logit| <= FLIP_MARGIN_REL * max|oracle logit|
                          -- a sign may differ only where the oracle itself is within rounding of the threshold.  FLIP_MARGIN_REL = 1e-5:
                          ~6x the typical logit error of the fp32-class arithmetic (84 fp32 eps of the logit range).  An input that tipped
                          fails it by two orders of magnitude (panoptic seed 11: 9.2e-4).
    meets_bar_plain_mean  the north star's literal statistic: mean over the 100 queries of mask IoU >= 0.999 AND semantic argmax agreement
                          >= 99.9 %.  The headline boolean.
    meets_bar_pooled      pooled mask IoU >= 0.999 AND mean IoU over reference masks of >= 64 px >= 0.999 AND semantic argmax >= 99.9 %
                          (reported; one flipped pixel of a 4-pixel mask does not decide it).
    meets_north_star_bar  = meets_bar_plain_mean AND flips_within_margin, on every seeded input.

Knife-edge list (tests/golden/knife_edge_inputs.json): inputs on which the REFERENCE's fp32 result is itself on the edge, each entry
carrying the fixture that shows it (the oracle with every linear layer in float64 -- more exact than the reference -- lands elsewhere).
Such an input passes when it meets the property against the oracle, or is within KNIFE_EDGE_PIXELS pixels of its control's flipped set
(recorded as `side`: "oracle" | "float64_control"); tests report the second outcome as xfail, never as a pass.
"""
from __future__ import annotations

import json
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KNIFE_EDGE_JSON = os.path.join(ROOT, "tests", "golden", "knife_edge_inputs.json")
FLIP_MARGIN_REL = 1e-5
KNIFE_EDGE_PIXELS = 16

GATE = {"version": 4,
        "meets_north_star_bar": "= meets_bar_plain_mean AND flips_within_margin on every seeded input (inputs on the committed knife-edge list: see "
                                "`knife_edge_inputs`)",
        "meets_bar_plain_mean": "on every seeded input: mean over the 100 queries of mask IoU vs the CPU oracle >= 0.999 AND semantic argmax agreement "
                                ">= 99.9 % (north_star's literal statistic) -- the headline boolean",
        "flips_within_margin": f"on every seeded input: every mask pixel whose sign differs from the oracle's has an oracle |logit| <= {FLIP_MARGIN_REL:g} x "
                               "max|oracle logit| (a sign may differ only where the oracle itself is within rounding of the threshold; ~6x the typical "
                               "1.6e-6 logit error of the fp32-class arithmetic; an input tipped by the thresholded attention-mask feedback of TD:754-760 "
                               "fails it by two orders of magnitude)",
        "meets_bar_pooled": "reported, not gating: pooled mask IoU >= 0.999 AND mean IoU over reference masks of >= 64 px >= 0.999 AND semantic argmax "
                            "agreement >= 99.9 %",
        "knife_edge_inputs": "tests/golden/knife_edge_inputs.json: inputs on which the reference's own fp32 result is on the edge (each entry carries its "
                             f"float64-control fixture); such an input passes on the oracle's side by the property above, or within {KNIFE_EDGE_PIXELS} pixels of "
                             "its control's flipped set (`side`); none of bench.py's default seeds 0-4 is on the list",
        "fp32_control": "other_modes.fp32 (the exact-fp32 GPU mode on the same inputs) is reported as information: it is NOT part of any pass / fail "
                        "decision since gate version 4"}


def knife_edge_inputs():
    if not os.path.exists(KNIFE_EDGE_JSON):
        return []
    with open(KNIFE_EDGE_JSON) as f:
        return json.load(f)["inputs"]


def knife_edge_entry(task, size, inputs_seed, weights_seed=0, batch=1, image=0):
    """The list entry of image `image` of the seeded input, or None.  (An entry without an "image" key is image 0 of its batch.)"""
    for e in knife_edge_inputs():
        if (e["task"], e["size"], e["inputs_seed"], e["weights_seed"], e.get("batch", 1), e.get("image", 0)) == (task, size, inputs_seed, weights_seed, batch, image):
            return e
    return None


def parity_of(g, w_):
    """One image: product result dict `g` (device or host tensors) vs oracle result dict `w_` (host)."""
    gmp = g["mask_pred"].cpu()
    wmp = w_["mask_pred"]
    gm, wm = gmp > 0, wmp > 0
    inter = (gm & wm).flatten(1).sum(1).float()
    union = (gm | wm).flatten(1).sum(1).float()
    iou = torch.where(union > 0, inter / union.clamp(min=1), torch.ones_like(union))
    big = wm.flatten(1).sum(1) >= 64                     # a smaller mask's IoU moves in steps of >= 1/64 per flipped pixel
    diff = gm != wm
    rng = wmp.abs().max()
    margin = float(wmp.abs()[diff].max() / rng) if bool(diff.any()) else 0.0
    p = {"mask_iou_mean": round(float(iou.mean()), 5), "mask_iou_min": round(float(iou.min()), 5),
         "mask_iou_mean_area_ge_64": round(float(iou[big].mean()), 6) if bool(big.any()) else None, "ref_masks_lt_64px": int((~big).sum()),
         "mask_iou_pooled": round(float(inter.sum() / union.sum().clamp(min=1)), 6), "flipped_mask_pixels": int(diff.sum()),
         "flip_margin_rel_max": float(f"{margin:.3e}"),
         "mask_logit_rel_err": float(f"{((gmp - wmp).abs().max() / rng).item():.3e}"),
         "mask_pixel_agreement": round(float((gm == wm).float().mean()), 6)}
    if "sem_seg" in g and "sem_seg" in w_:
        p["semantic_argmax_agreement"] = round(float((g["sem_seg"].argmax(0).cpu() == w_["sem_seg"].argmax(0)).float().mean()), 6)
    if "panoptic_seg" in g and "panoptic_seg" in w_:
        p["panoptic_id_agreement"] = round(float((g["panoptic_seg"][0].cpu() == w_["panoptic_seg"][0]).float().mean()), 6)
        p["panoptic_segments"] = [len(g["panoptic_seg"][1]), len(w_["panoptic_seg"][1])]
    p["flips_within_margin"] = bool(margin <= FLIP_MARGIN_REL)
    p["meets_bar_pooled"] = at_pooled(p)
    p["meets_bar_plain_mean"] = at_plain(p)
    return p


def at_pooled(p):
    big = p["mask_iou_mean_area_ge_64"]
    return bool(p["mask_iou_pooled"] >= 0.999 and (big is None or big >= 0.999) and p.get("semantic_argmax_agreement", 1.0) >= 0.999)


def at_plain(p):
    return bool(p["mask_iou_mean"] >= 0.999 and p.get("semantic_argmax_agreement", 1.0) >= 0.999)


def flipped_set(g, w_):
    gm, wm = g["mask_pred"].cpu() > 0, w_["mask_pred"] > 0
    return {tuple(int(v) for v in r) for r in torch.nonzero(gm != wm).tolist()}


def judge(p, g=None, w_=None, entry=None):
    """Pass / fail of one input under gate version 4.  `entry`: its knife-edge list entry, if any (then g / w_ are needed for the fall-back
    comparison with the control's flipped set).  Returns (passes, side) and records both in p."""
    ok = bool(p["meets_bar_plain_mean"] and p["flips_within_margin"])
    side = "oracle" if ok else None
    if not ok and entry is not None and g is not None:
        import numpy as np
        ctl = np.load(os.path.join(ROOT, entry["control"]))
        control = {tuple(int(v) for v in r) for r in ctl["flipped_qyx"].tolist()}
        sym = len(flipped_set(g, w_) ^ control)
        p["knife_edge_symmetric_difference_vs_control"] = sym
        if sym <= KNIFE_EDGE_PIXELS:
            ok, side = True, entry.get("control_side", "float64_control")
    if entry is not None:
        p["on_knife_edge_list"] = True
    p["passes_gate"], p["side"] = ok, side
    return ok, side
