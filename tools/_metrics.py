"""Result-comparison helpers shared by the tools/exp_*.py experiments (not product code)."""
import torch


def metrics(g, w_):
    gm, wm = g["mask_pred"] > 0, w_["mask_pred"] > 0
    inter = (gm & wm).flatten(1).sum(1).float()
    union = (gm | wm).flatten(1).sum(1).float()
    iou = torch.where(union > 0, inter / union.clamp(min=1), torch.ones_like(union))
    rel = ((g["mask_pred"] - w_["mask_pred"]).abs().max() / w_["mask_pred"].abs().max()).item()
    return {"mask_rel_err": rel, "iou_mean": float(iou.mean()), "iou_min": float(iou.min()), "iou_pooled": float(inter.sum() / union.sum()),
            "pos_frac": float(wm.float().mean()), "n_empty_ref": int((wm.flatten(1).sum(1) == 0).sum()),
            "pix_agree": float((gm == wm).float().mean()),
            "sem_agree": float((g["sem_seg"].argmax(0) == w_["sem_seg"].argmax(0)).float().mean()),
            "pan_agree": float((g["panoptic_seg"][0] == w_["panoptic_seg"][0]).float().mean()),
            "segs": [len(g["panoptic_seg"][1]), len(w_["panoptic_seg"][1])]}


def clone(r):
    return {"mask_pred": r["mask_pred"].clone(), "sem_seg": r["sem_seg"].clone(),
            "panoptic_seg": (r["panoptic_seg"][0].clone(), r["panoptic_seg"][1])}
