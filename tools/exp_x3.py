"""Experiment (not product): does a split-bf16 ("bf16x3": a_hi.w_hi + a_lo.w_hi + a_hi.w_lo over a 3x longer K panel, fp32 accumulate)
GEMM reach the north-star parity bar that plain bf16 misses?  The exact-fp32 GPU mode (== CPU oracle to 2e-6) is the reference here;
the split is done with torch ops on the GPU (experiment only) and fed to the shipped bf16 MFMA GEMM."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from psalm_amd.config import PsalmConfig
from psalm_amd.model import PSALM
from psalm_amd.synthetic import make_inputs, make_state_dict
from psalm_amd import hip_ops as H

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = PsalmConfig(seg_task="panoptic")
sd = make_state_dict(cfg, seed=0)
inputs = make_inputs(cfg, "panoptic", size=size, batch=1, seed=0)
inputs["images"] = inputs["images"].cuda()

def metrics(g, w_):
    gm, wm = g["mask_pred"] > 0, w_["mask_pred"] > 0
    inter = (gm & wm).flatten(1).sum(1).float(); union = (gm | wm).flatten(1).sum(1).float()
    iou = torch.where(union > 0, inter / union.clamp(min=1), torch.ones_like(union))
    rel = ((g["mask_pred"] - w_["mask_pred"]).abs().max() / w_["mask_pred"].abs().max()).item()
    return {"mask_rel_err": rel, "iou_mean": float(iou.mean()), "iou_min": float(iou.min()), "iou_pooled": float(inter.sum() / union.sum()),
            "pos_frac": float(wm.float().mean()), "n_empty_ref": int((wm.flatten(1).sum(1) == 0).sum()),
            "pix_agree": float((gm == wm).float().mean()),
            "sem_agree": float((g["sem_seg"].argmax(0) == w_["sem_seg"].argmax(0)).float().mean()),
            "pan_agree": float((g["panoptic_seg"][0] == w_["panoptic_seg"][0]).float().mean()),
            "segs": [len(g["panoptic_seg"][1]), len(w_["panoptic_seg"][1])]}

def clone(r):
    return {"mask_pred": r["mask_pred"].clone(), "sem_seg": r["sem_seg"].clone(), "panoptic_seg": (r["panoptic_seg"][0].clone(), r["panoptic_seg"][1])}

def main():
    out = {}
    m32 = PSALM(cfg, sd, precision="fp32")
    m32.eval_seg(**inputs); torch.cuda.synchronize()
    t0 = time.perf_counter(); ref = clone(m32.eval_seg(**inputs)[0]); torch.cuda.synchronize(); out["fp32_ms"] = (time.perf_counter() - t0) * 1e3
    recs = []; m32.ops.lib.records = recs; m32.eval_seg(**inputs); torch.cuda.synchronize(); m32.ops.lib.records = None
    agg = {}
    for name, a, e0, e1 in recs:
        d = agg.setdefault(name, [0, 0.0]); d[0] += 1; d[1] += e0.elapsed_time(e1)
    out["fp32_breakdown_ms"] = {k: [v[0], round(v[1], 3)] for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}

    # ---- x3 GEMM via monkeypatch
    ops = m32.ops
    orig = ops.gemm
    wcache = {}
    def split(t):
        hi = t.to(torch.bfloat16); lo = (t - hi.float()).to(torch.bfloat16); return hi, lo
    def gemm_x3(a, w, bias=None, residual=None, act=H.ACT_NONE, act_col_start=0, out=None, out_dtype=None):
        if w.dtype != torch.float32 or a.dtype != torch.float32:
            return orig(a, w, bias, residual, act, act_col_start, out, out_dtype)
        key = (w.data_ptr(), tuple(w.shape), w.stride(0))
        static = key in wkeys
        wp = wcache.get(key) if static else None
        if wp is None:
            hi, lo = split(w if w.is_contiguous() else w.contiguous())
            wp = torch.cat((hi, hi, lo), 1).contiguous()
            if static: wcache[key] = wp
        hi, lo = split(a if a.is_contiguous() else a.contiguous())
        ap = torch.cat((hi, lo, hi), 1).contiguous()
        if out is None:
            out = ops.empty(a.shape[0], w.shape[0], dtype=out_dtype or torch.float32)
        return orig(ap, wp, bias, residual, act, act_col_start, out, None)
    wkeys = {(t.data_ptr(), tuple(t.shape), t.stride(0)) for t in m32.w.values() if t.dim() == 2}
    ops.gemm = gemm_x3
    m32.eval_seg(**inputs); torch.cuda.synchronize()
    t0 = time.perf_counter(); got = clone(m32.eval_seg(**inputs)[0]); torch.cuda.synchronize(); out["x3_emulated_ms"] = (time.perf_counter() - t0) * 1e3
    out["x3_vs_fp32"] = metrics(got, ref)
    ops.gemm = orig
    del m32; torch.cuda.empty_cache()
    mb = PSALM(cfg, sd, precision="bf16")
    got = clone(mb.eval_seg(**inputs)[0]); torch.cuda.synchronize()
    out["bf16_vs_fp32"] = metrics(got, ref)
    print(json.dumps(out, indent=1))



if __name__ == '__main__':
    main()
