"""Wide parity run (VERDICT r03 "Next" #1): the GPU path in the DEFAULT arithmetic (`f16x3`: three f16 products) and in the exact-fp32 mode
(`fp32`: the oracle's arithmetic in another summation order -- how far does an input move under re-ordering ALONE?) against the CPU oracle
over many seeded inputs: by default panoptic 1024x1024 inputs seeds 0-15 and referring 640x640 batch 4 inputs seeds 3-15 (the input sets of
profiles/r03s_cpu_emulation_*).

    python tools/parity_wide.py [--sets panoptic:1024:1:0-15,referring:640:4:3-15] [--modes f16x3,fp32] [--out gpurun_out/r04_parity_wide.jsonl]
                                [--workers W] [--threads T]

The oracle (fp32 torch on the host) is the slow part (~15 s per 1024 image): W worker processes with T threads each compute it for the
inputs in parallel (default: as many 16-thread workers as the host has cores for) while the GPU modes run; the comparison itself runs on the
GPU.  One JSON line per (input, image, mode) and a summary line per mode:
    gate "pooled":      per image pooled mask IoU >= 0.999 AND mean IoU over reference masks of >= 64 px >= 0.999 AND labels >= 99.9 %
    gate "plain_mean":  per image mean IoU over the 100 queries >= 0.999 AND labels >= 99.9 %       (north_star's literal statistic)
and per input the ratio of flipped pixels f16x3 : fp32 mode (the fp32 mode is the noise floor any re-implementation sits on)."""
import argparse
import json
import os
import sys
import time

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse_sets(s):
    out = []
    for part in s.split(","):
        task, size, batch, seeds = part.split(":")
        lo, _, hi = seeds.partition("-")
        out.append((task, int(size), int(batch), list(range(int(lo), int(hi or lo) + 1))))
    return out


EMU = os.environ.get("PARITY_EMU") == "1"      # dry run of this script without a GPU: tiny architecture, kernels in the host emulation (tests/emu)


def _cfg(task):
    from psalm_amd.config import PsalmConfig
    return PsalmConfig.tiny(task) if EMU else PsalmConfig(seg_task=task)


def _inputs(cfg, task, size, batch, seed):
    from psalm_amd.synthetic import make_inputs
    return make_inputs(cfg, task, size=size, batch=batch, seed=seed, **({"num_classes": 9} if EMU else {}))


def oracle_worker(wid, threads, jobs, results, wseed, done):
    """jobs: (task, size, batch, seed); results: (job, list of per-image dicts of CPU tensors, seconds)"""
    torch.set_num_threads(threads)
    from oracle import psalm_oracle as O
    from psalm_amd.config import PsalmConfig
    from psalm_amd.synthetic import make_inputs, make_state_dict
    sds = {}
    while True:
        job = jobs.get()
        if job is None:
            done.wait()              # the shared-memory handles of queued results are served by THIS process: stay until the reader is through
            return
        task, size, batch, seed = job
        cfg = _cfg(task)
        if task not in sds:
            sds[task] = make_state_dict(cfg, seed=wseed)
        inputs = _inputs(cfg, task, size, batch, seed)
        t0 = time.perf_counter()
        torch.manual_seed(1234)
        want = O.eval_seg(sds[task], cfg, **inputs)
        secs = time.perf_counter() - t0
        slim = []
        for w_ in want:
            d = {"mask_pred": w_["mask_pred"].contiguous().share_memory_()}
            if "sem_seg" in w_:
                d["sem_argmax"] = w_["sem_seg"].argmax(0).to(torch.int16).share_memory_()
            if "panoptic_seg" in w_:
                d["pan_ids"] = w_["panoptic_seg"][0].to(torch.int32).share_memory_()
                d["pan_segments"] = len(w_["panoptic_seg"][1])
            slim.append(d)
        results.put((job, slim, secs))


from oracle import parity_gate as PG  # noqa: E402


@torch.no_grad()
def compare_dev(g, w_):
    """GPU result dict vs slim oracle dict, computed on the device"""
    dev = g["mask_pred"].device
    wp = w_["mask_pred"].to(dev, non_blocking=False)
    gp = g["mask_pred"]
    gm, wm = gp > 0, wp > 0
    inter = (gm & wm).flatten(1).sum(1).double()
    union = (gm | wm).flatten(1).sum(1).double()
    iou = torch.where(union > 0, inter / union.clamp(min=1), torch.ones_like(union))
    area = wm.flatten(1).sum(1)
    big = area >= 64
    out = {"mask_iou_mean": round(float(iou.mean()), 6), "mask_iou_min": round(float(iou.min()), 5),
           "mask_iou_mean_area_ge_64": round(float(iou[big].mean()), 6) if bool(big.any()) else None,
           "mask_iou_pooled": round(float(inter.sum() / union.sum()), 6) if float(union.sum()) > 0 else 1.0,
           "flipped_pixels": int((gm != wm).sum()), "ref_positive_pixels": int(wm.sum()), "ref_masks_empty": int((area == 0).sum()),
           "ref_masks_lt_64px": int(((area > 0) & ~big).sum()),
           "mask_logit_rel_err": float(f"{((gp - wp).abs().max() / wp.abs().max()).item():.3e}")}
    # gate version 4 (oracle/parity_gate.py): the largest oracle |logit| at a pixel whose sign differs, relative to the logit range
    diff = gm != wm
    out["flip_margin_rel_max"] = float(f"{(wp.abs()[diff].max() / wp.abs().max()).item():.3e}") if bool(diff.any()) else 0.0
    if "sem_argmax" in w_ and "sem_argmax" in g:
        out["semantic_argmax_agreement"] = round(float((g["sem_argmax"] == w_["sem_argmax"].to(dev).long()).double().mean()), 6)
    if "pan_ids" in w_ and "panoptic_seg" in g:
        out["panoptic_id_agreement"] = round(float((g["panoptic_seg"][0] == w_["pan_ids"].to(dev).to(g["panoptic_seg"][0].dtype)).double().mean()), 6)
        out["panoptic_segments"] = [len(g["panoptic_seg"][1]), w_["pan_segments"]]
    return out


def gates(r):
    lab = r.get("semantic_argmax_agreement", 1.0) >= 0.999
    big = r["mask_iou_mean_area_ge_64"]
    margin = bool(r["flip_margin_rel_max"] <= 1e-5)
    return {"pooled": bool(r["mask_iou_pooled"] >= 0.999 and (big is None or big >= 0.999) and lab),
            "plain_mean": bool(r["mask_iou_mean"] >= 0.999 and lab), "flips_within_margin": margin,
            "v4": bool(margin and r["mask_iou_mean"] >= 0.999 and lab)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sets", default="panoptic:1024:1:0-15,referring:640:4:3-15")
    ap.add_argument("--modes", default="f16x3,fp32")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r05_parity_wide.jsonl"))
    ap.add_argument("--workers", type=int, default=0)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--weights-seed", type=int, default=0)
    ap.add_argument("--gemm-policy", default="", help="comma-separated psalm_gemm_set_tile_policy codes (kernel A/B runs)")
    args = ap.parse_args()
    ncpu = os.cpu_count() or 8
    threads = min(args.threads, ncpu)
    workers = args.workers or max(1, min(8, ncpu // threads))
    sets = parse_sets(args.sets)
    jobs_list = [(task, size, batch, s) for task, size, batch, seeds in sets for s in seeds]

    ctx = mp.get_context("spawn")
    jobs, results, done = ctx.Queue(), ctx.Queue(), ctx.Event()
    for j in jobs_list:
        jobs.put(j)
    for _ in range(workers):
        jobs.put(None)
    procs = [ctx.Process(target=oracle_worker, args=(i, threads, jobs, results, args.weights_seed, done), daemon=True) for i in range(workers)]
    for p in procs:
        p.start()

    from psalm_amd.config import PsalmConfig
    from psalm_amd.model import PSALM
    from psalm_amd.synthetic import make_inputs, make_state_dict
    torch.set_num_threads(max(1, min(8, ncpu - workers * threads)) if ncpu > workers * threads else 4)
    modes = args.modes.split(",")
    t_start = time.perf_counter()
    gpu_out, cur_task, models = {}, None, {}
    for task, size, batch, seeds in sets:
        cfg = _cfg(task)
        if task != cur_task:
            models.clear()
            sd = make_state_dict(cfg, seed=args.weights_seed)
            if EMU:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                from ops_backend import make_ops
                models = {m: PSALM(cfg, sd, ops=make_ops("emu"), precision=m) for m in modes}
            else:
                torch.cuda.empty_cache()
                models = {m: PSALM(cfg, sd, precision=m, use_graphs=False) for m in modes}
            cur_task = task
            for model in models.values():
                for code in [int(c) for c in args.gemm_policy.split(",") if c]:
                    model.ops.gemm_tile_policy(code)
        for s in seeds:
            inputs = _inputs(cfg, task, size, batch, s)
            for m, model in models.items():
                torch.manual_seed(1234)
                got = model.eval_seg(**inputs)
                if not EMU:
                    torch.cuda.synchronize()
                for r_ in got:                                   # keep the (C, H, W) class map as its argmax only (0.5 GB per 1024 image)
                    if "sem_seg" in r_:
                        r_["sem_argmax"] = r_.pop("sem_seg").argmax(0)
                    r_.pop("instances", None)
                gpu_out[(task, size, batch, s, m)] = got
    models.clear()
    t_gpu = time.perf_counter() - t_start
    print(f"# GPU modes done in {t_gpu:.1f} s; waiting for the oracle workers ({workers} x {threads} threads)", file=sys.stderr, flush=True)

    rows = []
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        for _ in range(len(jobs_list)):
            job, want, secs = results.get(timeout=1800)
            task, size, batch, s = job
            for m in modes:
                got = gpu_out.pop((task, size, batch, s, m))
                for b in range(len(want)):
                    r = {"task": task, "size": size, "batch": batch, "weights_seed": args.weights_seed, "inputs_seed": s, "image": b, "mode": m,
                         **compare_dev(got[b], want[b]), "oracle_seconds": round(secs, 1), "oracle_threads": threads}
                    r["gates"] = gates(r)
                    # r06: images on the committed knife-edge list (oracle/parity_gate.py; per-image entries): outside the margin they pass within 16
                    # pixels of their control's flipped set, as in bench.py / tests/test_9_e2e_gpu.py
                    entry = PG.knife_edge_entry(task, size, s, args.weights_seed, batch=batch, image=b)
                    if entry is not None:
                        r["on_knife_edge_list"] = True
                        if not r["gates"]["v4"]:
                            import numpy as np
                            ctl = {tuple(int(v) for v in q) for q in np.load(os.path.join(ROOT, entry["control"]))["flipped_qyx"].tolist()}
                            mine = {tuple(int(v) for v in q) for q in torch.nonzero((got[b]["mask_pred"] > 0).cpu() != (want[b]["mask_pred"] > 0)).tolist()}
                            r["knife_edge_symmetric_difference_vs_control"] = len(mine ^ ctl)
                            r["gates"]["v4_with_knife_edge_list"] = len(mine ^ ctl) <= PG.KNIFE_EDGE_PIXELS
                            r["side"] = entry.get("control_side", "float64_control") if r["gates"]["v4_with_knife_edge_list"] else None
                    rows.append(r)
                    f.write(json.dumps(r) + "\n")
                    f.flush()
                del got
            del want
        summ = {"summary": {}, "wall_seconds": round(time.perf_counter() - t_start, 1), "gpu_seconds": round(t_gpu, 1),
                "oracle_workers": workers, "oracle_threads": threads}
        for m in modes:
            for task in sorted({r["task"] for r in rows}):
                rs = [r for r in rows if r["mode"] == m and r["task"] == task]
                if not rs:
                    continue
                summ["summary"][f"{m}/{task}"] = {
                    "images": len(rs), "at_gate_pooled": sum(r["gates"]["pooled"] for r in rs), "at_gate_plain_mean": sum(r["gates"]["plain_mean"] for r in rs),
                    "flips_within_margin": sum(r["gates"]["flips_within_margin"] for r in rs), "at_gate_v4": sum(r["gates"]["v4"] for r in rs),
                    "flip_margin_rel_max_over_inputs_within_margin": max((r["flip_margin_rel_max"] for r in rs if r["gates"]["flips_within_margin"]), default=0.0),
                    "inputs_outside_margin": [[r["inputs_seed"], r["image"], r["flip_margin_rel_max"], r["mask_logit_rel_err"], r["flipped_pixels"]] for r in rs if not r["gates"]["flips_within_margin"]],
                    "inputs_outside_margin_and_not_within_16_pixels_of_a_listed_control": [[r["inputs_seed"], r["image"]] for r in rs if not r["gates"]["flips_within_margin"] and not r["gates"].get("v4_with_knife_edge_list", False)],
                    "mask_iou_mean_min": min(r["mask_iou_mean"] for r in rs), "mask_iou_pooled_min": min(r["mask_iou_pooled"] for r in rs),
                    "mask_iou_mean_area_ge_64_min": min((r["mask_iou_mean_area_ge_64"] for r in rs if r["mask_iou_mean_area_ge_64"] is not None), default=None),
                    "mask_logit_rel_err_max": max(r["mask_logit_rel_err"] for r in rs), "flipped_pixels_max": max(r["flipped_pixels"] for r in rs),
                    "flipped_pixels_total": sum(r["flipped_pixels"] for r in rs),
                    "semantic_argmax_agreement_min": min((r.get("semantic_argmax_agreement", 1.0) for r in rs)),
                    "panoptic_id_agreement_min": min((r.get("panoptic_id_agreement", 1.0) for r in rs)),
                    "inputs_moved_gt_1e-4": [[r["inputs_seed"], r["image"], r["mask_logit_rel_err"], r["flipped_pixels"]] for r in rs if r["mask_logit_rel_err"] > 1e-4]}
        if "f16x3" in modes and "fp32" in modes:
            key = lambda r: (r["task"], r["inputs_seed"], r["image"])      # noqa: E731
            fl32 = {key(r): r["flipped_pixels"] for r in rows if r["mode"] == "fp32"}
            worst = sorted(((r["flipped_pixels"], fl32[key(r)], key(r)) for r in rows if r["mode"] == "f16x3"), key=lambda t: -(t[0] - 2 * t[1]))[:5]
            summ["f16x3_vs_fp32_floor"] = {"rule": "f16x3 flipped pixels <= max(2 x fp32-mode flipped pixels, 8) per image",
                                           "violations": [list(k) for a, b_, k in ((w[0], w[1], w[2]) for w in worst) if a > max(2 * b_, 8)],
                                           "worst": [{"input": list(k), "f16x3": a, "fp32": b_} for a, b_, k in worst]}
        f.write(json.dumps(summ) + "\n")
        print(json.dumps(summ), flush=True)
    done.set()
    for p in procs:
        p.join(timeout=10)


if __name__ == "__main__":
    main()
