"""AUTHORING CONTAINER ONLY (needs /root/reference): the reference's OWN `PSALM.eval_seg` -- unmodified source, imported through
tests/golden/ref_shim.py, the seeded synthetic checkpoint loaded with load_state_dict(strict=True) -- timed on this container's host cores on
the bench workload (BASELINE.json configs[1]: COCO-panoptic 1024x1024, batch 1, full 24-layer model), as BASELINE.md section 3 specifies:
fp32, torch.no_grad, 1 warm-up + 3 timed runs, median.  The GPU box has no /root/reference, so bench.py's `cpu_baseline` stays the oracle
("port") timed there and cites this file (profiles/r04_reference_cpu.json) next to it.

    python tools/time_reference_cpu.py [--size 1024] [--runs 3] [--seeds 0] [--out profiles/r04_reference_cpu.json]

Also records, per input seed, how the ORACLE (oracle/psalm_oracle.py, the restatement the GPU path is checked against) compares with the
reference itself on the full-size input: the same statistics tools/parity_wide.py reports for the GPU modes."""
import argparse
import json
import os
import platform
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def cpu_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--seeds", default="0")
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_reference_cpu.json"))
    args = ap.parse_args()
    import make_golden as MG
    from exp_x8_cpu import compare
    from oracle import psalm_oracle as O
    from psalm_amd.config import PsalmConfig
    from psalm_amd.synthetic import make_inputs, make_state_dict
    torch.set_num_threads(args.threads)
    cfg = PsalmConfig(seg_task="panoptic")
    sd = make_state_dict(cfg, seed=0, include_lm_head=True)
    model = MG.build_reference(cfg, sd)
    seeds = [int(s) for s in args.seeds.split(",")]
    cap = {}
    model.predictor.register_forward_hook(lambda m, i, o: cap.update(pred_masks=o["pred_masks"]))     # the reference returns no mask logits
    res = {"what": "reference PSALM.eval_seg (unmodified source via tests/golden/ref_shim.py), fp32, torch CPU",
           "workload": f"COCO-panoptic {args.size}x{args.size} batch 1, 24-layer Phi, 134 class prompts, 100 queries (bench.py's workload, weights seed 0)",
           "cpu": cpu_name(), "threads": args.threads, "torch": torch.__version__, "per_seed": []}
    for seed in seeds:
        inputs = make_inputs(cfg, "panoptic", size=args.size, batch=1, seed=seed)
        times = []
        out = None
        for i in range(1 + (args.runs if seed == seeds[0] else 1)):          # warm-up + timed runs on the first seed; one timed run on the others
            torch.manual_seed(MG.RNG_SEED_AT_CALL)
            t0 = time.perf_counter()
            with torch.no_grad():
                out = model.eval_seg(**inputs)
            dt = time.perf_counter() - t0
            if i > 0:
                times.append(dt)
            print(f"seed {seed} run {i}: {dt:.2f} s", file=sys.stderr, flush=True)
        torch.manual_seed(MG.RNG_SEED_AT_CALL)
        t0 = time.perf_counter()
        want = O.eval_seg({k: v for k, v in sd.items()}, cfg, **inputs)
        t_or = time.perf_counter() - t0
        r = out[0]
        # the reference's full-resolution mask logits: its own up-sampling (LP:1401-1406) of the predictor output captured by the hook
        ref_mask = torch.nn.functional.interpolate(cap["pred_masks"], size=(args.size, args.size), mode="bilinear", align_corners=False)[0]
        cmp_ = compare({"mask_pred": want[0]["mask_pred"]}, {"mask_pred": ref_mask})
        row = {"inputs_seed": seed, "reference_seconds": [round(t, 2) for t in times], "oracle_seconds": round(t_or, 2)}
        if "sem_seg" in r:
            row["oracle_vs_reference_semantic_argmax_agreement"] = round(float((want[0]["sem_seg"].argmax(0) == r["sem_seg"].argmax(0)).float().mean()), 6)
        if "panoptic_seg" in r:
            row["oracle_vs_reference_panoptic_id_agreement"] = round(float((want[0]["panoptic_seg"][0] == r["panoptic_seg"][0]).float().mean()), 6)
            row["panoptic_segments"] = [len(want[0]["panoptic_seg"][1]), len(r["panoptic_seg"][1])]
        if cmp_:
            row["oracle_vs_reference_masks"] = cmp_
        res["per_seed"].append(row)
        print(json.dumps(row), flush=True)
    t = sorted(res["per_seed"][0]["reference_seconds"])
    med = t[len(t) // 2]
    res["seconds_per_image_median"] = round(med, 2)
    res["images_per_s"] = round(1.0 / med, 4)
    res["note"] = ("the reference runs the Swin tower twice per image (LP:787, LP:1369) and MSDeformAttn through its pure-PyTorch grid_sample fallback "
                   "(no CPU kernel in the extension); both are part of what a user of the reference's CPU path gets and are included")
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "per_seed"}))


if __name__ == "__main__":
    main()
