"""Host-thread sweep of the CPU oracle (bench.py's `cpu_baseline`, kind "port") on the GPU box: VERDICT r03 weak #9 -- 64 threads on this model
were slower than the reference on 8.  One 1024x1024 panoptic image (bench.py's workload), a warm-up on a 256x256 image per thread count, then
`--runs` timed runs each; writes {"seconds_by_threads": {...}, "best_threads": T} for bench.py (profiles/r04_cpu_baseline_threads.json).

    python tools/cpu_baseline_sweep.py [--threads 8,16,32,64] [--runs 2] [--out gpurun_out/r04_cpu_baseline_threads.json]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="8,16,32,64")
    ap.add_argument("--runs", type=int, default=2)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r04_cpu_baseline_threads.json"))
    args = ap.parse_args()
    from oracle import psalm_oracle as O
    from psalm_amd.config import PsalmConfig
    from psalm_amd.synthetic import make_inputs, make_state_dict
    cfg = PsalmConfig(seg_task="panoptic")
    sd = make_state_dict(cfg, seed=0)
    big = make_inputs(cfg, "panoptic", size=args.size, batch=1, seed=0)
    small = make_inputs(cfg, "panoptic", size=256, batch=1, seed=0)
    ncpu = os.cpu_count() or 1
    res = {}
    for t in [int(x) for x in args.threads.split(",")]:
        if t > ncpu:
            continue
        torch.set_num_threads(t)
        O.eval_seg(sd, cfg, **small)
        ts = []
        for _ in range(args.runs):
            t0 = time.perf_counter()
            O.eval_seg(sd, cfg, **big)
            ts.append(round(time.perf_counter() - t0, 2))
        res[str(t)] = ts
        print(t, ts, flush=True)
    best = min(res, key=lambda k: min(res[k]))
    out = {"what": f"oracle/psalm_oracle.py eval_seg, panoptic {args.size}x{args.size} batch 1, fp32, seconds per image by torch thread count",
           "host_cpus": ncpu, "seconds_by_threads": res, "best_threads": int(best)}
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
