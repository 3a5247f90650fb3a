"""Experiment (not product): throughput with N images in flight on ONE GPU.

bench.py replays one image's hipGraph at a time; every kernel with a partial last wave of blocks (Phi [k|v|q|fc1]: 224 tiles on 256 CUs), every
single-wave launch whose blocks run their load / compute / store phases in lock step, and the ~100 latency-bound M = 100 GEMMs of the mask decoder
leave the chip partly idle.  Here N independent model instances (own activation buffers, own captured graphs) are driven by N host threads on
N HIP streams, so the hardware interleaves the launches of different images.

    python tools/exp_inflight.py [N=2] [steps=20]    -> one JSON line: sequential images/s, N-in-flight images/s"""
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    from psalm_amd.config import PsalmConfig
    from psalm_amd.model import PSALM
    from psalm_amd.synthetic import make_inputs, make_state_dict
    cfg = PsalmConfig(seg_task="panoptic")
    sd = make_state_dict(cfg, seed=0)
    models, inputs, streams = [], [], []
    for i in range(n):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            m = PSALM(cfg, sd, precision="f16x3", use_graphs=True)
            m.graph_outputs = "alias"
            inp = make_inputs(cfg, "panoptic", size=1024, batch=1, seed=i)
            inp["images"] = inp["images"].cuda()
            for _ in range(4):                            # eager, capture, 2 replays -- one model at a time (capture is process-global)
                m.eval_seg(**inp)
            torch.cuda.synchronize()
        models.append(m)
        inputs.append(inp)
        streams.append(s)

    def run_seq(k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(streams[0]):
            for _ in range(k):
                models[0].eval_seg(**inputs[0])
        torch.cuda.synchronize()
        return k / (time.perf_counter() - t0)

    def worker(i, k, bar):
        with torch.cuda.stream(streams[i]):
            bar.wait()
            for _ in range(k):
                models[i].eval_seg(**inputs[i])
            streams[i].synchronize()

    def run_par(k):
        bar = threading.Barrier(n + 1)
        th = [threading.Thread(target=worker, args=(i, k, bar)) for i in range(n)]
        for t in th:
            t.start()
        torch.cuda.synchronize()
        bar.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        return n * k / (time.perf_counter() - t0)

    seq = [run_seq(steps) for _ in range(3)]
    par = [run_par(steps) for _ in range(3)]
    print(json.dumps({"in_flight": n, "steps": steps, "sequential_img_s": [round(x, 2) for x in seq], "in_flight_img_s": [round(x, 2) for x in par],
                      "gain": round(max(par) / max(seq), 3)}))


if __name__ == "__main__":
    main()
