"""Per-precision-mode timing + parity at the bench workload (1024^2 panoptic), reference = the exact-fp32 GPU mode (== CPU oracle to ~2e-6).
    python tools/exp_modes.py [size] [modes,comma,separated]"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from psalm_amd.config import PsalmConfig
from psalm_amd.model import PSALM
from psalm_amd.synthetic import make_inputs, make_state_dict
from tools._metrics import metrics, clone


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["f16x3", "bf16"]
    cfg = PsalmConfig(seg_task="panoptic")
    sd = make_state_dict(cfg, seed=0)
    inputs = make_inputs(cfg, "panoptic", size=size, batch=1, seed=0)
    inputs["images"] = inputs["images"].cuda()
    m32 = PSALM(cfg, sd, precision="fp32")
    ref = clone(m32.eval_seg(**inputs)[0])
    del m32
    torch.cuda.empty_cache()
    out = {}
    for mode in modes:
        parts = mode.split("+")                      # e.g. f16x3+overlap, f16x3+overlap+nofuse (split-f16 outputs of GEMMs / attention off)
        nofuse = "nofuse" in parts                    # (then: un-permuted fc1 rows and three f16 products in the Phi GEMMs too)
        m = PSALM(cfg, sd, precision=parts[0], use_graphs=True, **({"paired_split_stores": False} if nofuse else {"paired_split_stores": False} if "nopair" in parts else {}))
        m.overlap_streams = "overlap" in parts
        if nofuse:
            m.fuse_split = False
        for _ in range(3):
            r = m.eval_seg(**inputs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            r = m.eval_seg(**inputs)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 100
        got = clone(r[0])
        m.use_graphs = False
        m.eval_seg(**inputs)
        recs = []
        m.ops.lib.records = recs
        m.eval_seg(**inputs)
        torch.cuda.synchronize()
        m.ops.lib.records = None
        agg = {}
        for name, a, e0, e1, *_ in recs:
            d = agg.setdefault(name, [0, 0.0]); d[0] += 1; d[1] += e0.elapsed_time(e1)
        while mode in out:
            mode += "'"                             # repeated modes (interleaved A/B runs) keep separate entries
        out[mode] = {"ms_per_image_graph": round(ms, 3), "images_per_s": round(1e3 / ms, 2), "vs_fp32": metrics(got, ref),
                     "breakdown_ms": {k: [v[0], round(v[1], 3)] for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]}}
        print(mode, json.dumps(out[mode]), flush=True)
        del m
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
