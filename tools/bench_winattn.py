"""Micro-benchmark of the fp32 window attention at the four Swin-B stages of a 1024^2 image (12 x 12 windows, head dim 32).
(The wavefronts-per-(window, head) choice it was written to A/B -- 3 when <= 320 pairs, else 1, profiles/r02n_winattn_nwv.jsonl -- is now
fixed in psalm_window_attention; the environment knob is gone.)"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from psalm_amd import hip_ops as H


def main():
    libs = [None]
    if "--libs" in sys.argv:                                     # two builds of the library on the same box, alternating (e.g. the r04 attention build)
        libs = sys.argv[sys.argv.index("--libs") + 1].split(",")
    for rnd in range(2):
        for lib in libs:
            ops = H.Ops(os.path.join(ROOT, lib)) if lib else H.get_ops()
            run(ops, {"round": rnd, "lib": lib or os.path.relpath(ops.lib_path, ROOT)})


def run(ops, tag):
    g = torch.Generator().manual_seed(0)
    out = dict(tag)
    for stage, (hw, heads) in enumerate(((256, 4), (128, 8), (64, 16), (32, 32))):
        nW = (hw + 11) // 12
        C = heads * 32
        rows = nW * nW * 144
        qkv = (torch.randn(rows, 3 * C, generator=g) * 0.5).cuda()
        table = torch.randn(23 * 23, heads, generator=g).cuda()
        for shift in (0, 6):
            for _ in range(3):
                o = ops.window_attention(qkv, table, 1, nW, nW, heads, 12, shift)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                o = ops.window_attention(qkv, table, 1, nW, nW, heads, 12, shift)
            e1.record()
            torch.cuda.synchronize()
            out[f"stage{stage}_shift{shift}"] = {"us": round(e0.elapsed_time(e1) * 1e3 / 30, 2), "checksum": float(o.double().sum())}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
