"""Micro-benchmark of the mask decoder's fp32 cross-attention (psalm_mha_attention_f32: 100 queries x 8 heads x 32 over the three pixel-decoder
levels of a 1024^2 image: 1024 / 4096 / 16384 keys, masked) per library -- two builds compared on the same box, alternating, with a bitwise
comparison of their outputs:
    python tools/bench_mha.py [--libs psalm_amd/lib/libpsalm_hip.so,tools/experiments/_build/libpsalm_hip_r04attn.so]"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from psalm_amd import hip_ops as H


def main():
    libs = [None]
    if "--libs" in sys.argv:
        libs = sys.argv[sys.argv.index("--libs") + 1].split(",")
    Q, heads, D = 100, 8, 256
    g = torch.Generator().manual_seed(0)
    cases = {}
    for Lk in (1024, 4096, 16384):
        q = torch.randn(Q, D, generator=g).cuda()
        k = torch.randn(Lk, D, generator=g).cuda()
        v = torch.randn(Lk, D, generator=g).cuda()
        mask = (torch.rand(1, Q, Lk, generator=g) < 0.9).to(torch.uint8)
        mask[0, 5] = 1                                           # an all-masked row (TD:647)
        ram = (mask.sum(-1) == Lk).to(torch.uint8).contiguous().cuda()
        cases[Lk] = (q, k, v, mask.cuda(), ram)
    outs = {}
    for rnd in range(2):
        for lib in libs:
            ops = H.Ops(os.path.join(ROOT, lib)) if lib else H.get_ops()
            row = {}
            for Lk, (q, k, v, mask, ram) in cases.items():
                for _ in range(5):
                    o = ops.mha_attention(q, k, v, 1, Q, Lk, heads, mask, ram)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(100):
                    o = ops.mha_attention(q, k, v, 1, Q, Lk, heads, mask, ram)
                e1.record()
                torch.cuda.synchronize()
                row[f"Lk{Lk}"] = {"us_per_call_incl_combine": round(e0.elapsed_time(e1) * 10, 2), "checksum": float(o.double().sum())}
                outs.setdefault(Lk, {})[lib] = o.clone()
            print(json.dumps({"round": rnd, "lib": lib or os.path.relpath(ops.lib_path, ROOT), **row}), flush=True)
    if len(libs) > 1:
        print(json.dumps({"bitwise_equal_across_libs": {str(Lk): bool(all(torch.equal(o_[libs[0]], o_[l]) for l in libs[1:])) for Lk, o_ in outs.items()}}))


if __name__ == "__main__":
    main()
