"""Micro-benchmark of the fp32 causal attention (Phi prefill shape of the bench workload: 32 heads x 64, one [k|v|q] row buffer; L = 899 and the
bucketed 928) -- per library, so that two builds of the kernel are compared on the same box, back to back, with a checksum of the output:
    python tools/bench_attn.py [--libs psalm_amd/lib/libpsalm_hip.so,tools/experiments/_build/libpsalm_hip_r04attn.so]
(profiles/r02n_attn_variants.jsonl holds the variants measured and dropped in r02: V fragments from a contiguous / transposed pre-pass copy,
8 wavefronts per block, K-fragment prefetch.)"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from psalm_amd import hip_ops as H


def main():
    heads, hd, rot = 32, 64, 32
    Hd = heads * hd
    libs = [None]
    if "--libs" in sys.argv:
        libs = sys.argv[sys.argv.index("--libs") + 1].split(",")
    for rnd in range(2):                                         # two alternating rounds: clock / box drift shows up as a difference between them
        for lib in libs:
            ops = H.Ops(os.path.join(ROOT, lib)) if lib else H.get_ops()
            g = torch.Generator().manual_seed(0)
            out = {}
            for L in (899, 928):
                ld = 3 * Hd
                buf = (torch.randn(L, ld, generator=g) * 0.5).cuda()
                o = torch.zeros(L, Hd, device="cuda")
                inv = torch.arange(rot // 2, dtype=torch.float32)
                fr = torch.arange(L, dtype=torch.float32)[:, None] * (1.0 / (10000.0 ** (2 * inv / rot)))[None]
                emb = torch.cat((fr, fr), -1)
                cos, sin = emb.cos().contiguous().cuda(), emb.sin().contiguous().cuda()
                km = torch.ones(1, L, dtype=torch.uint8, device="cuda")
                if L == 928:
                    km[0, 899:] = 0                              # the bucketed prompt: 29 masked positions behind the real ones
                for _ in range(5):
                    ops.causal_attention(buf, 2 * Hd, 0, Hd, o, 0, cos, sin, km, 1, L, heads, hd, rot)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(100):
                    ops.causal_attention(buf, 2 * Hd, 0, Hd, o, 0, cos, sin, km, 1, L, heads, hd, rot)
                e1.record()
                torch.cuda.synchronize()
                out[f"L{L}"] = {"us_per_call_incl_rope_prepass": round(e0.elapsed_time(e1) * 1e3 / 100, 2), "checksum": float(o[:899].double().sum()),
                                "abs_checksum": float(o[:899].double().abs().sum())}
            print(json.dumps({"round": rnd, "lib": lib or os.path.relpath(ops.lib_path, ROOT), **out}), flush=True)


if __name__ == "__main__":
    main()
