"""Micro-benchmark of the fp32 causal attention (Phi prefill shape of the bench workload: L = 899, 32 heads x 64, one [k|v|q] row buffer
of 3*2048 columns).  PSALM_ATTN_PAIR = 0 / 1: one query tile per block / a balanced pair (see causal_attention_f32_splitk_kernel); one run
per value:   for v in 0 1; do PSALM_ATTN_PAIR=$v python tools/bench_attn.py; done
(profiles/r02n_attn_variants.jsonl also holds the variants that were measured and dropped: V fragments from a contiguous / transposed
pre-pass copy (PSALM_ATTN_V), 8 wavefronts per block (PSALM_ATTN_NW), K-fragment prefetch + mask / rescale shortcuts (PSALM_ATTN_PF).)"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from psalm_amd import hip_ops as H


def main():
    L, heads, hd, rot = 899, 32, 64, 32
    Hd = heads * hd
    ops = H.Ops(os.environ["PSALM_LIB"]) if os.environ.get("PSALM_LIB") else H.get_ops()      # PSALM_LIB: an experiment build of the library
    g = torch.Generator().manual_seed(0)
    out = {}
    for ld in (3 * Hd, 3 * Hd + 8192):                       # fused-split layout / the [k|v|q|fc1] layout
        buf = (torch.randn(L, ld, generator=g) * 0.5).cuda()
        o = torch.zeros(L, Hd, device="cuda")
        inv = torch.arange(rot // 2, dtype=torch.float32)
        fr = torch.arange(L, dtype=torch.float32)[:, None] * (1.0 / (10000.0 ** (2 * inv / rot)))[None]
        emb = torch.cat((fr, fr), -1)
        cos, sin = emb.cos().contiguous().cuda(), emb.sin().contiguous().cuda()
        km = torch.ones(1, L, dtype=torch.uint8, device="cuda")
        for _ in range(5):
            ops.causal_attention(buf, 2 * Hd, 0, Hd, o, 0, cos, sin, km, 1, L, heads, hd, rot)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ops.causal_attention(buf, 2 * Hd, 0, Hd, o, 0, cos, sin, km, 1, L, heads, hd, rot)
        e1.record()
        torch.cuda.synchronize()
        out[f"ld{ld}"] = {"us_per_call": round(e0.elapsed_time(e1) * 1e3 / 50, 2), "checksum": float(o.double().sum())}
    print(json.dumps({"lib": os.path.basename(ops.lib_path), "PSALM_ATTN_PAIR": os.environ.get("PSALM_ATTN_PAIR", "1"), **out}))


if __name__ == "__main__":
    main()
