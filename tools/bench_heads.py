"""Mask-decoder head kernels on the GPU: psalm_ln_mlp3 / psalm_linear_res_ln (weights staged through LDS vs per-lane fragment loads)
against the separate launches they replace (LayerNorm + 3 GEMMs; GEMM + LayerNorm), with a bitwise cross-check of the two variants.
    python tools/bench_heads.py [--json out.json]"""
import json
import sys

import torch

sys.path.insert(0, ".")
from psalm_amd import hip_ops as H  # noqa: E402
from psalm_amd.hip_ops import get_ops  # noqa: E402


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    ops = get_ops()
    g = torch.Generator().manual_seed(0)
    Q, D = 100, 256
    x = (torch.randn(Q, D, generator=g) * 2).cuda()
    ga, be = torch.randn(D, generator=g).cuda(), torch.randn(D, generator=g).cuda()
    ws = [(torch.randn(D, D, generator=g) * D ** -0.5).bfloat16().cuda() for _ in range(3)]
    bs = [torch.randn(D, generator=g).cuda() for _ in range(3)]
    a = torch.randn(Q, D, generator=g).bfloat16().cuda()
    qe = torch.randn(Q, D, generator=g).cuda()
    o2 = torch.empty(Q, D, dtype=torch.bfloat16, device="cuda")
    o3 = torch.empty(Q, D, dtype=torch.bfloat16, device="cuda")
    res = {}

    def unfused_head():
        y = ops.layernorm(x, ga, be, out_dtype=torch.bfloat16)
        for j in range(3):
            y = ops.gemm(y, ws[j], bs[j], act=0 if j == 2 else H.ACT_RELU, out_dtype=torch.bfloat16)
        return y

    def unfused_tail():
        return ops.layernorm(ops.gemm(a, ws[0], bs[0], residual=x, out_dtype=torch.float32), ga, be, out2=o2, add=qe, out3=o3)

    res["head_unfused_us"] = round(timeit(unfused_head), 2)
    res["tail_unfused_us"] = round(timeit(unfused_tail), 2)
    outs = {}
    for staged in (True, False):
        ops.heads_variant(staged)
        tag = "staged" if staged else "direct"
        res[f"head_{tag}_us"] = round(timeit(lambda: ops.ln_mlp3(x, ga, be, ws, bs)), 2)
        res[f"tail_{tag}_us"] = round(timeit(lambda: ops.linear_res_ln(a, ws[0], bs[0], x, ga, be, out2=o2, add=qe, out3=o3)), 2)
        outs[tag] = (ops.ln_mlp3(x, ga, be, ws, bs)[1].clone(), ops.linear_res_ln(a, ws[0], bs[0], x, ga, be).clone())
    ops.heads_variant(True)
    res["head_staged_eq_direct"] = bool(torch.equal(outs["staged"][0], outs["direct"][0]))
    res["tail_staged_eq_direct"] = bool(torch.equal(outs["staged"][1], outs["direct"][1]))
    res["head_vs_unfused_maxdiff"] = float((outs["staged"][0].float() - unfused_head().float()).abs().max())
    res["tail_vs_unfused_maxdiff"] = float((outs["staged"][1] - unfused_tail()).abs().max())
    mism = 0
    for _ in range(100):                                     # repeated launches against the first result (race screen)
        if not torch.equal(ops.ln_mlp3(x, ga, be, ws, bs)[1], outs["staged"][0]) or \
           not torch.equal(ops.linear_res_ln(a, ws[0], bs[0], x, ga, be), outs["staged"][1]):
            mism += 1
    res["staged_repeat_mismatches"] = f"{mism}/100"
    print(json.dumps(res))
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
