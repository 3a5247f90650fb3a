"""A/B of the phased slice kernel with the matrix instructions of all-padding m-tiles left out (policy 2582, <.., 32, 4, 2, ..>) against the
default (2581) on the two Phi GEMMs of the 1024^2 image (M = 899: 12 % of the matrix instructions are on padding rows).  Alternating rounds,
back to back launches, HIP events.   python tools/bench_gemm_skip_pad.py [out.json]"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from psalm_amd.hip_ops import get_ops


def main():
    ops = get_ops()
    out = {}
    quick = "--quick" in sys.argv                 # one shape, two rounds (for a kernel trace inside a few seconds)
    for M, N, K in ((899, 14336, 2048), (899, 2048, 10240), (1024, 14336, 2048))[:1 if quick else 3]:
        a = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") * 0.05
        asp, wsp = ops.split_f16(a), ops.split_f16(w)
        c = torch.empty(M, N, device="cuda")
        row = {"2581": [], "2582": []}
        ref = None
        for rnd in range(2 if quick else 4):
            for pol in (2581, 2582):
                ops.gemm_tile_policy(pol)
                try:
                    for _ in range(3):
                        ops.gemm_x3(asp, wsp, out=c)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(40):
                        ops.gemm_x3(asp, wsp, out=c)
                    e1.record()
                    torch.cuda.synchronize()
                    row[str(pol)].append(round(e0.elapsed_time(e1) / 40 * 1e3, 1))
                    row["kernel_" + str(pol)] = ops.gemm_last_kernel()
                    if ref is None:
                        ref = c.clone()
                    else:
                        row["identical"] = bool(torch.equal(ref, c)) and row.get("identical", True)
                finally:
                    ops.gemm_tile_policy(2581)
        out[f"M{M} N{N} K{K}"] = row
        print(M, N, K, row, flush=True)
    if len(sys.argv) > 1 and not sys.argv[1].startswith("--"):
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
