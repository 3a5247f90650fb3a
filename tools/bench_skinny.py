"""The mask decoder's M = 100 fp32 GEMMs (psalm_gemm on fp32 operands -> gemm_f32_skinny_kernel), per library: 33 launches per (library, shape),
meant to be read from a kernel trace (tools/rocpd_blocks.py <db> gemm_f32_skinny 33) -- the launches are shorter than a Python call.
    python tools/bench_skinny.py [--libs a.so,b.so]"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from psalm_amd import hip_ops as H

SHAPES = [(100, 256, 256), (100, 2048, 256), (100, 256, 2048), (100, 133, 256)]


def main():
    libs = [None]
    if "--libs" in sys.argv:
        libs = sys.argv[sys.argv.index("--libs") + 1].split(",")
    g = torch.Generator().manual_seed(0)
    data = [(torch.randn(M, K, generator=g).cuda(), torch.randn(N, K, generator=g).cuda(), torch.randn(N, generator=g).cuda(), torch.randn(M, N, generator=g).cuda()) for M, N, K in SHAPES]
    for rnd in range(2):
        for lib in libs:
            ops = H.Ops(os.path.join(ROOT, lib)) if lib else H.get_ops()
            row = {"round": rnd, "lib": lib or os.path.relpath(ops.lib_path, ROOT)}
            for (M, N, K), (a, w, b, r) in zip(SHAPES, data):
                out = torch.empty(M, N, device="cuda")
                for _ in range(33):
                    ops.gemm(a, w, bias=b, residual=r, out=out)
                torch.cuda.synchronize()
                row[f"M{M} N{N} K{K}"] = {"kernel": ops.gemm_last_kernel(), "checksum": float(out.double().sum())}
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
