"""Split-f16 (X3) GEMM micro-benchmark over the shapes of the f16x3 image, per tile policy.
    python tools/bench_gemm_x3.py [--ring] [--lib path/to/other/libpsalm_hip.so] [out.json]"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from psalm_amd.hip_ops import get_ops

SHAPES = [(899, 14336, 2048), (899, 2048, 10240), (4096, 2048, 512), (4096, 512, 2048), (5184, 1536, 512), (5184, 512, 512),
          (21504, 1024, 256), (21504, 256, 1024), (21504, 256, 256), (65536, 512, 128), (65536, 128, 512), (16384, 1024, 256),
          (1024, 4096, 1024), (100, 65536, 256), (65536, 256, 2304),
          (21504, 288, 256), (1296, 3072, 1024), (17424, 768, 256), (69696, 384, 128), (256, 2048, 18432), (1024, 1024, 4096)]
# 3301 / 3302: slice forms with one block per CU (r02l: lose); 3303 / 3304: 32-deep slices in a 2- / 3-deep ring (the K-panel form's LDS footprint)
# r05 (--ring): the ring depths that spend the LDS of a second resident block on prefetch distance: 3307 = 64 x 128, 64-deep slices, 3 stages
# (144 KB); 3308 = 128 x 128, 32-deep slices, 3 stages (96 KB); 3302 = 32-deep slices, 4 stages
POLICIES = [("auto", [0]), ("auto_s3", [3303]), ("t128", [128]), ("t128_s3", [128, 3303]), ("t64", [64]), ("t64_s3", [64, 3303]), ("t64_s4", [64, 3304]),
            ("t64_slice", [64, 3301]), ("t256", [256]), ("auto_ps", [2581]), ("t256_ps", [256, 2581])]     # _ps: 256 x 256 phased loop on 32-deep slices (r04)
RING = [("auto", [0]), ("t64_k32n2", [64, 3303]), ("t64_k32n3", [64, 3304]), ("t64_k32n4", [64, 3302]), ("t64_k64n2", [64, 3301]), ("t64_k64n3", [64, 3307]),
        ("t128_k32n2", [128, 3303]), ("t128_k32n3", [128, 3308]), ("t128_k32n4", [128, 3302]), ("t128_k64n2", [128, 3301])]
RING_SHAPES = [(4096, 512, 2048), (4096, 2048, 512), (5184, 1536, 512), (5184, 512, 512), (21504, 1024, 256), (21504, 256, 1024), (21504, 256, 256),
               (100, 65536, 256), (65536, 512, 128), (21504, 288, 256), (16384, 1024, 256), (69696, 384, 128), (1024, 1024, 4096), (1024, 4096, 1024),
               (65536, 128, 512), (16384, 256, 1024), (17424, 768, 256), (1296, 3072, 1024), (16384, 768, 256), (1296, 1024, 1024), (256, 2048, 18432),
               (65536, 256, 256), (69696, 128, 128), (17424, 256, 256), (4096, 512, 1024), (4096, 768, 256), (16384, 256, 512), (16384, 256, 256)]


# r06 (--mid): the mid-size forms of the slice GEMM (psalm_gemm_set_tile_policy 4400 + form) against the r05 kernels (4409), fp32 output and -- on
# the shapes the image runs with it -- paired split-f16 output behind GELU
MID = [("r05", [4409]), ("ilv", [4401]), ("256x128_ilv", [4402]), ("256x128", [4404]), ("128x256", [4405]), ("256x128_lw2", [4406]), ("128x256_lw2", [4407]),
       ("256x128_lw4", [4408]), ("128x128_lw2", [4410]), ("64x128_lw2", [4411])]
# r06 (--midsplit): loader-wave blocks + split-K (forms 12 / 13) on the long-K problems whose large tiles alone cannot fill the chip
MIDSPLIT = [("r05", [4409]), ("auto", [4400]), ("256x128_lw4_split", [4412]), ("128x128_lw2_split", [4413])]
MIDSPLIT_SHAPES = [(4096, 512, 2048), (4096, 512, 1024), (1296, 1024, 1024), (1024, 4096, 1024), (1296, 3072, 1024), (5184, 512, 512), (1024, 1024, 4096),
                   (1024, 1024, 2048), (256, 2048, 2048), (256, 2048, 1024), (1024, 256, 1024), (16384, 256, 1024), (4096, 256, 512)]
MID_SO = {(4096, 2048, 512), (21504, 1024, 256), (65536, 512, 128), (16384, 1024, 256), (1024, 4096, 1024)}


def main():
    ops = get_ops()
    if "--lib" in sys.argv:                                     # A/B against a side library (tools/experiments/_build/...)
        i = sys.argv.index("--lib")
        from psalm_amd.hip_ops import Ops
        ops = Ops(os.path.join(ROOT, sys.argv[i + 1]))
        del sys.argv[i:i + 2]
    out = {}
    ring = "--ring" in sys.argv
    if ring:
        sys.argv.remove("--ring")
    mid = "--mid" in sys.argv
    if mid:
        sys.argv.remove("--mid")
    midsplit = "--midsplit" in sys.argv
    if midsplit:
        sys.argv.remove("--midsplit")
    global POLICIES
    if ring:
        POLICIES = RING
    if mid:
        POLICIES = MID
    if midsplit:
        POLICIES = MIDSPLIT
    shapes = MIDSPLIT_SHAPES if midsplit else (RING_SHAPES if (ring or mid) else SHAPES)
    if mid:
        shapes = [(M, N, K, so) for (M, N, K) in shapes if M > 192 for so in ((False, True) if (M, N, K) in MID_SO else (False,))]
    else:
        shapes = [(M, N, K, False) for (M, N, K) in shapes]
    if "--shapes" in sys.argv:                                 # --shapes "256,2048,18432;256,2048,9216": only these (fp32 output)
        i = sys.argv.index("--shapes")
        shapes = [tuple(int(v) for v in t.split(",")) + (False,) for t in sys.argv[i + 1].split(";")]
        del sys.argv[i:i + 2]
        if ring:
            POLICIES = POLICIES + [("auto_default", [0]), ("t128", [128]), ("t64_slice64", [64, 3301]), ("t128_slice64", [128, 3301]), ("t256", [256])]
    for M, N, K, so_out in shapes:
        a = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda")
        asp, wsp = ops.split_f16(a), ops.split_f16(w)
        c = torch.empty(M, N, device="cuda")
        if so_out:
            bias = torch.randn(N, device="cuda")
            so = torch.zeros(M, 2 * N, dtype=torch.float16, device="cuda")
            so_inv = torch.empty(M, device="cuda")
            bnd = torch.tensor([2.0 ** 14 * float(w.abs().sum(1).max()), float(bias.abs().max()), 0.0, 0.0], device="cuda")

            def launch():
                ops.gemm_x3_split(asp, wsp, bias, 2, so, so_inv, bnd, split_col_off=0, split_col_start=0, act_col_start=0, paired=True)
        else:
            def launch():
                ops.gemm_x3(asp, wsp, out=c)
        row = {}
        for name, pol in POLICIES:
            for p in pol:
                ops.gemm_tile_policy(p)
            try:
                for _ in range(3):
                    launch()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 30
                e0.record()
                for _ in range(reps):
                    launch()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / reps * 1e3
                row[name] = {"us": round(us, 1), "TF_alg": round(2.0 * M * N * K / us / 1e6, 1), "describe": ops.gemm_describe(M, N, 3 * asp.Kp, x3=True), "kernel": ops.gemm_last_kernel()}
            except Exception as ex:  # noqa
                row[name] = {"error": str(ex)[:80]}
            finally:
                ops.gemm_tile_policy(1282)
                ops.gemm_tile_policy(640)
                ops.gemm_tile_policy(3300)
                ops.gemm_tile_policy(2582 if (mid or midsplit) else 2580)
                ops.gemm_tile_policy(4400)
                ops.gemm_tile_policy(0)
        out[f"M{M} N{N} K{K}" + (" so" if so_out else "")] = row
        print(M, N, K, "so" if so_out else "", {k: v.get("us") for k, v in row.items()}, flush=True)
        del a, w, asp, wsp, c
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
