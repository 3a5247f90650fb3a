"""Experiment (not product): block-level time line of the direct-to-LDS GEMM kernel on the GEMM shapes of the f16x3 image.

A SEPARATE copy of the library is built with -DPSALM_GEMM_TIMELINE (psalm_amd/csrc/gemm.hip: PSALM_TL stamps s_memrealtime, 100 MHz, by thread
0 of every block at: 0 entry, 1 prologue copies issued, 2 first K tile visible, 3 K loop done, 4 tile transposed into LDS, 5 stores drained).

    python tools/experiments/gemm_timeline.py --build          (authoring container: hipcc cross-compile -> tools/experiments/_build/)
    python tools/experiments/gemm_timeline.py [out.json]       (GPU box)

Per shape and tile policy: the launch's wall time from events, the span first-entry -> last-stamp, and the median / p90 over blocks of each
phase -- i.e. whether a launch is prologue- (cold operands), K-loop- (copy latency per step) or epilogue- (store tail) bound, and how far the
block starts are spread."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "experiments", "_build")
LIB = os.path.join(OUT, "libpsalm_hip_tl.so")


def build():
    from psalm_amd import build as B
    B.build()                                            # product objects (everything except gemm.o is reused)
    os.makedirs(OUT, exist_ok=True)
    obj = os.path.join(OUT, "gemm_tl.o")
    src = os.path.join(B.CSRC, "gemm.hip")
    if not os.path.exists(obj) or os.path.getmtime(src) > os.path.getmtime(obj):
        subprocess.check_call([B.HIPCC] + B.FLAGS + ["-DPSALM_GEMM_TIMELINE", "-c", src, "-o", obj])
    objs = [o for o in glob.glob(os.path.join(B.LIBDIR, "obj", "*.o")) if os.path.basename(o) != "gemm.o"] + [obj]
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    print(LIB)


# (M, N, K, split-f16 output from column (None: fp32 output), activation)
SHAPES = [(899, 14336, 2048, None, 0), (899, 14336, 2048, 6144, 3), (899, 14336, 2048, 6144, 3, 1), (899, 2048, 10240, None, 0),
          (899, 2048, 10240, None, 0, 1), (4096, 2048, 512, None, 0),
          (4096, 2048, 512, 0, 2), (4096, 512, 2048, None, 0), (5184, 1536, 512, None, 0), (5184, 512, 512, None, 0),
          (21504, 1024, 256, None, 0), (21504, 1024, 256, 0, 1), (21504, 256, 1024, None, 0), (21504, 256, 256, None, 0),
          (65536, 512, 128, None, 0), (100, 65536, 256, None, 0)]
POLICIES = [("auto", [0]), ("t128", [128]), ("t64", [64]), ("t256", [256])]


def main():
    global SHAPES
    if os.environ.get("TL_SHAPES") == "paired":          # the split-f16-output epilogues, LDS transpose vs paired stores
        SHAPES = [(899, 14336, 2048, 6144, 3, 1), (899, 14336, 2048, 6144, 3, 1, 1), (899, 14336, 2048, None, 0, 1), (4096, 2048, 512, 0, 2), (4096, 2048, 512, 0, 2, 0, 1),
                  (4096, 2048, 512, None, 0), (21504, 1024, 256, 0, 1), (21504, 1024, 256, 0, 1, 0, 1)]
    import ctypes
    import torch
    from psalm_amd import hip_ops as H
    ops = H.Ops(LIB)
    nslot = 1 << 16
    tl = torch.zeros(nslot * 8, dtype=torch.int64, device="cuda")
    assert ops._cdll_raw.psalm_gemm_timeline_buffer(ctypes.c_void_p(tl.data_ptr())) == 0
    out = {}
    for shape in SHAPES:
        M, N, K, so_from, act = shape[:5]
        paired = len(shape) > 6 and shape[6]                 # paired split-f16 stores (timing only: the W rows are NOT permuted here)
        a = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") * 0.05
        asp, wsp = ops.split_f16(a), ops.split_f16(w)
        c = torch.empty(M, N, device="cuda")
        bias = torch.randn(N, device="cuda")
        if so_from is not None:                              # the model's fused form: columns >= so_from leave as the next GEMM's split-f16 operand
            kp_out = (N - so_from + 2048 + 63) // 64 * 64
            so = torch.zeros(M, 2 * kp_out, dtype=torch.float16, device="cuda")
            so_inv = torch.empty(M, device="cuda")
            bnd = torch.tensor([2.0 ** 14 * float(w.abs().sum(1).max()), float(bias.abs().max()), 0.0, 0.0], device="cuda")

            def launch():
                ops.gemm_x3_split(asp, wsp, bias, act, so, so_inv, bnd, split_col_off=2048, split_col_start=so_from, act_col_start=so_from, out=c,
                                  paired=bool(paired))
        else:
            def launch():
                ops.gemm_x3(asp, wsp, out=c)
        big = torch.empty(64 << 20, device="cuda")        # 256 MB: evicts the operands from the Infinity Cache between cold launches
        row = {}
        for name, pol in (POLICIES[:1] if os.environ.get("TL_SHAPES") else POLICIES):
            for p in pol:
                ops.gemm_tile_policy(p)
            try:
                desc = ops.gemm_describe(M, N, 3 * asp.Kp, x3=True)
                res = {}
                for mode in ("warm", "cold"):
                    for _ in range(2):
                        launch()
                    if mode == "cold":
                        big.fill_(1.0)
                    tl.zero_()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    launch()
                    e1.record()
                    torch.cuda.synchronize()
                    t = tl.view(nslot, 8).cpu()
                    t = t[t[:, 0] > 0][:, :6].double()
                    if desc[3] > 1:                          # split-K: the reduce kernel is outside the stamped kernel
                        pass
                    t0 = t[:, 0].min()
                    us = (t - t0) / 100.0                     # 100 MHz -> us
                    ph = us[:, 1:] - us[:, :-1]

                    def q(v, f):
                        return round(float(v.quantile(f)), 2)
                    res[mode] = {"event_us": round(e0.elapsed_time(e1) * 1e3, 1), "blocks": int(t.shape[0]),
                                 "span_us": round(float(us[:, 5].max()), 2),
                                 "start_p50_p90_max": [q(us[:, 0], .5), q(us[:, 0], .9), round(float(us[:, 0].max()), 2)],
                                 "end_p10_p50": [q(us[:, 5], .1), q(us[:, 5], .5)],
                                 "phase_p50": {n_: q(ph[:, i], .5) for i, n_ in enumerate(("setup+issue", "first_tile", "k_loop", "to_lds", "store"))},
                                 "phase_p90": {n_: q(ph[:, i], .9) for i, n_ in enumerate(("setup+issue", "first_tile", "k_loop", "to_lds", "store"))},
                                 "block_total_p50": q(us[:, 5] - us[:, 0], .5)}
                row[name] = {"describe": list(desc), **res}
            except Exception as ex:  # noqa
                row[name] = {"error": str(ex)[:100]}
            finally:
                for p in (1282, 640, 3300, 0):
                    ops.gemm_tile_policy(p)
        key = f"M{M} N{N} K{K}" + (f" so>={so_from} act{act}" if so_from is not None else "") + (" paired" if paired else "")
        out[key] = row
        print(key, json.dumps(row), flush=True)
        del a, w, asp, wsp, c, big
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


def ablate_mid():
    """r06: the same switches in the GENERIC slice loop (64 x 128 / 128 x 128 tiles of r05, 256 x 128 of r06) on the mid-size shapes of the image:
    which resource a K step of these launches waits for.  One JSON line per (shape, form, mask)."""
    import ctypes
    import torch
    from psalm_amd import hip_ops as H
    ops = H.Ops(LIB)
    nslot = 1 << 16
    tl = torch.zeros(nslot * 8, dtype=torch.int64, device="cuda")
    assert ops._cdll_raw.psalm_gemm_timeline_buffer(ctypes.c_void_p(tl.data_ptr())) == 0
    for (M, N, K) in ((4096, 2048, 512), (4096, 512, 2048), (5184, 512, 512), (21504, 256, 1024), (21504, 256, 256), (65536, 256, 256)):
        a = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") * 0.05
        asp, wsp = ops.split_f16(a), ops.split_f16(w)
        c = torch.empty(M, N, device="cuda")
        for form in (4409, 4404):
            ops.gemm_tile_policy(form)
            try:
                for mask in (0, 1, 2, 4, 8, 3, 5, 6, 7, 9, 15):
                    assert ops._cdll_raw.psalm_gemm_ablate(mask) == 0
                    for _ in range(3):
                        ops.gemm_x3(asp, wsp, out=c)
                    tl.zero_()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    ops.gemm_x3(asp, wsp, out=c)
                    e1.record()
                    torch.cuda.synchronize()
                    t = tl.view(nslot, 8).cpu()
                    t = t[t[:, 0] > 0][:, :6].double()
                    us = (t - t[:, 0].min()) / 100.0
                    ph = us[:, 1:] - us[:, :-1]
                    row = {"shape": [M, N, K], "form": form, "kernel": ops.gemm_last_kernel(), "ablate": mask, "blocks": int(t.shape[0]),
                           "off": [n for b, n in ((1, "copies"), (2, "frag_reads"), (4, "mfma"), (8, "barriers")) if mask & b],
                           "event_us": round(e0.elapsed_time(e1) * 1e3, 1), "k_loop_p50_us": round(float(ph[:, 2].quantile(.5)), 2),
                           "k_loop_p90_us": round(float(ph[:, 2].quantile(.9)), 2), "setup_p50": round(float(ph[:, 0].quantile(.5)), 2),
                           "first_tile_p50": round(float(ph[:, 1].quantile(.5)), 2), "epilogue_p50": round(float((ph[:, 3] + ph[:, 4]).quantile(.5)), 2),
                           "span_us": round(float(us[:, 5].max()), 2)}
                    print(json.dumps(row), flush=True)
            finally:
                ops._cdll_raw.psalm_gemm_ablate(0)
                ops.gemm_tile_policy(4400)
        del a, w, asp, wsp, c


def ablate():
    """K-loop time of the slice-form phased kernel (policy 2581) with parts of the loop switched off (psalm_gemm_ablate, experiment build):
    1 copies, 2 fragment reads, 4 matrix instructions, 8 phase barriers.  One JSON line per (shape, mask)."""
    import ctypes
    import torch
    from psalm_amd import hip_ops as H
    ops = H.Ops(LIB)
    nslot = 1 << 16
    tl = torch.zeros(nslot * 8, dtype=torch.int64, device="cuda")
    assert ops._cdll_raw.psalm_gemm_timeline_buffer(ctypes.c_void_p(tl.data_ptr())) == 0
    out = []
    for (M, N, K, pol) in ((899, 14336, 2048, [2581]), (899, 2048, 10240, [256, 2581]), (4096, 4096, 4096, [256, 2581]), (899, 14336, 2048, [2580])):
        a = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") * 0.05
        asp, wsp = ops.split_f16(a), ops.split_f16(w)
        c = torch.empty(M, N, device="cuda")
        for p in pol:
            ops.gemm_tile_policy(p)
        try:
            for mask in ((0,) if 2580 in pol else (0, 1, 2, 4, 8, 3, 5, 6, 9, 7, 11, 13, 14, 15)):
                assert ops._cdll_raw.psalm_gemm_ablate(mask) == 0
                for _ in range(3):
                    ops.gemm_x3(asp, wsp, out=c)
                tl.zero_()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.gemm_x3(asp, wsp, out=c)
                e1.record()
                torch.cuda.synchronize()
                t = tl.view(nslot, 8).cpu()
                t = t[t[:, 0] > 0][:, :6].double()
                us = (t - t[:, 0].min()) / 100.0
                ph = us[:, 1:] - us[:, :-1]
                row = {"shape": [M, N, K], "policy": pol, "kernel": ops.gemm_last_kernel(), "ablate": mask,
                       "off": [n for b, n in ((1, "copies"), (2, "frag_reads"), (4, "mfma"), (8, "barriers")) if mask & b],
                       "event_us": round(e0.elapsed_time(e1) * 1e3, 1), "k_loop_p50_us": round(float(ph[:, 2].quantile(.5)), 2),
                       "k_loop_p90_us": round(float(ph[:, 2].quantile(.9)), 2), "span_us": round(float(us[:, 5].max()), 2)}
                out.append(row)
                print(json.dumps(row), flush=True)
        finally:
            ops._cdll_raw.psalm_gemm_ablate(0)
            for p in (2580, 0):
                ops.gemm_tile_policy(p)
        del a, w, asp, wsp, c
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    elif "--ablate-mid" in sys.argv:
        ablate_mid()
    elif "--ablate" in sys.argv:
        ablate()
    else:
        main()
