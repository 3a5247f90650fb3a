"""profiles/rNN_pmc_hbm_traffic.json from the per-kernel PMC summary written by tools/rocpd_pmc.py (--json):

    python tools/make_traffic_json.py gpurun_out/final_pmc.json profiles/r01_pmc_hbm_traffic.json [source-label]

HBM bytes per launch = FETCH_SIZE x 2 (gfx950 wide-read correction, MI355X_MICROARCH.md §HBM; applied by rocpd_pmc.py as
`hbm_read_bytes_per_dispatch_corrected`) + WRITE_SIZE, KiB -> bytes, mean over all profiled dispatches.  rocprofv3 kernel names are
mapped to the instantiation names bench.py uses (`gemm_bf16_glds_kernel<{bf16|f32},BM,BN,WM,WN[,conv]>`; ring depth, BK and the
K-loop variant are not part of that name, so instantiations differing only there are merged, weighted by dispatch count)."""
import json
import re
import sys


def bench_name(rk):
    m = re.search(r"gemm_bf16_glds_kernel<([^>]*)>", rk)
    if m:
        a = [x.strip() for x in m.group(1).split(",")]
        tc = "bf16" if a[0] == "unsigned short" else "f32"
        conv = len(a) > 6 and a[6] == "true"
        x3 = len(a) > 10 and a[10] in ("true", "1", "2")           # split-f16 variant (template parameter X3; the split-output epilogue
        #                                                            variant, parameter SO, runs the same K loop and shares the name)
        return f"gemm_bf16_glds_kernel<{tc},{a[1]},{a[2]},{a[3]},{a[4]}{',conv' if conv else ''}{',x3' if x3 else ''}>"
    m = re.search(r"gemm_bf16_skinny_kernel<([^>]*)>", rk)
    if m:
        a = [x.strip() for x in m.group(1).split(",")]
        return f"gemm_bf16_skinny_kernel<{'bf16' if a[0] == 'unsigned short' else 'f32'}{',x3' if len(a) > 1 and a[1] == 'true' else ''}>"
    m = re.search(r"(\w+)<", rk) or re.search(r"(\w+)\(", rk)
    return m.group(1) if m else rk


def main():
    src, dst = sys.argv[1], sys.argv[2]
    label = sys.argv[3] if len(sys.argv) > 3 else src
    with open(src) as f:
        pmc = json.load(f)
    out = {}
    for rk, row in pmc.items():
        rd, wr = row.get("hbm_read_bytes_per_dispatch_corrected"), row.get("hbm_write_bytes_per_dispatch")
        if rd is None or wr is None:
            continue
        n = row.get("dispatches_profiled", 1)
        e = out.setdefault(bench_name(rk), {"rocprof_kernels": [], "dispatches_profiled": 0, "_rd": 0.0, "_wr": 0.0})
        e["rocprof_kernels"].append(rk)
        e["dispatches_profiled"] += n
        e["_rd"] += rd * n
        e["_wr"] += wr * n
    for e in out.values():
        n = e["dispatches_profiled"]
        e["hbm_read_bytes_per_launch"] = round(e.pop("_rd") / n)
        e["hbm_write_bytes_per_launch"] = round(e.pop("_wr") / n)
        e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]
    doc = {"method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, no trace domains) on `python bench.py --steps 2 "
                     "--warmup 0 --no-cpu-baseline --eager`; KiB -> bytes; FETCH_SIZE doubled (gfx950 under-counts wide coalesced reads "
                     "by 2x, MI355X_MICROARCH.md §HBM); per-dispatch mean over all launches of the kernel",
           "source": label, "kernels": out}
    with open(dst, "w") as f:
        json.dump(doc, f, indent=1)
    print(f"{len(out)} kernels -> {dst}")


if __name__ == "__main__":
    main()
