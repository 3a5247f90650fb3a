"""profiles/rNN_pmc_hbm_traffic.json from the per-kernel PMC summary written by tools/rocpd_pmc.py (--json):

    python tools/make_traffic_json.py gpurun_out/final_pmc.json profiles/r01_pmc_hbm_traffic.json [source-label]

HBM bytes per launch = FETCH_SIZE x 2 (gfx950 wide-read correction, MI355X_MICROARCH.md §HBM; applied by rocpd_pmc.py as
`hbm_read_bytes_per_dispatch_corrected`) + WRITE_SIZE, KiB -> bytes, mean over all profiled dispatches.  rocprofv3 kernel names are
keyed by the template instantiation as the trace spells it (what bench.py gets from psalm_gemm_last_kernel)."""
import json
import re
import sys


def bench_name(rk):
    """rocprofv3 kernel name -> the key bench.py uses: the template instantiation exactly as the trace spells it (bench.py gets the same
    string from psalm_gemm_last_kernel), without the `void ` prefix and the argument list; plain function name for non-templates."""
    m = re.search(r"(\w+<.*>)\s*\(", rk) or re.search(r"(\w+<.*>)", rk)
    if m:
        return m.group(1)
    m = re.search(r"(\w+)\(", rk)
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
