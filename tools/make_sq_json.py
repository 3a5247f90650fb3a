"""profiles/<round>_sq_summary.json from one SQ-counter pass (tools/rocpd_pmc.py --json of `rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python bench.py --eager ...`):
per kernel, the fractions tools/sq_fractions.py prints.  bench.py attaches the dominant kernel's entry to `roofline.sq_counters`.

    python tools/make_sq_json.py gpurun_out/r04j_pmc_sq.json profiles/r04_sq_summary.json "profiles/r04j_pmc_sq.json" """
import json
import sys


def main():
    src, dst, cite = sys.argv[1], sys.argv[2], sys.argv[3]
    d = json.load(open(src))
    out = {"source": cite, "how": "one rocprofv3 --pmc pass of the bench command (eager launches, single stream), per-dispatch means; fractions of "
                                  "SQ_WAVE_CYCLES: parked = SQ_WAIT_ANY (s_waitcnt / s_barrier), stalled = SQ_WAIT_INST_ANY (issue stall), issuing = "
                                  "SQ_ACTIVE_INST_ANY; matrix_pipe_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), in CYCLES; "
                                  "clock_GHz = GRBM_GUI_ACTIVE / 8 / the launch's duration in the pass (meaningful for launches of >= 100 us)",
           "kernels": {}}
    for name, row in d.items():
        per = lambda c: row[c]["per_dispatch"] if c in row else None
        wc, gui, us = per("SQ_WAVE_CYCLES"), per("GRBM_GUI_ACTIVE"), row.get("avg_duration_us_in_this_pass")
        if not wc or not gui:
            continue
        key = name.replace("void ", "").split("(")[0]
        e = {"launches_profiled": row["dispatches_profiled"], "parked": round(per("SQ_WAIT_ANY") / wc, 3), "stalled": round(per("SQ_WAIT_INST_ANY") / wc, 3),
             "issuing": round(per("SQ_ACTIVE_INST_ANY") / wc, 3),
             "matrix_pipe_busy": round((per("SQ_VALU_MFMA_BUSY_CYCLES") or 0.0) / (1024.0 * gui / 8.0), 4),
             "lds_bank_conflict": round((per("SQ_LDS_BANK_CONFLICT") or 0.0) / per("SQ_LDS_IDX_ACTIVE"), 3) if per("SQ_LDS_IDX_ACTIVE") else 0.0}
        if us:
            e["avg_launch_us_in_pass"] = round(us, 2)
            if us >= 100.0:
                e["clock_GHz"] = round(gui / 8.0 / (us * 1e3), 3)
        out["kernels"][key] = e
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print(f"{len(out['kernels'])} kernels -> {dst}")


if __name__ == "__main__":
    main()
