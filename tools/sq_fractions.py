"""Where the wavefront cycles of the top kernels go: per-kernel fractions from ONE rocprofv3 --pmc pass of SQ counters
(tools/rocpd_pmc.py --json output), following /opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots":

    SQ_WAIT_ANY       wavefront parked (s_waitcnt / s_barrier)            \
    SQ_WAIT_INST_ANY  issue stall (matrix-pipe RAW, busy pipe)             >  disjoint, together ~ SQ_WAVE_CYCLES (quad-cycles)
    SQ_ACTIVE_INST_ANY an instruction of the wavefront is issuing         /
    SQ_VALU_MFMA_BUSY_CYCLES  matrix-pipe busy cycles, summed over SIMDs  (32 per v_mfma_f32_32x32x16_f16)
    SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE   extra LDS cycles / all LDS-array cycles
    GRBM_GUI_ACTIVE   shader-engine active cycles, summed over the 8 XCDs

    python tools/sq_fractions.py <pmc.json> [--top N]  ->  one line per kernel (text table)

matrix_pipe_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): the fraction of the kernel's wall cycles the matrix pipes
of the whole chip were busy, in CYCLES -- `roofline.mfma_issue.frac_of_f16_peak` is the same quantity against the 2.4 GHz the 2.5 PFLOP/s peak
assumes; the two differ by the clock the launch actually ran at (column GHz = GRBM_GUI_ACTIVE / 8 / the launch's duration in this pass)."""
import json
import sys


def main():
    path = sys.argv[1]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 14
    d = json.load(open(path))
    print(f"{'disp':>5} {'parked':>7} {'stall':>7} {'issuing':>7} {'mfma':>6} {'lds_cf':>6} {'us':>8} {'GHz':>5}  kernel")
    for name, row in list(d.items())[:top]:
        def per(c):
            v = row.get(c)
            return v["per_dispatch"] if v else None
        wc, gui = per("SQ_WAVE_CYCLES"), per("GRBM_GUI_ACTIVE")
        if not wc:
            continue
        f = lambda c: (per(c) or 0.0) / wc
        mfma = (per("SQ_VALU_MFMA_BUSY_CYCLES") or 0.0) / (1024.0 * gui / 8.0) if gui else float("nan")
        lds = (per("SQ_LDS_BANK_CONFLICT") or 0.0) / per("SQ_LDS_IDX_ACTIVE") if per("SQ_LDS_IDX_ACTIVE") else 0.0
        us = row.get("avg_duration_us_in_this_pass")
        ghz = gui / 8.0 / (us * 1e3) if us and gui else float("nan")       # shader clock the launch actually ran at (counter pass)
        print(f"{row['dispatches_profiled']:5d} {f('SQ_WAIT_ANY'):7.3f} {f('SQ_WAIT_INST_ANY'):7.3f} {f('SQ_ACTIVE_INST_ANY'):7.3f} {mfma:6.3f} {lds:6.3f} "
              f"{us if us else float('nan'):8.2f} {ghz:5.2f}  {name[:150]}")


if __name__ == "__main__":
    main()
