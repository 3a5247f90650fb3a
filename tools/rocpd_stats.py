"""Summarise a rocprofv3 (ROCm 7.2) rocpd SQLite result the way `--stats` CSVs do: per-kernel calls / total / average /
min / max / share.  `rocprofv3 --kernel-trace --stats` writes <dir>/<host>/<pid>_results.db on this image; this turns it
into the text summary committed under profiles/.

    python tools/rocpd_stats.py gpurun_out/prof_x/*/*_results.db > profiles/r01_x_kernel_stats.txt
"""
import sqlite3
import sys


def main(path, top=60):
    c = sqlite3.connect(path)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
                     "max(accum_vgpr_count), max(sgpr_count), max(lds_size) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f"# source: {path}")
    print(f"# total kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    print("# med_us = median launch duration: what bench.py's HIP-event figures aggregate by (its instrumented steps count a (kernel, shape)'s launches with their")
    print("#          median, so that one stalled eager launch does not move a line); avg_us is the mean over ALL launches of the process, cold first ones included")
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'med_us':>9} {'min_us':>9} {'max_us':>9} {'pct':>6} {'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'lds':>7}  name")
    for n, k, s, a, mn, mx, vg, ag, sg, lds in rows[:top]:
        d = [r[0] for r in c.execute("select duration from kernels where name = ? order by duration", (n,)).fetchall()]
        med = d[len(d) // 2] if d else 0
        print(f"{k:7d} {s / 1e6:10.3f} {a / 1e3:10.2f} {med / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100 * s / tot:6.2f} {vg or 0:5d} {ag or 0:5d} {sg or 0:5d} {lds or 0:7d}  {n[:150]}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60)
