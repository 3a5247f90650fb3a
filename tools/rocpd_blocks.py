"""Per-configuration kernel durations of a micro-benchmark from its rocprofv3 kernel trace: the dispatches whose name contains a substring, in
start order, cut into consecutive blocks of N launches (a micro-benchmark's warm-up + timed launches of ONE configuration), median per block.
Host-side timing of back-to-back launches cannot see below the ~25 us a Python launch costs; the trace can.

    python tools/rocpd_blocks.py <results.db> window_attention 33
"""
import sqlite3
import sys


def main():
    path, sub, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
    c = sqlite3.connect(path)
    cols = [d[0] for d in c.execute("select * from kernels limit 1").description]
    s_col = "start" if "start" in cols else [x for x in cols if "start" in x][0]
    rows = c.execute(f'select name, duration from kernels where name like ? order by "{s_col}"', (f"%{sub}%",)).fetchall()
    print(f"# {path}: {len(rows)} dispatches matching '{sub}', blocks of {n}")
    for b in range(0, len(rows) - n + 1, n):
        blk = rows[b:b + n]
        d = sorted(r[1] for r in blk)
        names = sorted(set(r[0].split("(")[0][-60:] for r in blk))
        print(f"{b // n:4d}  median {d[len(d) // 2] / 1e3:8.2f} us   min {d[0] / 1e3:8.2f}   {' | '.join(names)}")


if __name__ == "__main__":
    main()
