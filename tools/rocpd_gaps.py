"""Idle time between kernels in a rocprofv3 kernel trace (rocpd SQLite): over the last K images of a run, how much of the wall time has NO kernel
running, and which (previous kernel -> next kernel) hand-overs own that idle time.  An image = the interval between two starts of a marker kernel
that runs once per image (default: patch_im2col_kernel, the first kernel of the Swin stem).

    python tools/rocpd_gaps.py gpurun_out/prof_x/*/*_results.db [--images 6] [--marker patch_im2col] [--top 25]
"""
import collections
import sqlite3
import sys


def short(n):
    n = n.split("(")[0]
    return n[:90]


def main():
    path = sys.argv[1]
    arg = lambda k, d: (sys.argv[sys.argv.index(k) + 1] if k in sys.argv else d)
    images, marker, top = int(arg("--images", 6)), arg("--marker", "patch_im2col"), int(arg("--top", 25))
    c = sqlite3.connect(path)
    cur = c.execute("select * from kernels limit 1")
    cols = [d[0] for d in cur.description]
    s_col = "start" if "start" in cols else [x for x in cols if "start" in x][0]
    e_col = "end" if "end" in cols else [x for x in cols if "end" in x][0]
    rows = c.execute(f'select name, "{s_col}", "{e_col}" from kernels order by "{s_col}"').fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    print(f"# source: {path}   columns: {cols}")
    print(f"# {len(rows)} dispatches, {len(marks)} starts of the marker '{marker}'")
    if len(marks) < images + 1:
        images = len(marks) - 1
    lo, hi = marks[-images - 1], marks[-1]
    win = rows[lo:hi]
    t0, t1 = win[0][1], rows[hi][1]
    busy, gaps, pair = 0, [], collections.defaultdict(lambda: [0, 0])
    cur_end, last_name = win[0][1], None
    for n, s, e in win:
        if s > cur_end:
            if last_name is not None:
                g = s - cur_end
                gaps.append(g)
                p = pair[(short(last_name), short(n))]
                p[0] += g
                p[1] += 1
            busy += e - s
            cur_end, last_name = e, n
        else:
            if e > cur_end:
                busy += e - cur_end
                cur_end, last_name = e, n
    span = t1 - t0
    print(f"# last {images} images: span {span / 1e6 / images:.3f} ms per image, some kernel running {busy / 1e6 / images:.3f} ms, idle {(span - busy) / 1e6 / images:.3f} ms "
          f"({100.0 * (span - busy) / span:.1f} %), {len(win) / images:.0f} dispatches and {len(gaps) / images:.0f} idle gaps per image")
    gaps.sort()
    if gaps:
        q = lambda f: gaps[min(len(gaps) - 1, int(f * len(gaps)))] / 1e3
        print(f"# gap length: median {q(0.5):.2f} us, p90 {q(0.9):.2f}, p99 {q(0.99):.2f}, max {gaps[-1] / 1e3:.1f}; sum of kernel durations per image "
              f"{sum(e - s for _, s, e in win) / 1e6 / images:.3f} ms (> busy time when two streams overlap)")
    print(f"{'idle_us/img':>12} {'count/img':>10} {'avg_us':>8}  previous kernel -> next kernel")
    for (a, b), (g, k) in sorted(pair.items(), key=lambda kv: -kv[1][0])[:top]:
        print(f"{g / 1e3 / images:12.1f} {k / images:10.1f} {g / 1e3 / k:8.2f}  {a}  ->  {b}")


if __name__ == "__main__":
    main()
