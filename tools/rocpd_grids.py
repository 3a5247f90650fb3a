"""Per (kernel, grid) rows of a rocprofv3 rocpd result: launches, median duration, blocks, threads per block -- to find launches whose SHAPE leaves the chip
idle (few blocks x long duration) or under-fills its wavefronts.   python tools/rocpd_grids.py <results.db> [min_total_us]"""
import sqlite3
import sys


def main(path, min_total_us=30.0):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
    gx = [k for k in ("grid_x", "grid_size_x") if k in cols]
    wx = [k for k in ("workgroup_x", "workgroup_size_x") if k in cols]
    if not gx or not wx:
        print("columns:", cols)
        return
    g, w = gx[0][:-1], wx[0][:-1]
    q = (f"select name, {g}x, {g}y, {g}z, {w}x, {w}y, {w}z, count(*), sum(duration), min(duration), max(duration) from kernels "
         f"group by name, {g}x, {g}y, {g}z, {w}x order by 9 desc")
    print(f"{'calls':>6} {'total_us':>9} {'avg_us':>8} {'blocks':>8} {'thr':>5} {'blk/CU':>7}  name")
    for n, gx_, gy, gz, wx_, wy, wz, k, s, mn, mx in c.execute(q).fetchall():
        if s / 1e3 < min_total_us:
            continue
        thr = wx_ * wy * wz
        total = gx_ * gy * gz
        blocks = total // thr if total % thr == 0 and total >= thr else total      # (rocprof reports grid in work-items)
        print(f"{k:6d} {s / 1e3:9.1f} {s / k / 1e3:8.2f} {blocks:8d} {thr:5d} {blocks / 256:7.2f}  {n[:110]}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 30.0)
