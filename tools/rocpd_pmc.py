"""Per-kernel sums of rocprofv3 --pmc counters from one or more rocpd SQLite results (ROCm 7.2 default output).

    python tools/rocpd_pmc.py <results.db> [<results.db> ...] [--json out.json] [--top N]

Prints, per kernel name: dispatches and, for every collected counter, the total and the per-dispatch mean.
FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3 derived counters); on gfx950 FETCH_SIZE under-counts wide coalesced reads
by exactly 2x (/opt/skills/guides/MI355X_MICROARCH.md §HBM) -- the correction is applied where this script reports bytes."""
import json
import sqlite3
import sys
from collections import defaultdict


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 30
    jout = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    if jout in args:
        args.remove(jout)
    if "--top" in sys.argv and str(top) in args:
        args.remove(str(top))
    data = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))      # kernel -> counter -> [dispatches, sum]
    dur = defaultdict(lambda: [0, 0.0])
    for path in args:
        c = sqlite3.connect(path)
        seen = set()
        for kname, cname, value, did, d in c.execute("select kernel_name, counter_name, value, dispatch_id, duration from counters_collection"):
            e = data[kname][cname]
            e[0] += 1
            e[1] += value
            if (path, did) not in seen:
                seen.add((path, did))
                dur[kname][0] += 1
                dur[kname][1] += d or 0
    out = {}
    order = sorted(data, key=lambda k: -dur[k][1])[:top]
    for k in order:
        row = {"dispatches_profiled": max(v[0] for v in data[k].values())}
        if dur[k][0] and dur[k][1]:
            row["avg_duration_us_in_this_pass"] = dur[k][1] / dur[k][0] * 1e-3          # rocpd durations are ns
        for cn, (n, sm) in sorted(data[k].items()):
            row[cn] = {"sum": sm, "per_dispatch": sm / n}
        f = data[k].get("FETCH_SIZE")
        w = data[k].get("WRITE_SIZE")
        if f or w:
            fb = (f[1] / f[0] * 1024 * 2) if f else None            # KiB -> bytes, x2 gfx950 wide-read correction
            wb = (w[1] / w[0] * 1024) if w else None
            row["hbm_read_bytes_per_dispatch_corrected"] = fb
            row["hbm_write_bytes_per_dispatch"] = wb
        out[k] = row
        print(k[:140])
        for cn, v in row.items():
            print(f"    {cn:44s} {v}")
    if jout:
        with open(jout, "w") as fjson:
            json.dump(out, fjson, indent=1)


if __name__ == "__main__":
    main()
