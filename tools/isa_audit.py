"""Static audit of the gfx950 ISA of every kernel (no GPU needed): per kernel the number of global loads, of `s_waitcnt vmcnt(0)`
(a load whose result is waited for alone = one dependent L2 / HBM round trip), of counted waits, exec-mask branches, scratch
(spill) instructions, VGPRs and occupancy.  A kernel whose vmcnt(0) count is close to its load count fetches one element at a time
(guarded loads: `cond ? p[i] : 0` compiles to a branch + wait per load) -- see DESIGN.md "ISA audit".
    python tools/isa_audit.py [--all]        (default: only kernels with >= 8 loads and vmcnt(0) >= loads / 2, or with scratch)"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "psalm_amd", "csrc")


def audit(path, show_all):
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only", path, "-o", asm, "-I", os.path.join(ROOT, "include"),
                            "-I", CSRC, "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
        res, cur = {}, None
        for ln in r.stderr.split("\n"):
            m = re.search(r"Function Name: (\S+)", ln)
            if m:
                cur = res.setdefault(m.group(1), {})
            for key, pat in (("vgpr", r"VGPRs: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)")):
                m = re.search(pat, ln)
                if m and cur is not None and " " + key not in cur:
                    cur.setdefault(key, int(m.group(1)))
        name, stats = None, {}
        for ln in open(asm):
            m = re.match(r"^(_Z\w+):", ln)
            if m:
                name = m.group(1)
                stats[name] = dict(loads=0, w0=0, wn=0, execz=0, spill=0)
                continue
            if name is None:
                continue
            st = stats[name]
            if re.search(r"\bglobal_load_(dword|ushort|short|ubyte|sbyte|dwordx2|dwordx3|dwordx4)\b", ln) and "lds" not in ln:
                st["loads"] += 1
            if "s_waitcnt vmcnt(0)" in ln:
                st["w0"] += 1
            elif "s_waitcnt vmcnt" in ln:
                st["wn"] += 1
            if "s_cbranch_execz" in ln:
                st["execz"] += 1
            if "scratch_" in ln:
                st["spill"] += 1
            if "s_endpgm" in ln:
                name = None
        rows = []
        for k, st in stats.items():
            flagged = (st["loads"] >= 8 and st["w0"] >= 0.5 * st["loads"]) or st["spill"] > 0
            if show_all or flagged:
                r_ = res.get(k, {})
                try:
                    short = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip() or k
                except OSError:
                    short = k
                rows.append((os.path.basename(path), short[:96], st["loads"], st["w0"], st["wn"], st["execz"], st["spill"], r_.get("vgpr"), r_.get("occ")))
        return rows


def main():
    show_all = "--all" in sys.argv
    print(f"{'file':18s} {'loads':>5s} {'vm(0)':>5s} {'vm(N)':>5s} {'execz':>5s} {'spill':>5s} {'vgpr':>4s} {'occ':>3s}  kernel")
    for path in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        for f, k, lo, w0, wn, ex, sp, vg, oc in audit(path, show_all):
            print(f"{f:18s} {lo:5d} {w0:5d} {wn:5d} {ex:5d} {sp:5d} {str(vg):>4s} {str(oc):>3s}  {k}")


if __name__ == "__main__":
    main()
