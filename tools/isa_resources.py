"""Register / scratch / LDS budget of every kernel in the SHIPPED code objects (psalm_amd/lib/libpsalm_hip.so, no GPU needed), from the AMDGPU
metadata notes: what decides how many wavefronts a SIMD holds.  rocprofv3's `vgpr` column shows HALF a wave64 kernel's allocation -- the r04
window-attention kernel "used 164" and ran one wavefront per SIMD on 321; this prints the real numbers.
    python tools/isa_resources.py [lib.so] [name-substring]"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_smem_check import LLVM, ROOT, code_objects


def resources(lib):
    """{kernel symbol: {vgpr, agpr, sgpr, scratch, lds, max_flat_workgroup_size}} over all gfx950 code objects of the library."""
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for co in code_objects(lib, td):
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            cur = {}
            for ln in notes.split("\n"):
                m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", ln)
                if not m:
                    continue
                k, v = m.group(1), m.group(2).strip().strip("'")
                if k == "agpr_count" and "agpr" in cur:           # a new kernel record starts (agpr_count is its first key in the dump)
                    if "name" in cur:
                        out[cur["name"]] = cur
                    cur = {}
                if k in ("agpr_count", "vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size"):
                    cur[{"agpr_count": "agpr", "vgpr_count": "vgpr", "sgpr_count": "sgpr", "private_segment_fixed_size": "scratch",
                         "group_segment_fixed_size": "lds", "max_flat_workgroup_size": "wg"}[k]] = int(v)
                elif k == "name":
                    cur["name"] = v
            if "name" in cur:
                out[cur["name"]] = cur
    return out


def waves_per_simd(r):
    """Resident wavefronts per SIMD the register allocation allows (512 registers per lane, allocation granule 8, at most 8)."""
    tot = (r["vgpr"] + 7) // 8 * 8                            # .vgpr_count = arch + acc registers of the unified file
    return max(1, min(8, 512 // max(tot, 1)))


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else os.path.join(ROOT, "psalm_amd", "lib", "libpsalm_hip.so")
    sub = sys.argv[-1] if len(sys.argv) > 1 and not sys.argv[-1].endswith(".so") else ""
    res = resources(lib)
    print(f"{len(res)} kernels")
    for k, r in sorted(res.items(), key=lambda kv: -kv[1].get("vgpr", 0)):
        if sub in k:
            print(f"{r.get('vgpr', 0):4d} regs ({r.get('agpr', 0):3d} acc) {waves_per_simd(r)} waves/SIMD  scratch {r.get('scratch', 0):5d}  lds {r.get('lds', 0):7d}  {k[:110]}")
