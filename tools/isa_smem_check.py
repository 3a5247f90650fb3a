"""Screen of the SHIPPED code objects (psalm_amd/lib/libpsalm_hip.so, no GPU needed) for scalar-memory loads whose immediate
offset is not dword aligned.  gfx950's scalar memory unit ignores the low two address bits, and the ROCm 7.2 compiler was seen
splitting an aligned kernarg address into an unaligned SGPR base + unaligned immediate (loop-strength-reduced index into a by-value
struct kernel argument, msda_fused8_kernel r01) -- the load then returns a neighbouring slot.  That is invisible to the host
emulator and to source-level sanitizers; this screen is the CPU-side guard (tests/test_0_isa.py).
    python tools/isa_smem_check.py [lib.so]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib, td):
    """Every gfx950 code object bundled in the library's .hip_fatbin section (one bundle per translation unit)."""
    fat = os.path.join(td, "fat.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    outs = []
    for i, st in enumerate(starts):
        part = os.path.join(td, f"bundle{i}.bin")
        with open(part, "wb") as f:
            f.write(blob[st:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = os.path.join(td, f"co{i}.o")
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}", f"--output={co}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True, text=True)
        if r.returncode == 0 and os.path.getsize(co) > 0:
            outs.append(co)
    return outs


def unaligned_smem(lib):
    """[(kernel, instruction)] for every s_load / s_buffer_load with an immediate offset that is not a multiple of 4."""
    bad, nk = [], 0
    with tempfile.TemporaryDirectory() as td:
        for co in code_objects(lib, td):
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", co], capture_output=True, text=True).stdout
            cur = None
            for ln in dis.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
                if m:
                    cur = m.group(1)
                    nk += 1
                    continue
                m = re.search(r"\bs_(?:buffer_)?load_dword\w*\s+[^,]+,\s*[^,]+,\s*(0x[0-9a-f]+|-?\d+)\b", ln)
                if m and int(m.group(1), 0) % 4:
                    bad.append((cur, ln.split("//")[0].strip()))
    return bad, nk


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "psalm_amd", "lib", "libpsalm_hip.so")
    bad, nk = unaligned_smem(lib)
    print(f"{nk} kernels scanned, {len(bad)} unaligned scalar loads")
    for k, ins in bad:
        print(k, "|", ins)
    sys.exit(1 if bad else 0)
