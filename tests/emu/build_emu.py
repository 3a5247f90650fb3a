"""TEST INFRASTRUCTURE: compile the unmodified HIP kernel sources for the HOST against the
functional HIP stand-in (tests/emu/hip/hip_runtime.h) -> tests/emu/_build/libpsalm_emu.so.
Same C ABI as libpsalm_hip.so, host pointers instead of device pointers.  Never loaded by psalm_amd."""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "psalm_amd", "csrc")
# EMU_ASAN=1: AddressSanitizer build (own directory) -- every global / LDS / stack access of the kernels checked at source level.  Run as
#   LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0 EMU_ASAN=1 \
#       python -m pytest tests -m "not gpu" -k <kernel tests>
ASAN = os.environ.get("EMU_ASAN") == "1"
OUT = os.path.join(HERE, "_build_asan" if ASAN else "_build")
LIB = os.path.join(OUT, "libpsalm_emu.so")
CXX = os.environ.get("EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
FLAGS = ["-x", "c++", "-std=c++17", "-O2", "-fPIC", "-DPSALM_EMU_BUILD", "-Wno-unknown-attributes", "-Wno-unknown-pragmas",
         "-Wno-pass-failed", "-I", HERE, "-I", os.path.join(ROOT, "include")]
LDFLAGS = []
if ASAN:
    FLAGS += ["-fsanitize=address", "-fno-omit-frame-pointer", "-g"]
    LDFLAGS += ["-fsanitize=address", "-shared-libasan"]


def build(force=False, verbose=True):
    os.makedirs(OUT, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    deps = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))) + [os.path.join(HERE, "hip", "hip_runtime.h")]
    jobs, objs = [], []
    for s in srcs:
        o = os.path.join(OUT, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or any(os.path.getmtime(d) > os.path.getmtime(o) for d in [s] + deps):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        return s, subprocess.run([CXX] + FLAGS + ["-c", s, "-o", o], capture_output=True, text=True)

    if jobs:
        with ThreadPoolExecutor(max_workers=8) as ex:
            for s, r in ex.map(cc, jobs):
                if r.returncode:
                    print(r.stderr, file=sys.stderr)
                    raise RuntimeError(f"emu compile failed: {s}")
                if verbose:
                    print(f"[emu] compiled {os.path.basename(s)}")
    if jobs or force or not os.path.exists(LIB):
        r = subprocess.run([CXX, "-shared", "-fPIC", "-o", LIB] + LDFLAGS + objs, capture_output=True, text=True)
        if r.returncode:
            print(r.stderr, file=sys.stderr)
            raise RuntimeError("emu link failed")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
