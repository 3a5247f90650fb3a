"""Builds psalm_amd/lib/libpsalm_hip.so from psalm_amd/csrc/*.hip with hipcc for gfx950.

    python -m psalm_amd.build [--force]

hipcc cross-compiles without a GPU, so this runs in the authoring container; the built .so is
git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libpsalm_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-I", os.path.join(os.path.dirname(HERE), "include")]


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(os.path.dirname(HERE), "include", "*.h")))
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _newer([s] + hdrs, o):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return s, r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for s, r in ex.map(cc, jobs):
                if verbose and (r.stderr.strip() or r.returncode):
                    print(r.stderr, file=sys.stderr)
                if r.returncode:
                    raise RuntimeError(f"hipcc failed on {s}")
                if verbose:
                    print(f"[psalm_amd.build] compiled {os.path.basename(s)}")
    if force or jobs or not os.path.exists(LIB):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, capture_output=True, text=True)
        if r.returncode:
            print(r.stderr, file=sys.stderr)
            raise RuntimeError("link failed")
        if verbose:
            print(f"[psalm_amd.build] linked {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
