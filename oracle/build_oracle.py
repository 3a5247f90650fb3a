"""Compiles the C part of the oracle (test infrastructure) with gcc -> oracle/_build/.
The reference is Python + one CUDA extension that cannot be built without nvcc
(ops/setup.py:50-54 demands CUDA_HOME), so there is no oracle/_ref: "reference unbuildable here"."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build")


def build(force=False):
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(HERE, "msda_ref.c")
    libs = {}
    for name, flags in (("libmsda_ref.so", []), ("libmsda_ref_f64.so", ["-DMSDA_ACC_DOUBLE"])):
        lib = os.path.join(OUT, name)
        if force or not os.path.exists(lib) or os.path.getmtime(src) > os.path.getmtime(lib):
            subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", lib, src, "-lm"] + flags, check=True)
        libs[name] = lib
    return libs


if __name__ == "__main__":
    print(build(force=True))
