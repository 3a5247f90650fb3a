"""CPU-side screen of the shipped gfx950 code objects (no GPU, no compute): no scalar load may carry an immediate offset that is
not dword aligned -- the r01 round-end GPU fault was exactly that (tools/isa_smem_check.py explains the mechanism).  Named test_0_*
so it runs before everything else."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
LIB = os.path.join(ROOT, "psalm_amd", "lib", "libpsalm_hip.so")


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"), reason="needs the ROCm llvm tools")
def test_no_unaligned_scalar_loads_in_shipped_code_objects():
    import isa_smem_check
    if not os.path.exists(LIB):
        from psalm_amd import build
        build.build(verbose=False)
    bad, nk = isa_smem_check.unaligned_smem(LIB)
    assert nk > 100, f"only {nk} kernels found in {LIB}: extraction broken?"
    assert not bad, "unaligned scalar-memory immediates (hardware drops the low 2 bits):\n" + "\n".join(f"{k}: {i}" for k, i in bad)
