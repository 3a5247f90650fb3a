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


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"), reason="needs the ROCm llvm tools")
def test_register_budgets_of_the_hot_kernels():
    """The register allocation decides how many wavefronts a SIMD holds, and nothing else in the CPU suite sees it: r04's window-attention
    kernel was designed for 7 resident waves per CU and compiled to 321 registers = one per SIMD (rocprofv3's `vgpr` column shows half a
    wave64 kernel's allocation, which read as comfortable) -- DESIGN.md section 0 item 7a.  The budgets the kernels' launch geometry relies on,
    read from the code objects' metadata (tools/isa_resources.py)."""
    import isa_resources
    if not os.path.exists(LIB):
        from psalm_amd import build
        build.build(verbose=False)
    res = isa_resources.resources(LIB)
    assert len(res) > 100, f"only {len(res)} kernel records found in {LIB}: extraction broken?"

    def one(sub):
        hits = {k: r for k, r in res.items() if sub in k}
        assert hits, f"no kernel matches {sub}"
        return hits

    # window attention: three wavefronts per pair, K through LDS -> three blocks (9 waves) per CU need <= 168 registers and no spills;
    # the one-wavefront flavour keeps K in registers: one wave per SIMD by design, but nothing may go to scratch
    for sub in ("window_attention_f32_mfma_kernelILi32ELi12ELb0ELi3ELb1E", "window_attention_f32_mfma_kernelILi32ELi12ELb1ELi3ELb1E"):
        for k, r in one(sub).items():
            assert r["vgpr"] <= 168 and r["scratch"] == 0, (k, r)
    for k, r in one("window_attention_f32_mfma_kernelILi32ELi12ELb1ELi1ELb0E").items():
        assert r["scratch"] == 0 and r["vgpr"] <= 512, (k, r)
    # Phi's causal attention: three waves per SIMD
    for k, r in one("causal_attention_f32_splitk_kernel").items():
        assert r["vgpr"] <= 168 and r["scratch"] <= 16, (k, r)
    # split-f16 GEMMs: the 64 x 128 kernels on 32-deep slices in two stages run three blocks per CU (48 KB of LDS each); the phased
    # 256 x 256 kernels two waves per SIMD (a handful of epilogue-only spills are known)
    for k, r in one("gemm_bf16_glds_kernelIfLi64ELi128ELi2ELi2ELi2ELb0ELi32ELi0ELi2E").items():
        assert r["vgpr"] <= 168 and r["scratch"] == 0 and r["lds"] <= 49152, (k, r)
    for k, r in one("gemm_bf16_glds_kernelIfLi256ELi256ELi2ELi4ELi2ELb0ELi32ELi4ELi2E").items():
        assert r["vgpr"] <= 256 and r["scratch"] <= 32 and r["lds"] == 131072, (k, r)
    # the fused MSDeformAttn gather keeps 16 corner fetches in flight at three waves per SIMD
    for k, r in one("msda_fused8_kernelIffLi3ELi4E").items():
        assert r["vgpr"] <= 168 and r["scratch"] == 0, (k, r)
