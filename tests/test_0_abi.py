"""The C-ABI library loads and exports every entry point include/psalm_hip.h declares (no compute calls: CPU only),
and the product accessor refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "psalm_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(psalm_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_whole_surface():
    names = _declared()
    assert "psalm_msda_forward" in names and "psalm_gemm" in names and len(names) >= 30


def test_library_exports_every_declared_symbol():
    import torch  # noqa: F401  (torch's libamdhip64 must be loaded first, see __graft_entry__.build)
    from psalm_amd import build as hip_build
    lib = hip_build.build(verbose=False)
    so = ctypes.CDLL(lib)
    missing = [n for n in _declared() if not hasattr(so, n)]
    assert not missing, f"declared in include/psalm_hip.h but not exported by {lib}: {missing}"
    so.psalm_backend.restype = ctypes.c_char_p
    assert so.psalm_backend() == b"hip-gfx950"
    # the built library, the header constant and the binding's constant are one number (a stale library under a newer binding
    # would take integers for pointers)
    import re
    from psalm_amd import hip_ops as H
    hdr = int(re.search(r"#define\s+PSALM_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))
    assert so.psalm_abi_version() == hdr == H.ABI_VERSION


def test_binding_refuses_a_library_of_another_abi_version(tmp_path, monkeypatch):
    """Ops.__init__ compares psalm_abi_version() with the version it was written against BEFORE any other call."""
    from psalm_amd import build as hip_build
    from psalm_amd import hip_ops as H
    lib = hip_build.build(verbose=False)
    monkeypatch.setattr(H, "ABI_VERSION", H.ABI_VERSION + 1)
    with pytest.raises(H.PsalmHipError, match="psalm_abi_version"):
        H.Ops(lib)


def test_every_exported_entry_point_is_declared():
    """The reverse direction: nothing callable from Python that the header does not document."""
    import subprocess
    from psalm_amd import build as hip_build
    lib = hip_build.build(verbose=False)
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout
    exported = sorted(set(re.findall(r" T (psalm_[a-z0-9_]+)$", out, flags=re.M)))
    undeclared = [n for n in exported if n not in _declared() and n != "psalm_set_error"]
    assert not undeclared, f"exported but missing from include/psalm_hip.h: {undeclared}"


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from psalm_amd import hip_ops
    with pytest.raises(hip_ops.PsalmHipError):
        hip_ops.get_ops()
    with pytest.raises(hip_ops.PsalmHipError):
        hip_ops.Ops("/nonexistent/libpsalm_hip.so")
