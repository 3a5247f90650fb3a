import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Test files are numbered so that a plain `pytest tests -x` meets a broken kernel in its own small test first:
#   0 shipped-ISA screen, C-ABI exports | 1 row / attention / image / post-processing ops | 2 GEMM family | 3 MSDeformAttn
#   4 oracle vs reference-generated goldens | 5 multi-process (gloo) | 6 whole tiny model on the host emulator | 7 builder / drop-in API
#   8 evaluator-facing outputs, pre-processing | 9 whole model on the GPU (goldens, BASELINE.json configs)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: multi-second CPU test")
    # The CPU oracle is the long pole of the GPU suite (tests/test_9: one 1024^2 image = one full fp32 model evaluation on the host), and torch's
    # default of one thread per hardware thread OVERSUBSCRIBES it on the GPU box: profiles/r04_cpu_baseline_threads.json, 8 / 16 / 32 / 64 threads
    # = 9.0 / 6.8 / 7.5 / 13.9 s per image.  Same results (the tests' bars are tolerances or integer statistics), half the wall time.
    try:
        import torch
        torch.set_num_threads(min(16, os.cpu_count() or 1))
    except Exception:
        pass


def pytest_collection_modifyitems(config, items):
    """`gpu` tests are skipped (not failed) on a machine without a GPU or without the built HIP library, so a plain `pytest tests`
    works everywhere; on the GPU box nothing is skipped."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    have_lib = os.path.exists(os.path.join(ROOT, "psalm_amd", "lib", "libpsalm_hip.so"))
    if have_gpu and have_lib:
        return
    why = "no GPU visible" if not have_gpu else "psalm_amd/lib/libpsalm_hip.so not built"
    skip = pytest.mark.skip(reason=f"gpu test: {why}")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
