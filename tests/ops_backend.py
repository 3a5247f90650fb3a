"""`ops` fixture: the same op-level tests run against
  * "emu": the HIP kernel sources compiled for the host (tests/emu) -- CPU, small sizes, not a GPU test;
  * "hip": the real libpsalm_hip.so on an MI355X (marked gpu).
"""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))

_CACHE = {}


def make_ops(kind):
    from psalm_amd.hip_ops import Ops, get_ops
    if kind not in _CACHE:
        if kind == "emu":
            import build_emu
            _CACHE[kind] = Ops(build_emu.build(verbose=False))
        else:
            _CACHE[kind] = get_ops()
    return _CACHE[kind]


@pytest.fixture(params=["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def ops(request):
    if request.param == "hip" and not torch.cuda.is_available():
        pytest.skip("no GPU")
    return make_ops(request.param)
