set -x
timeout 600 python -m pytest tests/test_gemm.py -m gpu -q -x 2>&1 | tail -2
timeout 600 python tools/bench_gemm.py --ring 2>&1 | grep -v amdgpu.ids | tail -62
