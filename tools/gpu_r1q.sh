set -x
timeout 600 python -m pytest tests/test_gemm.py tests/test_ops.py -m gpu -q -x 2>&1 | tail -2
timeout 600 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids | tail -20
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r1q.log 2>&1; tail -1 gpurun_out/bench_r1q.log | cut -c1-200
