set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm.py -m gpu -q -x 2>&1 | tail -3
timeout 600 python tools/bench_gemm.py --ring 2>&1 | grep -v amdgpu.ids | tail -42
timeout 600 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids | tail -20
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r1h.log 2>&1; tail -1 gpurun_out/bench_r1h.log | cut -c1-200
