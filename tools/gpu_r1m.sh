set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_gemm.py -m gpu -q -x 2>&1 | tail -2
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r1m.log 2>&1; tail -1 gpurun_out/bench_r1m.log | cut -c1-200
