set -x
timeout 900 python -m pytest tests/test_gemm.py tests/test_ops.py -m gpu -q -x 2>&1 | tail -2
timeout 900 python -m pytest tests/test_e2e_gpu.py -m gpu -q -x 2>&1 | tail -2
for i in 1 2; do timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r1t_$i.log 2>&1; tail -1 gpurun_out/bench_r1t_$i.log | cut -c1-160; done
