set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm.py tests/test_ops.py -m gpu -q -x 2>&1 | tail -2
timeout 900 python -m pytest tests/test_e2e_gpu.py -m gpu -q -x 2>&1 | tail -2
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r1l.log 2>&1; tail -1 gpurun_out/bench_r1l.log | cut -c1-200
timeout 600 python tools/bench_gemm.py --p64 2>&1 | grep -E "policy|w2|fc2|proj.conv2|pd.l2" 
