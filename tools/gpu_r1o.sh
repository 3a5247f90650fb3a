set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_e2e_gpu.py -m gpu -q 2>&1 | tail -12
for i in 1 2 3; do timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r1o_$i.log 2>&1; tail -1 gpurun_out/bench_r1o_$i.log | cut -c1-160; done
