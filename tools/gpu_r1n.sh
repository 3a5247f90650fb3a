set -x
mkdir -p gpurun_out
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r1n.log 2>&1; tail -1 gpurun_out/bench_r1n.log | cut -c1-200
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-prefetch > gpurun_out/bench_r1n_nopf.log 2>&1; tail -1 gpurun_out/bench_r1n_nopf.log | cut -c1-200
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r1n2.log 2>&1; tail -1 gpurun_out/bench_r1n2.log | cut -c1-200
timeout 600 python -m pytest tests/test_e2e_gpu.py -m gpu -q -x -k "graph or instance" 2>&1 | tail -2
