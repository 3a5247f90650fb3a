set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops.py -m gpu -q -x -k "mha or causal" 2>&1 | tail -2
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r1j.log 2>&1; tail -1 gpurun_out/bench_r1j.log | cut -c1-200
timeout 1500 python tools/bench_configs.py --fp32 --json gpurun_out/configs_r1j.json 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tail -8
