set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm.py -m gpu -q -x 2>&1 | tail -2
timeout 600 python -m pytest tests/test_e2e_gpu.py -m gpu -q -x -k "fp8" 2>&1 | tail -3
grep fp8 gpurun_out/parity_report.jsonl
timeout 1500 python tools/bench_configs.py --json gpurun_out/configs_r1r.json 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tail -8
