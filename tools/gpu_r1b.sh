set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm.py tests/test_ops.py -m gpu -q -x 2>&1 | tail -6
timeout 300 python tools/bench_gemm.py --json gpurun_out/gemm_r1b.json 2>&1 | tail -22
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --breakdown gpurun_out/breakdown_r1b.json > gpurun_out/bench_r1b.log 2>&1; tail -2 gpurun_out/bench_r1b.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/breakdown_r1b.json'))
for k,v in list(d.items())[:14]: print(f"{k:34s} {v['launches_per_step']:7.1f} {v['ms_per_step']:8.3f}")
PY
timeout 600 python -m pytest tests/test_e2e_gpu.py -m gpu -q -x 2>&1 | tail -5
