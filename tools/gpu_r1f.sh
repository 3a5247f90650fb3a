set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm.py tests/test_ops.py -m gpu -q -x 2>&1 | tail -3
timeout 900 python -m pytest tests/test_e2e_gpu.py -m gpu -q -x 2>&1 | tail -3
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --breakdown gpurun_out/breakdown_r1f.json > gpurun_out/bench_r1f.log 2>&1; tail -1 gpurun_out/bench_r1f.log | cut -c1-250
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --eager > gpurun_out/bench_r1f_eager.log 2>&1; tail -1 gpurun_out/bench_r1f_eager.log | cut -c1-250
python - <<'PY'
import json
d=json.load(open('gpurun_out/breakdown_r1f.json'))
g=d.pop('_gemm_shapes')
print('total', sum(v['ms_per_step'] for v in d.values()), sum(v['launches_per_step'] for v in d.values()))
for k,v in list(d.items())[:22]: print(f"{k:34s} {v['launches_per_step']:7.1f} {v['ms_per_step']:8.3f}")
for k,v in list(g.items())[:12]: print(f"{k:46s} {v['launches_per_step']:6.1f} {v['ms_per_step']:8.3f} {v['TFLOPs']}")
PY
