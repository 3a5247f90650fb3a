# r03 call 7: two-group semantic kernel + slice-form (32-deep, 2 stages) K loop as the automatic choice (incl. split-f16 output)
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_1_ops.py tests/test_2_gemm.py -m gpu -x -q > gpurun_out/r03i_pytest_ops.log 2>&1; echo "pytest ops exit $?"; tail -3 gpurun_out/r03i_pytest_ops.log
timeout 900 python -m pytest tests/test_9_e2e_gpu.py -m gpu -x -q -k "config2 or golden_panoptic" > gpurun_out/r03i_pytest_e2e.log 2>&1; echo "pytest e2e exit $?"; tail -3 gpurun_out/r03i_pytest_e2e.log
cp -f gpurun_out/parity_report.jsonl gpurun_out/r03i_parity_report.jsonl 2>/dev/null
timeout 600 python bench.py --breakdown gpurun_out/r03i_breakdown.json > gpurun_out/r03i_bench.json 2> gpurun_out/r03i_bench.err; tail -1 gpurun_out/r03i_bench.json | cut -c1-400
python - <<'PY'
import json
b=json.load(open('gpurun_out/r03i_breakdown.json'))
for k,v in list(b.items())[:12]:
    if not k.startswith('_'): print(f"{k:40s} {v['launches_per_step']:7.1f} {v['ms_per_step']:8.3f}")
r=json.loads(open('gpurun_out/r03i_bench.json').read().strip().split('\n')[-1])
print(r['roofline']['hbm_bound_kernels']); print(r['parity_vs_cpu_oracle']['seeds']['mask_iou_mean_min'], r['parity_vs_cpu_oracle']['meets_north_star_bar'])
PY
