# r03 call 8: paired split-f16 stores (GEMM epilogue straight from the accumulators) -- tests, same-box A/B, bench
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_2_gemm.py -m gpu -x -q > gpurun_out/r03j_pytest_gemm.log 2>&1; echo "pytest gemm exit $?"; tail -3 gpurun_out/r03j_pytest_gemm.log
timeout 600 python tools/exp_modes.py 1024 f16x3+overlap,f16x3+overlap+nopair,f16x3+overlap,f16x3+overlap+nopair > gpurun_out/r03j_paired_ab.log 2>&1; tail -6 gpurun_out/r03j_paired_ab.log | cut -c1-600
timeout 600 python bench.py --breakdown gpurun_out/r03j_breakdown.json > gpurun_out/r03j_bench.json 2> gpurun_out/r03j_bench.err; tail -1 gpurun_out/r03j_bench.json | cut -c1-400
python - <<'PY'
import json
b=json.load(open('gpurun_out/r03j_breakdown.json'))
for k,v in list(b.items())[:8]:
    if not k.startswith('_'): print(f"{k:40s} {v['launches_per_step']:7.1f} {v['ms_per_step']:8.3f}")
for k,v in list(b['_gemm_shapes'].items())[:8]: print(f"{v['ms_per_step']:7.3f} {v['launches_per_step']:5.1f} {v['TFLOPs']:6.1f}  {k[:150]}")
r=json.loads(open('gpurun_out/r03j_bench.json').read().strip().split('\n')[-1])
print(r['parity_vs_cpu_oracle']['seeds']['mask_iou_mean_min'], r['parity_vs_cpu_oracle']['meets_north_star_bar'])
PY
