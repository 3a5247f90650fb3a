# r06s: causal attention with the two query tiles of a block side by side (PSALM_TUNE_ATTN_DUO): unit tests, bench A/B, kernel trace
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_1_ops.py -m gpu -q -x -p no:cacheprovider -k "causal or two_tiles" > gpurun_out/r06s_pytest.log 2>&1; tail -2 gpurun_out/r06s_pytest.log
B="python bench.py --no-side-modes --no-cpu-baseline --no-varied"
for t in 0 1 0 1; do
  timeout 300 $B --tuning 5=$t > gpurun_out/r06s_bench_$t.json 2> gpurun_out/r06s_bench_$t.err
  python - <<PY
import json
b = json.loads(open("gpurun_out/r06s_bench_$t.json").read().strip().splitlines()[-1])
p = b.get("parity_vs_cpu_oracle", {})
print("duo=$t", "value", b["value"], "gpu_ms", b["gpu_ms_per_step"], [s["flipped_mask_pixels"] for s in p.get("seeds", {}).get("per_seed", [])])
PY
done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-side-modes --no-varied --eager --no-overlap > $R/gpurun_out/r06s_prof_kt.log 2>&1
cd $R
python tools/rocpd_stats.py gpurun_out/prof_kt/*/*_results.db 90 > gpurun_out/r06s_kernel_stats.txt
rm -rf gpurun_out/prof_kt
grep -E "causal_attention" gpurun_out/r06s_kernel_stats.txt | cut -c1-170
