# r06r: register-resident patch_merge_ln: unit tests, bench A/B on the row-group switch, kernel trace
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_1_ops.py -m gpu -q -x -p no:cacheprovider -k "patch_merge or several_rows" > gpurun_out/r06r_pytest.log 2>&1; tail -2 gpurun_out/r06r_pytest.log
B="python bench.py --no-side-modes --no-cpu-baseline --no-varied"
for t in 0 1 0 1; do
  timeout 300 $B --tuning 4=$t > gpurun_out/r06r_bench_$t.json 2> gpurun_out/r06r_bench_$t.err
  python - <<PY
import json
b = json.loads(open("gpurun_out/r06r_bench_$t.json").read().strip().splitlines()[-1])
print("row_groups=$t", "value", b["value"], "gpu_ms", b["gpu_ms_per_step"])
PY
done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-side-modes --no-varied --eager --no-overlap > $R/gpurun_out/r06r_prof_kt.log 2>&1
cd $R
python tools/rocpd_stats.py gpurun_out/prof_kt/*/*_results.db 90 > gpurun_out/r06r_kernel_stats.txt
rm -rf gpurun_out/prof_kt
grep -E "patch_merge|im2col" gpurun_out/r06r_kernel_stats.txt | cut -c1-170
