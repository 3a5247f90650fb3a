# r06q: block-per-row im2col_split for the projector's few long rows; W-streaming GEMM shapes of the projector per tile policy; bench
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_2_gemm.py -m gpu -q -x -p no:cacheprovider -k "im2col" > gpurun_out/r06q_pytest.log 2>&1; tail -2 gpurun_out/r06q_pytest.log
timeout 600 python tools/bench_gemm_x3.py --ring --shapes "256,2048,18432;256,2048,9216;256,2048,1024" gpurun_out/r06q_gemm_projector.json 2>&1 | tail -4
B="python bench.py --no-side-modes --no-cpu-baseline --no-varied"
for t in a b; do
  timeout 300 $B > gpurun_out/r06q_bench_$t.json 2> gpurun_out/r06q_bench_$t.err
  python - <<PY
import json
b = json.loads(open("gpurun_out/r06q_bench_$t.json").read().strip().splitlines()[-1])
print("$t", "value", b["value"], "gpu_ms", b["gpu_ms_per_step"])
PY
done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-side-modes --no-varied --eager --no-overlap > $R/gpurun_out/r06q_prof_kt.log 2>&1
cd $R
python tools/rocpd_stats.py gpurun_out/prof_kt/*/*_results.db 90 > gpurun_out/r06q_kernel_stats.txt
rm -rf gpurun_out/prof_kt
grep -E "im2col" gpurun_out/r06q_kernel_stats.txt | cut -c1-170
