# r06k: predictor K / V front on the side stream: e2e stage / graph tests, then bench lines (this build vs the previous commit's library is not possible in one
# call -- compare with r06h / r06 final on other boxes; the in-call A/B is --no-overlap (one stream: the front back on the critical path) vs default)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_9_e2e_gpu.py -m gpu -q -x -p no:cacheprovider -k "stage_level or graph_replay or tiny_vs_oracle or config5 or config3_referring_640_batch4" > gpurun_out/r06k_pytest_e2e.log 2>&1; tail -2 gpurun_out/r06k_pytest_e2e.log
B="python bench.py --no-side-modes --no-cpu-baseline --no-varied"
for t in a b c; do
  timeout 300 $B > gpurun_out/r06k_bench_$t.json 2> gpurun_out/r06k_bench_$t.err
  python - <<PY
import json
b = json.loads(open("gpurun_out/r06k_bench_$t.json").read().strip().splitlines()[-1])
print("$t", "value", b["value"], "gpu_ms", b["gpu_ms_per_step"])
PY
done
