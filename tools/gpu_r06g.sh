# r06g: decoder launch fusion on hardware: op tests, the full-size stage-level bitwise test, quick bench lines with PSALM_TUNE_DECODER_FUSE off / on
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_1_ops.py -m gpu -q -x -p no:cacheprovider -k "chain or pair" > gpurun_out/r06g_pytest_ops.log 2>&1; tail -2 gpurun_out/r06g_pytest_ops.log
timeout 900 python -m pytest tests/test_9_e2e_gpu.py -m gpu -q -x -p no:cacheprovider -k "stage_level or graph_replay or tiny_vs_oracle" > gpurun_out/r06g_pytest_e2e.log 2>&1; tail -2 gpurun_out/r06g_pytest_e2e.log
B="python bench.py --no-side-modes --no-cpu-baseline --no-varied"
for t in off on off2 on2; do
  case $t in off*) TUNE="--tuning 3=0";; *) TUNE="";; esac
  timeout 300 $B $TUNE > gpurun_out/r06g_bench_$t.json 2> gpurun_out/r06g_bench_$t.err
  python - <<PY
import json
b = json.loads(open("gpurun_out/r06g_bench_$t.json").read().strip().splitlines()[-1])
print("$t", "value", b["value"], "gpu_ms", b["gpu_ms_per_step"])
PY
done
