# r06n: semantic pass without its SGPR spills (bit-identical): unit tests on the hardware, then the bench line's figure for the kernel (r06 final: 344.8 us)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_1_ops.py tests/test_6_model_emu.py -m gpu -q -x -p no:cacheprovider -k "semantic or postprocess or panoptic" > gpurun_out/r06n_pytest.log 2>&1; tail -3 gpurun_out/r06n_pytest.log
B="python bench.py --no-side-modes --no-cpu-baseline --no-varied"
for t in a b; do
  timeout 300 $B > gpurun_out/r06n_bench_$t.json 2> gpurun_out/r06n_bench_$t.err
  python - <<PY
import json
b = json.loads(open("gpurun_out/r06n_bench_$t.json").read().strip().splitlines()[-1])
print("$t", "value", b["value"], "gpu_ms", b["gpu_ms_per_step"], [ (k["kernel"][:28], k["avg_launch_us"]) for k in b["roofline"]["hbm_bound_kernels"]])
PY
done
