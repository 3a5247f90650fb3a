# r06m: stream-K hand-off on the phased 256 x 256 GEMM: unit tests on the hardware, micro-benchmark per slice count, then bench A/B (tuning 4 = 0 / 1 / 7)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_2_gemm.py -m gpu -q -x -p no:cacheprovider -k "streamk" > gpurun_out/r06m_pytest_streamk.log 2>&1; tail -3 gpurun_out/r06m_pytest_streamk.log
timeout 600 python tools/experiments/r06_streamk.py gpurun_out/r06m_streamk.json > gpurun_out/r06m_streamk.log 2>&1; tail -20 gpurun_out/r06m_streamk.log
B="python bench.py --no-side-modes --no-cpu-baseline --no-varied"
for t in 0 1 0 1; do
  timeout 300 $B --tuning 4=$t > gpurun_out/r06m_bench_$t.json 2> gpurun_out/r06m_bench_$t.err
  python - <<PY
import json
b = json.loads(open("gpurun_out/r06m_bench_$t.json").read().strip().splitlines()[-1])
print("streamk=$t", "value", b["value"], "gpu_ms", b["gpu_ms_per_step"], "dominant", b["roofline"]["kernel"][-40:], b["roofline"]["avg_launch_us"], "parity", b.get("parity", {}).get("meets_north_star_bar"))
PY
done
