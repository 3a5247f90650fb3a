# r05a: (1) profile pass of HEAD's kernels; (2) the full bench line (gate v4, varied streams, counter files of (1)); (3) A/B tail-in-graph;
# (4) the GPU tests this round's host-side changes touch.
set -x
mkdir -p gpurun_out
bash tools/gpu_r05_profile.sh r05a
timeout 900 python bench.py --breakdown gpurun_out/r05a_bench_breakdown.json > gpurun_out/r05a_bench.json 2> gpurun_out/r05a_bench.err; tail -1 gpurun_out/r05a_bench.json | cut -c1-600
python - <<'PY'
import json
b = json.loads(open("gpurun_out/r05a_bench.json").read().strip().splitlines()[-1])
print("value", b["value"], "gpu_ms", b["gpu_ms_per_step"], "host_ms", b["host_ms_per_step"], "graph", b["graph"])
r = b["roofline"]; print(r["kernel"], r["avg_launch_us"], r["frac"], "traffic", r["traffic"], "sq" , {k: r.get("sq_counters", {}).get(k) for k in ("matrix_pipe_busy", "clock_GHz", "kernel_in_the_pass")})
p = b["parity_vs_cpu_oracle"]; print({k: p[k] for k in ("meets_north_star_bar", "meets_bar_plain_mean", "flips_within_margin", "meets_bar_pooled")}, [(s["flipped_mask_pixels"], s["flip_margin_rel_max"], s["mask_logit_rel_err"]) for s in p["seeds"]["per_seed"]])
print("varied", json.dumps(b["other_modes"].get("varied"))[:1500])
print("fp32", b["other_modes"]["fp32"].get("value"), "bf16", b["other_modes"]["bf16"].get("value"), "inflight", b["two_in_flight"])
PY
timeout 300 python bench.py --no-side-modes --no-cpu-baseline --no-varied --graph-tail > gpurun_out/r05a_bench_graph_tail.json 2> gpurun_out/r05a_bench_graph_tail.err; tail -1 gpurun_out/r05a_bench_graph_tail.json | cut -c1-330
timeout 300 python bench.py --no-side-modes --no-cpu-baseline --no-varied > gpurun_out/r05a_bench_quick.json 2> gpurun_out/r05a_bench_quick.err; tail -1 gpurun_out/r05a_bench_quick.json | cut -c1-330
timeout 900 python -m pytest tests/test_9_e2e_gpu.py -m gpu -q -p no:cacheprovider -k "graph_replay or padded_box or seed11 or multi_seed or config2_panoptic_1024_f16x3" --durations=8 > gpurun_out/r05a_pytest_e2e.log 2>&1; tail -16 gpurun_out/r05a_pytest_e2e.log
timeout 300 python -m pytest tests/test_7_dropin.py tests/test_0_abi.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r05a_pytest_dropin.log 2>&1; tail -3 gpurun_out/r05a_pytest_dropin.log
