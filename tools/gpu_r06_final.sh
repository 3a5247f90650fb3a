# Round-6 final GPU pass: profile of HEAD's kernels (trace + HBM PMC + SQ) -> counter files -> the full bench line (reads them) -> the whole gpu-marked
# suite in ONE invocation (as the driver runs it) -> smoke.  Everything lands under gpurun_out/ (copied to profiles/ by hand).
set -x
mkdir -p gpurun_out gpurun_out/verify
R=$GRAFT_REPO_ROOT
bash tools/gpu_r06_profile.sh r06
timeout 900 python bench.py --breakdown gpurun_out/r06_bench_breakdown.json > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; tail -1 gpurun_out/r06_bench.json | cut -c1-400
python - <<'PY'
import json
b = json.loads(open("gpurun_out/r06_bench.json").read().strip().splitlines()[-1])
print("value", b["value"], "gpu_ms", b["gpu_ms_per_step"], "host_ms", b["host_ms_per_step"], "graph", b["graph"])
r = b["roofline"]; print(r["kernel"], r["avg_launch_us"], r["frac"], "traffic", r["traffic"], r.get("algorithmic_rows"), {k: r.get("sq_counters", {}).get(k) for k in ("matrix_pipe_busy", "clock_GHz", "avg_launch_us_in_pass")})
print([(h["kernel"][:30], h["avg_launch_us"], h["frac"]) for h in r["hbm_bound_kernels"]])
p = b["parity_vs_cpu_oracle"]; print({k: p[k] for k in ("meets_north_star_bar", "meets_bar_plain_mean", "flips_within_margin", "meets_bar_pooled")}, [(s["flipped_mask_pixels"], s["flip_margin_rel_max"], s["mask_logit_rel_err"]) for s in p["seeds"]["per_seed"]])
v = b["other_modes"].get("varied", {}); print("varied", {k: (v[k].get("images_per_s"), v[k].get("ratio_to_fixed_shape"), v[k].get("signature_misses")) for k in ("panoptic", "referring") if k in v})
print("fp32", b["other_modes"]["fp32"].get("value"), "bf16", b["other_modes"]["bf16"].get("value"), "inflight", (b["two_in_flight"] or {}).get("images_per_s"), "cpu", b["cpu_baseline"])
PY
timeout 1800 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r06_pytest_gpu_single_invocation.txt 2>&1; tail -3 gpurun_out/r06_pytest_gpu_single_invocation.txt
timeout 300 python __graft_entry__.py --smoke > gpurun_out/verify/smoke.log 2>&1; echo "smoke rc=$?"; tail -4 gpurun_out/verify/smoke.log
