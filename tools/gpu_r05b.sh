# r05b: (1) ring-depth sweep of the split-f16 GEMM on the image's mid-size shapes; (2) causal attention, buffer-descriptor fetches + plain-tile
# fast path vs the r04 kernel (same box, alternating); (3) tests of both; (4) quick bench line + the full one (parity leg).
set -x
mkdir -p gpurun_out
timeout 600 python tools/bench_gemm_x3.py --ring gpurun_out/r05b_gemm_x3_ring_sweep.json > gpurun_out/r05b_gemm_ring.log 2>&1; tail -30 gpurun_out/r05b_gemm_ring.log | cut -c1-330
timeout 300 python tools/bench_attn.py --libs psalm_amd/lib/libpsalm_hip.so,tools/experiments/_build/libpsalm_hip_r04attn.so > gpurun_out/r05b_bench_attn.jsonl 2>&1; cat gpurun_out/r05b_bench_attn.jsonl | cut -c1-400
timeout 600 python -m pytest tests/test_1_ops.py tests/test_2_gemm.py -m gpu -q -x -p no:cacheprovider -k "causal or gemm_x3" > gpurun_out/r05b_pytest_ops.log 2>&1; tail -3 gpurun_out/r05b_pytest_ops.log
timeout 900 python bench.py --breakdown gpurun_out/r05b_bench_breakdown.json > gpurun_out/r05b_bench.json 2> gpurun_out/r05b_bench.err; tail -1 gpurun_out/r05b_bench.json | cut -c1-300
python - <<'PY'
import json
b = json.loads(open("gpurun_out/r05b_bench.json").read().strip().splitlines()[-1])
print("value", b["value"], "gpu_ms", b["gpu_ms_per_step"], "host_ms", b["host_ms_per_step"])
p = b["parity_vs_cpu_oracle"]; print({k: p[k] for k in ("meets_north_star_bar", "meets_bar_plain_mean", "flips_within_margin", "meets_bar_pooled")}, [(s["flipped_mask_pixels"], s["flip_margin_rel_max"], s["mask_logit_rel_err"]) for s in p["seeds"]["per_seed"]])
d = json.load(open("gpurun_out/r05b_bench_breakdown.json"))
for k in ("psalm_causal_attention_f32_split", "psalm_window_attention_split", "psalm_gemm_x3", "psalm_gemm_x3_split", "psalm_gemm_x3_ln_split"):
    print(k, d[k])
PY
