# r05q: window attention -- r05's two flavours against the kernel they replace (tools/experiments/build_side_lib.sh r05head_winattn f7a93b9 attention), by
# KERNEL TRACE; the op tests and the e2e parity tests on hardware; a quick bench line
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_1_ops.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r05q_pytest_ops.log 2>&1; tail -2 gpurun_out/r05q_pytest_ops.log
LIBS=tools/experiments/_build/libpsalm_hip_r05head_winattn.so,psalm_amd/lib/libpsalm_hip.so
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_wa -- python $R/tools/bench_winattn.py --libs $LIBS > $R/gpurun_out/r05q_bench_winattn.jsonl 2>&1
cd $R
python tools/rocpd_blocks.py gpurun_out/prof_wa/*/*_results.db window_attention_f32_mfma 33 > gpurun_out/r05q_winattn_trace_blocks.txt 2>&1
rm -rf gpurun_out/prof_wa
python - <<'PY'
import re, json
rows = [l for l in open("gpurun_out/r05q_winattn_trace_blocks.txt") if re.match(r"\s*\d+\s+median", l)]
libs = ["r04 kernel", "r05 flavours"]
cfg = ["s1/0", "s1/6", "s2/0", "s2/6", "s3/0", "s3/6", "s4/0", "s4/6"]
for i in range(0, len(rows), 8):
    blk = rows[i:i + 8]
    print(i // 8 // len(libs), f"{libs[(i // 8) % len(libs)]:14s}", " ".join(f"{c}:{float(r.split()[2]):6.1f}" for c, r in zip(cfg, blk)))
cs = [json.loads(l) for l in open("gpurun_out/r05q_bench_winattn.jsonl") if l.startswith("{")]
print("checksums equal:", all(all(abs(c[k]["checksum"] - cs[0][k]["checksum"]) == 0 for k in c if k.startswith("stage")) for c in cs))
PY
timeout 900 python -m pytest tests/test_9_e2e_gpu.py -m gpu -q -x -p no:cacheprovider -k "golden or tiny_vs_oracle or graph_replay or stage_level or config2_panoptic_1024" > gpurun_out/r05q_pytest_e2e.log 2>&1; tail -2 gpurun_out/r05q_pytest_e2e.log
timeout 300 python bench.py --no-side-modes --no-cpu-baseline --no-varied --breakdown gpurun_out/r05q_bench_breakdown.json > gpurun_out/r05q_bench_quick.json 2> gpurun_out/r05q_bench_quick.err; tail -1 gpurun_out/r05q_bench_quick.json | cut -c1-260
python - <<'PY'
import json
b = json.loads(open("gpurun_out/r05q_bench_quick.json").read().strip().splitlines()[-1])
p = b["parity_vs_cpu_oracle"]; print("value", b["value"], "gpu_ms", b["gpu_ms_per_step"], {k: p[k] for k in ("meets_north_star_bar", "flips_within_margin")}, [(s["flipped_mask_pixels"], s["mask_logit_rel_err"]) for s in p["seeds"]["per_seed"]])
d = json.load(open("gpurun_out/r05q_bench_breakdown.json")); print({k: round(v["ms_per_step"], 3) for k, v in d.items() if "window_attention" in k})
PY
