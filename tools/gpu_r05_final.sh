# Round-5 final GPU pass: profile of HEAD's kernels (trace + HBM PMC + SQ) -> counter files -> the full bench line (reads them) -> kernel A/Bs
# against the r04 attention build -> where the copyBuffer / ATen rows of the trace come from -> the whole gpu-marked suite file by file -> smoke
# -> the wide parity run under gate version 4.  Everything lands under gpurun_out/ (copied to profiles/ by hand).
set -x
mkdir -p gpurun_out gpurun_out/verify
R=$GRAFT_REPO_ROOT
bash tools/gpu_r05_profile.sh r05
timeout 900 python bench.py --breakdown gpurun_out/r05_bench_breakdown.json > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err; tail -1 gpurun_out/r05_bench.json | cut -c1-400
python - <<'PY'
import json
b = json.loads(open("gpurun_out/r05_bench.json").read().strip().splitlines()[-1])
print("value", b["value"], "gpu_ms", b["gpu_ms_per_step"], "host_ms", b["host_ms_per_step"], "graph", b["graph"])
r = b["roofline"]; print(r["kernel"], r["avg_launch_us"], r["frac"], "traffic", r["traffic"], r.get("algorithmic_rows"), {k: r.get("sq_counters", {}).get(k) for k in ("matrix_pipe_busy", "clock_GHz", "avg_launch_us_in_pass")})
print([(h["kernel"][:30], h["avg_launch_us"], h["frac"]) for h in r["hbm_bound_kernels"]])
p = b["parity_vs_cpu_oracle"]; print({k: p[k] for k in ("meets_north_star_bar", "meets_bar_plain_mean", "flips_within_margin", "meets_bar_pooled")}, [(s["flipped_mask_pixels"], s["flip_margin_rel_max"], s["mask_logit_rel_err"]) for s in p["seeds"]["per_seed"]])
v = b["other_modes"].get("varied", {}); print("varied", {k: (v[k].get("images_per_s"), v[k].get("ratio_to_fixed_shape"), v[k].get("signature_misses")) for k in ("panoptic", "referring") if k in v})
print("fp32", b["other_modes"]["fp32"].get("value"), "bf16", b["other_modes"]["bf16"].get("value"), "inflight", (b["two_in_flight"] or {}).get("images_per_s"), "cpu", b["cpu_baseline"]["value"])
PY
timeout 200 python tools/bench_attn.py --libs psalm_amd/lib/libpsalm_hip.so,tools/experiments/_build/libpsalm_hip_r04attn.so > gpurun_out/r05_bench_attn.jsonl 2>&1; cut -c1-330 gpurun_out/r05_bench_attn.jsonl
timeout 200 python tools/bench_mha.py --libs psalm_amd/lib/libpsalm_hip.so,tools/experiments/_build/libpsalm_hip_r04attn.so > gpurun_out/r05_bench_mha.jsonl 2>&1; cut -c1-400 gpurun_out/r05_bench_mha.jsonl
# host time to ISSUE one image's launches (no graph): stage-level native calls vs one ctypes call per launch; and the eager images/s of both
python - > gpurun_out/r05_eager_host_issue.json <<'PY'
import json, sys, time, torch
sys.path.insert(0, ".")
from psalm_amd.config import PsalmConfig
from psalm_amd.model import PSALM
from psalm_amd.synthetic import make_inputs, make_state_dict
cfg = PsalmConfig(seg_task="panoptic"); sd = make_state_dict(cfg, seed=0)
m = PSALM(cfg, sd, precision="f16x3", use_graphs=False)
inp = make_inputs(cfg, "panoptic", size=1024, batch=1, seed=0); inp["images"] = inp["images"].cuda()
out = {}
for flag in (True, False, True, False):
    m.c_stages = flag
    for _ in range(3): m.eval_seg(**inp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m.eval_seg(**inp)
    torch.cuda.synchronize(); full = (time.perf_counter() - t0) * 100
    m._finalize = lambda r_, i_: r_
    issue = []
    for _ in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter(); m.eval_seg(**inp); issue.append((time.perf_counter() - t0) * 1e3)
    del m._finalize
    torch.cuda.synchronize()
    out.setdefault("c_stages=%s" % flag, []).append({"eager_ms_per_image": round(full, 2), "host_issue_ms_per_image_median": round(sorted(issue)[5], 2)})
print(json.dumps(out))
PY
cat gpurun_out/r05_eager_host_issue.json
# the copyBuffer / ATen rows of the kernel trace: per image, or the one-off weight preparation?  Same command, 2 vs 8 timed steps (6 vs 12 images)
cd /tmp && export TMPDIR=/tmp
for ST in 2 8; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_cb$ST -- python $R/bench.py --steps $ST --warmup 0 --no-cpu-baseline --no-side-modes --no-varied --eager --no-overlap > $R/gpurun_out/r05_prof_cb$ST.log 2>&1
done
cd $R
for ST in 2 8; do python tools/rocpd_stats.py gpurun_out/prof_cb$ST/*/*_results.db 90 > gpurun_out/r05_kernel_stats_steps$ST.txt; done
python - <<'PY'
import re
def rows(p):
    out = {}
    for l in open(p):
        m = re.match(r"\s*(\d+)\s+[\d.]+\s+[\d.]+\s+[\d.]+\s+[\d.]+\s+[\d.]+\s+[\d.]+\s+\d+\s+\d+\s+\d+\s+\d+\s+(.*)", l)
        if m: out[m.group(2)[:60]] = int(m.group(1))
    return out
a, b = rows("gpurun_out/r05_kernel_stats_steps2.txt"), rows("gpurun_out/r05_kernel_stats_steps8.txt")
for k in a:
    if any(t in k for t in ("copyBuffer", "fillBuffer", "at::native", "split_f16_row", "causal_attention_f32_splitk")):
        print(f"{a[k]:6d} -> {b.get(k, 0):6d}  (6 -> 12 images)  {k}")
PY
rm -rf gpurun_out/prof_cb2 gpurun_out/prof_cb8
for f in $(ls tests/test_*.py | sort); do
  n=$(basename $f .py)
  timeout 1200 python -m pytest $f -m gpu -q -x -p no:cacheprovider > gpurun_out/verify/$n.log 2>&1
  echo "$n rc=$? $(grep -E 'passed|failed|error|no tests ran|deselected' gpurun_out/verify/$n.log | tail -1)"
done
timeout 300 python __graft_entry__.py --smoke > gpurun_out/verify/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/verify/smoke.log
timeout 900 python tools/parity_wide.py --modes f16x3 --sets panoptic:1024:1:0-15,referring:640:4:3-15,region:1024:2:3-7 --out gpurun_out/r05_parity_wide.jsonl > gpurun_out/r05_parity_wide_summary.json 2> gpurun_out/r05_parity_wide.err; tail -1 gpurun_out/r05_parity_wide_summary.json | cut -c1-1800
