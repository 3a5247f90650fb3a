# r03 last call: profile pass of the final default (x8 on [dense|fc2] only) + configs 3 / 5 parity in that mode
set -x
mkdir -p gpurun_out
bash tools/gpu_profile.sh r03p > gpurun_out/r03p_profile.log 2>&1; tail -1 gpurun_out/r03p_bench.json | cut -c1-330; head -6 gpurun_out/r03p_kernel_stats.txt | cut -c1-200
timeout 330 python tools/bench_configs.py --no-bf16 --json gpurun_out/r03p_configs.json > gpurun_out/r03p_configs.log 2>&1; python - <<'PY'
import json
for l in open('gpurun_out/r03p_configs.log'):
    if not l.startswith('{'): continue
    r=json.loads(l); p=r.get('parity',{})
    print(r['config'],r['x8_gemms'],r['images_per_s'],{k:p.get(k) for k in ('mask_iou_mean_min','mask_iou_pooled_min','mask_logit_rel_err_max','flipped_pixels_max','meets_north_star_bar')})
PY
