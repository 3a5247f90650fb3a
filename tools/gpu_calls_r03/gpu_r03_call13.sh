set -x
mkdir -p gpurun_out
PARITY_X8=w1,w2,both PARITY_BATCH=4 timeout 600 python tools/parity_seeds.py referring 640 0:4 > gpurun_out/r03o_parity_referring_seed4_x8_per_gemm.jsonl 2> gpurun_out/r03o_parity.err; python - <<'PY'
import json
for l in open('gpurun_out/r03o_parity_referring_seed4_x8_per_gemm.jsonl'):
    r=json.loads(l)
    if 'mode' in r: print(r['mode'], r['image'], r['mask_iou_mean'], r['mask_iou_pooled'], r['flipped_pixels'], r['mask_logit_rel_err'])
PY
for m in w1 both off; do timeout 300 python bench.py --steps 20 --no-cpu-baseline --no-side-modes --llm-cross-fp8 $m 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$m', r['value'], r['ms_per_step'], r['roofline']['kernel'][-40:], r['roofline']['avg_launch_us'])"; done
