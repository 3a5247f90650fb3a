# r04 second GPU pass: what bounds the slice-form phased K loop (ablation), the new defaults (slice form on both Phi GEMMs), the seed-11 test
TAG=${1:-r04b}
set -x
mkdir -p gpurun_out
timeout 300 python tools/experiments/gemm_timeline.py --ablate gpurun_out/${TAG}_gemm_ablation.json > gpurun_out/${TAG}_gemm_ablation.jsonl 2> gpurun_out/${TAG}_gemm_ablation.err
cut -c1-400 gpurun_out/${TAG}_gemm_ablation.jsonl | head -60
timeout 900 python -m pytest tests/test_9_e2e_gpu.py tests/test_2_gemm.py -m gpu -q -x -p no:cacheprovider -k "seed11 or multi_seed or gemm_x3 or phased or ln_split or config3" > gpurun_out/${TAG}_pytest.log 2>&1; tail -3 gpurun_out/${TAG}_pytest.log
for i in 1 2; do
timeout 300 python bench.py --steps 20 --no-cpu-baseline --no-side-modes --breakdown gpurun_out/${TAG}_bench_breakdown.json 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_quick.json
python -c "import sys,json; d=json.load(open('gpurun_out/${TAG}_bench_quick.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel'][:90], r['avg_launch_us'], r['achieved'], r['all_mfma_gemms'])"
done
timeout 300 python bench.py --steps 5 --force-dist --no-cpu-baseline --no-side-modes > gpurun_out/${TAG}_nccl_stdout.txt 2>/dev/null; tail -c 300 gpurun_out/${TAG}_nccl_stdout.txt; echo; tail -1 gpurun_out/${TAG}_nccl_stdout.txt | python -c "import sys,json; json.loads(sys.stdin.read()); print('last stdout line is the JSON line')"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04b_bench_breakdown.json'))
for k,v in list(d['_gemm_shapes'].items())[:8]: print(v['ms_per_step'], v['launches_per_step'], v['TFLOPs'], k[:140])
PY
