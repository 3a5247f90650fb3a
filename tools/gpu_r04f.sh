# r04 sixth GPU pass: MSDeformAttn gather kernel with quad-shared bilinear taps -- unit tests, stand-alone A/B, whole-model A/B
TAG=${1:-r04f}
set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_3_msda.py -m gpu -q -x -p no:cacheprovider > gpurun_out/${TAG}_pytest.log 2>&1; tail -3 gpurun_out/${TAG}_pytest.log
timeout 300 python tools/bench_msda.py > gpurun_out/${TAG}_bench_msda.jsonl 2> gpurun_out/${TAG}_bench_msda.err; cat gpurun_out/${TAG}_bench_msda.jsonl | cut -c1-600
for flag in "" "--msda-per-lane" "" "--msda-per-lane"; do
timeout 300 python bench.py --steps 20 --no-cpu-baseline --no-side-modes $flag 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_quick.json
python -c "import sys,json; d=json.load(open('gpurun_out/${TAG}_bench_quick.json')); r=d['roofline']; print('msda[$flag]', d['value'], d['ms_per_step'], [ (h['kernel'], h['avg_launch_us'], h['frac']) for h in r['hbm_bound_kernels'] if 'msda' in h['kernel']])"
done 2>&1 | tee gpurun_out/${TAG}_bench_msda_model_ab.txt
