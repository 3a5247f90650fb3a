# r04 third GPU pass: Phi attention in split-f16 arithmetic (csrc/attention_x3.hip): unit tests on the hardware, whole-model A/B, wide parity, trace
TAG=${1:-r04c}
set -x
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_1_ops.py tests/test_2_gemm.py -m gpu -q -x -p no:cacheprovider -k "causal or paired or split_output or window" > gpurun_out/${TAG}_pytest.log 2>&1; tail -3 gpurun_out/${TAG}_pytest.log
for flag in "" "--attn-fp32" "" "--attn-fp32"; do
timeout 300 python bench.py --steps 20 --no-cpu-baseline --no-side-modes $flag 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_quick.json
python -c "import sys,json; d=json.load(open('gpurun_out/${TAG}_bench_quick.json')); r=d['roofline']; print('attn[$flag]', d['value'], d['ms_per_step'], r['kernel'][:90], r['avg_launch_us'], r['achieved'], r['all_mfma_gemms'])"
done 2>&1 | tee gpurun_out/${TAG}_bench_attn_ab.txt
timeout 1200 python tools/parity_wide.py --sets panoptic:1024:1:0-15,referring:640:4:3-15,region:1024:2:3-7 --modes f16x3 --out gpurun_out/${TAG}_parity_wide_attn_x3.jsonl \
    > gpurun_out/${TAG}_parity_wide.log 2>&1; tail -1 gpurun_out/${TAG}_parity_wide.log | cut -c1-2500
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-side-modes --eager --no-overlap"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -- $CMD > $R/gpurun_out/${TAG}_prof_kt.log 2>&1
cd $R
python tools/rocpd_stats.py gpurun_out/prof_kt/*/*_results.db 40 > gpurun_out/${TAG}_kernel_stats.txt
rm -rf gpurun_out/prof_kt
head -26 gpurun_out/${TAG}_kernel_stats.txt | cut -c1-200
