# Short end-of-round GPU pass: op-level tests, smoke, bench (+ CPU baseline), kernel trace and the two HBM-traffic PMC passes.
set -x
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_ops.py tests/test_abi.py tests/test_msda.py -m gpu -q 2>&1 | tail -2
python __graft_entry__.py --smoke 2>&1 | tail -2
timeout 400 python bench.py --breakdown gpurun_out/final_breakdown.json > gpurun_out/final_bench.log 2>&1; tail -1 gpurun_out/final_bench.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --eager"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_final_kt -- $CMD > $R/gpurun_out/prof_final_kt.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_final_fetch -- $CMD > $R/gpurun_out/prof_final_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_final_write -- $CMD > $R/gpurun_out/prof_final_write.log 2>&1
cd $R
python tools/rocpd_stats.py gpurun_out/prof_final_kt/*/*_results.db 60 > gpurun_out/final_kernel_stats.txt
python tools/rocpd_pmc.py gpurun_out/prof_final_fetch/*/*_results.db gpurun_out/prof_final_write/*/*_results.db --top 20 --json gpurun_out/final_pmc.json > gpurun_out/final_pmc.txt 2>&1
rm -rf gpurun_out/prof_final_kt gpurun_out/prof_final_fetch gpurun_out/prof_final_write
head -12 gpurun_out/final_kernel_stats.txt
