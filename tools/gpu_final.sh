set -x
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4
python __graft_entry__.py --smoke 2>&1 | tail -3
timeout 900 python bench.py --breakdown gpurun_out/final_breakdown.json > gpurun_out/final_bench.log 2>&1; tail -1 gpurun_out/final_bench.log
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --eager"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_final_kt -- $CMD > $R/gpurun_out/prof_final_kt.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_final_fetch -- $CMD > $R/gpurun_out/prof_final_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_final_write -- $CMD > $R/gpurun_out/prof_final_write.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $R/gpurun_out/prof_final_sq -- $CMD > $R/gpurun_out/prof_final_sq.log 2>&1
cd $R
python tools/rocpd_stats.py gpurun_out/prof_final_kt/*/*_results.db 60 > gpurun_out/final_kernel_stats.txt
python tools/rocpd_pmc.py gpurun_out/prof_final_fetch/*/*_results.db gpurun_out/prof_final_write/*/*_results.db gpurun_out/prof_final_sq/*/*_results.db --top 16 --json gpurun_out/final_pmc.json > gpurun_out/final_pmc.txt 2>&1
rm -rf gpurun_out/prof_final_kt gpurun_out/prof_final_fetch gpurun_out/prof_final_write gpurun_out/prof_final_sq
