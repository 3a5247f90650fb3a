set -x
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -E "^\s*(Name|name)?.*(MFMA|FETCH_SIZE|WRITE_SIZE|SQ_WAIT_INST_ANY|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|GRBM_GUI_ACTIVE|SQ_ACTIVE_INST_ANY|SQ_WAIT_ANY|SQ_INSTS_VALU_MFMA|TCC_HIT|TCC_MISS|LDS_BANK)" | cut -c1-160 | sort -u | head -40
timeout 600 python bench.py --steps 10 --warmup 2 --breakdown gpurun_out/breakdown_r1g.json > gpurun_out/bench_r1g.log 2>&1; tail -1 gpurun_out/bench_r1g.log
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --eager"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1g_kt -- $CMD > $R/gpurun_out/prof_r1g_kt.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_r1g_fetch -- $CMD > $R/gpurun_out/prof_r1g_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_r1g_write -- $CMD > $R/gpurun_out/prof_r1g_write.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $R/gpurun_out/prof_r1g_sq -- $CMD > $R/gpurun_out/prof_r1g_sq.log 2>&1
cd $R
python tools/rocpd_stats.py gpurun_out/prof_r1g_kt/*/*_results.db 45 > gpurun_out/r1g_kernel_stats.txt
python tools/rocpd_pmc.py gpurun_out/prof_r1g_fetch/*/*_results.db gpurun_out/prof_r1g_write/*/*_results.db gpurun_out/prof_r1g_sq/*/*_results.db --top 12 --json gpurun_out/r1g_pmc.json > gpurun_out/r1g_pmc.txt 2>&1
tail -3 gpurun_out/prof_r1g_sq.log
rm -rf gpurun_out/prof_r1g_kt gpurun_out/prof_r1g_fetch gpurun_out/prof_r1g_write gpurun_out/prof_r1g_sq
