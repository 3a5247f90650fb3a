# Round-6 profile pass of the kernels HEAD ships (the round-end pass): kernel trace + the two HBM PMC passes + one SQ-counter pass of
# `bench.py --steps 2 --eager --no-overlap`, summarised under the EXACT kernel names -> profiles/<tag>_* ON THE BOX, so that a bench run later in
# the same call reads this round's counter files.       bash tools/gpu_r06_profile.sh <tag>
TAG=${1:-r06}
set -x
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-side-modes --no-varied --eager --no-overlap"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -- $CMD > $R/gpurun_out/${TAG}_prof_kt.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -- $CMD > $R/gpurun_out/${TAG}_prof_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -- $CMD > $R/gpurun_out/${TAG}_prof_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    -d $R/gpurun_out/prof_sq -- $CMD > $R/gpurun_out/${TAG}_prof_sq.log 2>&1
cd $R
python tools/rocpd_stats.py gpurun_out/prof_kt/*/*_results.db 70 > gpurun_out/${TAG}_kernel_stats.txt
python tools/rocpd_pmc.py gpurun_out/prof_fetch/*/*_results.db gpurun_out/prof_write/*/*_results.db --top 24 --json gpurun_out/${TAG}_pmc_hbm.json > gpurun_out/${TAG}_pmc_hbm.txt 2>&1
python tools/make_traffic_json.py gpurun_out/${TAG}_pmc_hbm.json gpurun_out/${TAG}_pmc_hbm_traffic.json "profiles/${TAG}_pmc_hbm.json"
python tools/rocpd_pmc.py gpurun_out/prof_sq/*/*_results.db --top 40 --json gpurun_out/${TAG}_pmc_sq.json > gpurun_out/${TAG}_pmc_sq.txt 2>&1
python tools/sq_fractions.py gpurun_out/${TAG}_pmc_sq.json --top 40 > gpurun_out/${TAG}_sq_fractions.txt 2>&1
python tools/make_sq_json.py gpurun_out/${TAG}_pmc_sq.json gpurun_out/${TAG}_sq_summary.json "profiles/${TAG}_pmc_sq.json"
cp gpurun_out/${TAG}_pmc_hbm_traffic.json profiles/r06_pmc_hbm_traffic.json
cp gpurun_out/${TAG}_sq_summary.json profiles/r06_sq_summary.json
rm -rf gpurun_out/prof_kt gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq
head -14 gpurun_out/${TAG}_kernel_stats.txt | cut -c1-200
head -8 gpurun_out/${TAG}_sq_fractions.txt | cut -c1-170
