# Round profile pass on the GPU box: default bench line (+ CPU baseline, bf16 side line), rocprofv3 kernel trace of the same workload,
# the two HBM-traffic PMC passes (separate runs, no trace domains), per-kernel summaries under gpurun_out/ (copied to profiles/ by hand).
#   bash tools/gpu_profile.sh <tag>            e.g. r02e
TAG=${1:-r03}
set -x
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 600 python bench.py --breakdown gpurun_out/${TAG}_breakdown.json > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -1 gpurun_out/${TAG}_bench.json | cut -c1-600
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-side-modes --eager --no-overlap"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -- $CMD > $R/gpurun_out/${TAG}_prof_kt.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -- $CMD > $R/gpurun_out/${TAG}_prof_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -- $CMD > $R/gpurun_out/${TAG}_prof_write.log 2>&1
cd $R
python tools/rocpd_stats.py gpurun_out/prof_kt/*/*_results.db 70 > gpurun_out/${TAG}_kernel_stats.txt
python tools/rocpd_pmc.py gpurun_out/prof_fetch/*/*_results.db gpurun_out/prof_write/*/*_results.db --top 24 --json gpurun_out/${TAG}_pmc_hbm.json > gpurun_out/${TAG}_pmc_hbm.txt 2>&1
python tools/make_traffic_json.py gpurun_out/${TAG}_pmc_hbm.json gpurun_out/${TAG}_pmc_hbm_traffic.json "profiles/${TAG}_pmc_hbm.json"
rm -rf gpurun_out/prof_kt gpurun_out/prof_fetch gpurun_out/prof_write
head -30 gpurun_out/${TAG}_kernel_stats.txt
