# Round-4 final GPU pass: kernel trace + HBM PMC passes of HEAD's default -> traffic file -> the full bench line (reads it) -> BASELINE configs
# 3 / 5 -> the whole gpu-marked suite file by file -> smoke.  Everything lands under gpurun_out/ (copied to profiles/ by hand).
TAG=${1:-r04}
set -x
mkdir -p gpurun_out gpurun_out/verify
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-side-modes --eager --no-overlap"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -- $CMD > $R/gpurun_out/${TAG}_prof_kt.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -- $CMD > $R/gpurun_out/${TAG}_prof_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -- $CMD > $R/gpurun_out/${TAG}_prof_write.log 2>&1
cd $R
python tools/rocpd_stats.py gpurun_out/prof_kt/*/*_results.db 70 > gpurun_out/${TAG}_kernel_stats.txt
python tools/rocpd_pmc.py gpurun_out/prof_fetch/*/*_results.db gpurun_out/prof_write/*/*_results.db --top 24 --json gpurun_out/${TAG}_pmc_hbm.json > gpurun_out/${TAG}_pmc_hbm.txt 2>&1
python tools/make_traffic_json.py gpurun_out/${TAG}_pmc_hbm.json gpurun_out/${TAG}_pmc_hbm_traffic.json "profiles/${TAG}_pmc_hbm.json"
cp gpurun_out/${TAG}_pmc_hbm_traffic.json profiles/${TAG}_pmc_hbm_traffic.json
rm -rf gpurun_out/prof_kt gpurun_out/prof_fetch gpurun_out/prof_write
head -16 gpurun_out/${TAG}_kernel_stats.txt | cut -c1-190
timeout 900 python bench.py --breakdown gpurun_out/${TAG}_bench_breakdown.json > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -1 gpurun_out/${TAG}_bench.json | cut -c1-1200
timeout 600 python tools/bench_configs.py --only 3,5 --skip-oracle --no-bf16 --json gpurun_out/${TAG}_configs_3_5.json > gpurun_out/${TAG}_configs.log 2>&1; tail -2 gpurun_out/${TAG}_configs.log | cut -c1-400
for f in $(ls tests/test_*.py | sort); do
  n=$(basename $f .py)
  timeout 900 python -m pytest $f -m gpu -q -x -p no:cacheprovider > gpurun_out/verify/$n.log 2>&1
  echo "$n rc=$? $(grep -E 'passed|failed|error|no tests ran|deselected' gpurun_out/verify/$n.log | tail -1)"
done
timeout 300 python __graft_entry__.py --smoke > gpurun_out/verify/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/verify/smoke.log
