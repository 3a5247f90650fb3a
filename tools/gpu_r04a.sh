# Round-4 first GPU pass (VERDICT r03 "Next" #1, #2, #6): wide parity of the DEFAULT arithmetic with the exact-fp32 mode as control, the RCCL
# dry run on one GPU, one clean profile pass at HEAD (kernel trace + the two HBM PMC passes), BASELINE configs 3 / 5 in the default arithmetic,
# the host-thread sweep of the CPU oracle.  Everything lands under gpurun_out/ (copied to profiles/ by hand).
TAG=${1:-r04a}
set -x
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
nproc; free -g | head -2
# 1. wide parity: panoptic 1024 seeds 0-15, referring 640 b4 seeds 3-15 (= the r03s CPU sets), region 1024 b2 seeds 3-7 (config 5)
timeout 1500 python tools/parity_wide.py --sets panoptic:1024:1:0-15,referring:640:4:3-15,region:1024:2:3-7 --out gpurun_out/${TAG}_parity_wide.jsonl \
    > gpurun_out/${TAG}_parity_wide.log 2>&1; tail -1 gpurun_out/${TAG}_parity_wide.log | cut -c1-1500
# 2. host-thread sweep of the oracle (bench.py reads profiles/r04_cpu_baseline_threads.json; on this first pass it is not there yet -> 16)
timeout 600 python tools/cpu_baseline_sweep.py --out gpurun_out/${TAG}_cpu_baseline_threads.json > gpurun_out/${TAG}_cpu_sweep.log 2>&1; tail -1 gpurun_out/${TAG}_cpu_sweep.log
cp gpurun_out/${TAG}_cpu_baseline_threads.json profiles/r04_cpu_baseline_threads.json
# 3. the bench line with the RCCL dry run (world size 1: init_process_group("nccl"), broadcast, MIN / MAX checksum all-reduce, barriers)
timeout 900 python bench.py --force-dist --breakdown gpurun_out/${TAG}_bench_breakdown.json > gpurun_out/${TAG}_bench_nccl_world1.json 2> gpurun_out/${TAG}_bench.err
tail -1 gpurun_out/${TAG}_bench_nccl_world1.json | cut -c1-900; tail -3 gpurun_out/${TAG}_bench.err
# 4. BASELINE configs 3 / 5, default arithmetic: throughput + dominant kernels (parity of the same configs: step 1)
timeout 600 python tools/bench_configs.py --only 3,5 --skip-oracle --no-bf16 --json gpurun_out/${TAG}_configs_3_5.json > gpurun_out/${TAG}_configs.log 2>&1; tail -2 gpurun_out/${TAG}_configs.log | cut -c1-600
# 5. kernel trace + PMC passes of the default arithmetic at HEAD
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
head -24 gpurun_out/${TAG}_kernel_stats.txt
# 5b. r04 kernel candidate: the 256 x 256 phased K loop on 32-deep slices (policy 2581) -- unit tests on the hardware, 4-image parity, whole-model A/B
timeout 600 python -m pytest tests/test_2_gemm.py -m gpu -q -x -p no:cacheprovider -k "phased_slice or gemm_x3" > gpurun_out/${TAG}_pytest_gemm.log 2>&1; tail -2 gpurun_out/${TAG}_pytest_gemm.log
timeout 600 python tools/parity_wide.py --sets panoptic:1024:1:0-3 --modes f16x3 --gemm-policy 2581 --out gpurun_out/${TAG}_parity_phased_slice.jsonl > gpurun_out/${TAG}_parity_phased_slice.log 2>&1; tail -1 gpurun_out/${TAG}_parity_phased_slice.log | cut -c1-700
for pol in 2580 2581 2580 2581; do
  timeout 300 python bench.py --steps 20 --no-cpu-baseline --no-side-modes --gemm-policy $pol 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('policy $pol', d['value'], d['ms_per_step'], r['kernel'][:90], r['avg_launch_us'], r['achieved'])"
done 2>&1 | tee gpurun_out/${TAG}_bench_phased_slice_ab.txt
# 6. stand-alone GEMM baselines for the kernel work of the round: Phi shapes back to back, and their block time lines
timeout 300 python tools/bench_gemm_x3.py gpurun_out/${TAG}_gemm_x3_sweep.json > gpurun_out/${TAG}_gemm_x3_sweep.log 2>&1; head -3 gpurun_out/${TAG}_gemm_x3_sweep.log
[ -f tools/experiments/_build/libpsalm_hip_tl.so ] && timeout 300 python tools/experiments/gemm_timeline.py gpurun_out/${TAG}_gemm_timeline.json > gpurun_out/${TAG}_gemm_timeline.log 2>&1
tail -4 gpurun_out/${TAG}_gemm_timeline.log | cut -c1-400
