set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_e2e_gpu.py -m gpu -q -x 2>&1 | tail -3
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --eager > gpurun_out/bench_r1d_eager.log 2>&1; tail -1 gpurun_out/bench_r1d_eager.log | cut -c1-250
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --breakdown gpurun_out/breakdown_r1d.json > gpurun_out/bench_r1d.log 2>&1; tail -1 gpurun_out/bench_r1d.log | cut -c1-250
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1d -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --eager > $R/gpurun_out/prof_r1d.log 2>&1
cd $R && python tools/rocpd_stats.py gpurun_out/prof_r1d/*/*_results.db 45 > gpurun_out/prof_r1d_stats.txt; rm -rf gpurun_out/prof_r1d
