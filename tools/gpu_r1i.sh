set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm.py tests/test_ops.py -m gpu -q -x 2>&1 | tail -3
timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_builder.py -m gpu -q -x 2>&1 | tail -3
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --breakdown gpurun_out/breakdown_r1i.json > gpurun_out/bench_r1i.log 2>&1; tail -1 gpurun_out/bench_r1i.log | cut -c1-1200
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1i_kt -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --eager > $R/gpurun_out/prof_r1i_kt.log 2>&1
cd $R && python tools/rocpd_stats.py gpurun_out/prof_r1i_kt/*/*_results.db 50 > gpurun_out/r1i_kernel_stats.txt; rm -rf gpurun_out/prof_r1i_kt
