R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1u_kt -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --eager > $R/gpurun_out/prof_r1u_kt.log 2>&1
cd $R && python tools/rocpd_stats.py gpurun_out/prof_r1u_kt/*/*_results.db 60 > gpurun_out/r1u_kernel_stats.txt; rm -rf gpurun_out/prof_r1u_kt
