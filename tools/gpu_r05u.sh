mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_sk -- python $R/tools/bench_skinny.py --libs tools/experiments/_build/libpsalm_hip_prepipe.so,psalm_amd/lib/libpsalm_hip.so > $R/gpurun_out/r05u_bench_skinny.jsonl 2>&1
cd $R
python tools/rocpd_blocks.py gpurun_out/prof_sk/*/*_results.db gemm_f32_skinny 33 > gpurun_out/r05u_skinny_trace_blocks.txt 2>&1
rm -rf gpurun_out/prof_sk
cut -c1-200 gpurun_out/r05u_bench_skinny.jsonl | tail -4; cut -c1-110 gpurun_out/r05u_skinny_trace_blocks.txt
