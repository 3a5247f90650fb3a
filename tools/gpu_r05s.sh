# r05s: mask-decoder cross-attention with its loads batched (prologue + per-tile staging) against HEAD's kernel, by kernel trace
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
LIBS=psalm_amd/lib/libpsalm_hip.so,tools/experiments/_build/libpsalm_hip_mha2.so
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_mha -- python $R/tools/bench_mha.py --libs $LIBS > $R/gpurun_out/r05s_bench_mha.jsonl 2>&1
cd $R
cut -c1-300 gpurun_out/r05s_bench_mha.jsonl
python tools/rocpd_blocks.py gpurun_out/prof_mha/*/*_results.db mha_attention_f32_mfma 105 > gpurun_out/r05s_mha_trace_blocks.txt 2>&1
python tools/rocpd_blocks.py gpurun_out/prof_mha/*/*_results.db mha_f32_combine 105 >> gpurun_out/r05s_mha_trace_blocks.txt 2>&1
rm -rf gpurun_out/prof_mha
cat gpurun_out/r05s_mha_trace_blocks.txt | cut -c1-150
