# SQ counters of the fp32 causal attention kernel alone (tools/bench_attn.py), one rocprofv3 --pmc pass (no trace domains).
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/attn
cd /tmp && export TMPDIR=/tmp
PSALM_ATTN_PAIR=${1:-1} timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU \
    -d $R/gpurun_out/prof_attn -- python $R/tools/bench_attn.py > $R/gpurun_out/attn/pmc.log 2>&1
cd $R
python tools/rocpd_pmc.py gpurun_out/prof_attn/*/*_results.db --top 4 --json gpurun_out/attn/pmc_sq.json > gpurun_out/attn/pmc_sq.txt 2>&1
rm -rf gpurun_out/prof_attn
cat gpurun_out/attn/pmc_sq.txt | cut -c1-400 | head -40
