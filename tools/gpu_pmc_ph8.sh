#!/bin/bash
# PMC passes over tools/prof_gemm_ph8.py (counters only: no trace domains)
R=$(pwd); cd /tmp; export TMPDIR=/tmp
CMD="python $R/tools/prof_gemm_ph8.py $1 $2 $3"
cd $R
timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc8_a -- $CMD > $R/gpurun_out/pmc8_a.log 2>&1
timeout 120 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS -d $R/gpurun_out/pmc8_b -- $CMD > $R/gpurun_out/pmc8_b.log 2>&1
timeout 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS -d $R/gpurun_out/pmc8_c -- $CMD > $R/gpurun_out/pmc8_c.log 2>&1
python tools/rocpd_pmc.py gpurun_out/pmc8_a/*/*_results.db gpurun_out/pmc8_b/*/*_results.db gpurun_out/pmc8_c/*/*_results.db --top 4 --json gpurun_out/pmc8.json > gpurun_out/pmc8.txt 2>&1
grep -A40 "gemm_bf16_glds" gpurun_out/pmc8.txt | head -90
