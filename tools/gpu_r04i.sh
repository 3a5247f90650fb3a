# r04i: (1) the full-size configuration tests with the state dict / model shared between consecutive tests (durations reported);
# (2) one SQ-counter pass of the bench command (eager, single stream): where the wavefront cycles of the top kernels go.
set -x
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_9_e2e_gpu.py -m gpu -q -p no:cacheprovider -k "config2 or config3 or config5" --durations=12 > gpurun_out/r04i_pytest_configs.log 2>&1; tail -22 gpurun_out/r04i_pytest_configs.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    -d $R/gpurun_out/prof_sq -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-side-modes --eager --no-overlap > $R/gpurun_out/r04i_prof_sq.log 2>&1
cd $R
python tools/rocpd_pmc.py gpurun_out/prof_sq/*/*_results.db --top 16 --json gpurun_out/r04i_pmc_sq.json > gpurun_out/r04i_pmc_sq.txt 2>&1
python tools/sq_fractions.py gpurun_out/r04i_pmc_sq.json --top 16 > gpurun_out/r04i_sq_fractions.txt 2>&1
rm -rf gpurun_out/prof_sq
cat gpurun_out/r04i_sq_fractions.txt | cut -c1-200
