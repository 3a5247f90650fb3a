# r04j: the row-walking resize_planes kernel on the hardware: op tests (bit-equality with the one-pixel-per-thread kernel), the tests whose
# results pass through it (evaluator outputs, drop-in sequence, the 512 / 1024 panoptic goldens and configuration), and the default bench line.
set -x
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_1_ops.py tests/test_7_dropin.py tests/test_8_evalout.py tests/test_8_preprocess.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r04j_pytest_ops.log 2>&1; tail -3 gpurun_out/r04j_pytest_ops.log
timeout 500 python -m pytest tests/test_9_e2e_gpu.py -m gpu -q -x -p no:cacheprovider -k "golden_panoptic_512 or config2_panoptic_1024_f16x3 or tiny_vs_oracle" --durations=8 > gpurun_out/r04j_pytest_e2e.log 2>&1; tail -14 gpurun_out/r04j_pytest_e2e.log
timeout 600 python bench.py --breakdown gpurun_out/r04j_bench_breakdown.json > gpurun_out/r04j_bench.json 2> gpurun_out/r04j_bench.err; tail -1 gpurun_out/r04j_bench.json | cut -c1-400
python - <<'PY'
import json
b = json.loads(open("gpurun_out/r04j_bench.json").read().strip().splitlines()[-1])
print([(h["kernel"][:28], h["avg_launch_us"], h["frac"]) for h in b["roofline"]["hbm_bound_kernels"]])
print([r["flipped_mask_pixels"] for r in b["parity_vs_cpu_oracle"]["seeds"]["per_seed"]], b["parity_vs_cpu_oracle"]["meets_north_star_bar"], b["value"])
PY
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    -d $R/gpurun_out/prof_sq -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-side-modes --eager --no-overlap > $R/gpurun_out/r04j_prof_sq.log 2>&1
cd $R
python tools/rocpd_pmc.py gpurun_out/prof_sq/*/*_results.db --top 40 --json gpurun_out/r04j_pmc_sq.json > gpurun_out/r04j_pmc_sq.txt 2>&1
python tools/sq_fractions.py gpurun_out/r04j_pmc_sq.json --top 40 > gpurun_out/r04j_sq_fractions.txt 2>&1
rm -rf gpurun_out/prof_sq
head -8 gpurun_out/r04j_sq_fractions.txt | cut -c1-160; grep -E "resize|panoptic_argmax|semantic|msda" gpurun_out/r04j_sq_fractions.txt | cut -c1-160
