set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_2_gemm.py -m gpu -x -q > gpurun_out/r03l_pytest_gemm.log 2>&1; echo "pytest gemm exit $?"; tail -3 gpurun_out/r03l_pytest_gemm.log
TL_SHAPES=paired timeout 600 python tools/experiments/gemm_timeline.py gpurun_out/r03l_gemm_timeline_so_specialised.json > gpurun_out/r03l_gemm_timeline.log 2>&1; tail -2 gpurun_out/r03l_gemm_timeline.log | cut -c1-200
timeout 600 python tools/exp_modes.py 1024 f16x3+overlap,f16x3+overlap+nopair,f16x3+overlap,f16x3+overlap+nopair > gpurun_out/r03l_paired_ab.log 2>&1; tail -5 gpurun_out/r03l_paired_ab.log | cut -c1-330
