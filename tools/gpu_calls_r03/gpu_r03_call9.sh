set -x
mkdir -p gpurun_out
TL_SHAPES=paired timeout 600 python tools/experiments/gemm_timeline.py gpurun_out/r03k_gemm_timeline_paired.json > gpurun_out/r03k_gemm_timeline_paired.log 2>&1; tail -3 gpurun_out/r03k_gemm_timeline_paired.log | cut -c1-300
timeout 600 python tools/exp_modes.py 1024 f16x3+overlap,f16x3+overlap+nopair,f16x3+overlap,f16x3+overlap+nopair > gpurun_out/r03k_paired_ab.log 2>&1; tail -5 gpurun_out/r03k_paired_ab.log | cut -c1-400
