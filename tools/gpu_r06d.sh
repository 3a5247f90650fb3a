# r06d: loader-wave (LW) forms of the slice GEMM on hardware: bitwise tests, then the per-shape sweep
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_2_gemm.py -m gpu -q -x -p no:cacheprovider -k "mid_forms" > gpurun_out/r06d_pytest_gemm.log 2>&1; tail -3 gpurun_out/r06d_pytest_gemm.log
timeout 900 python tools/bench_gemm_x3.py --mid gpurun_out/r06d_mid_sweep.json > gpurun_out/r06d_mid_sweep.log 2>&1; tail -40 gpurun_out/r06d_mid_sweep.log
