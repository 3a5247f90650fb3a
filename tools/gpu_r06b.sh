# r06b: the mid-size forms of the split-f16 slice GEMM on hardware: bitwise tests, then the per-shape sweep (tools/bench_gemm_x3.py --mid)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_2_gemm.py -m gpu -q -x -p no:cacheprovider -k "mid_forms or xcd or x3" > gpurun_out/r06b_pytest_gemm.log 2>&1; tail -3 gpurun_out/r06b_pytest_gemm.log
timeout 900 python tools/bench_gemm_x3.py --mid gpurun_out/r06b_mid_sweep.json > gpurun_out/r06b_mid_sweep.log 2>&1; tail -40 gpurun_out/r06b_mid_sweep.log
