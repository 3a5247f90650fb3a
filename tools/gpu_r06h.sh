# r06h: native post-processing + everything so far on hardware: the e2e suite (all configs), ops suites, one quick bench line
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_9_e2e_gpu.py -m gpu -q -x -p no:cacheprovider --durations=8 > gpurun_out/r06h_pytest_e2e.log 2>&1; tail -14 gpurun_out/r06h_pytest_e2e.log
timeout 600 python -m pytest tests/test_1_ops.py tests/test_2_gemm.py tests/test_3_msda.py tests/test_7_dropin.py tests/test_8_evalout.py tests/test_8_preprocess.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r06h_pytest_ops.log 2>&1; tail -3 gpurun_out/r06h_pytest_ops.log
timeout 300 python bench.py --no-side-modes --no-cpu-baseline --no-varied > gpurun_out/r06h_bench.json 2> gpurun_out/r06h_bench.err; tail -1 gpurun_out/r06h_bench.json | cut -c1-300
