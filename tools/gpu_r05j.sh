# r05j: hand-pipelined fragment reads in the slice-form K loop (64 x 128 / 128 x 128 tiles): GEMM tests on hardware, the ring sweep against the
# previous build of gemm.hip (tools/experiments/_build/libpsalm_hip_prepipe.so), a quick bench line
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_2_gemm.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r05j_pytest_gemm.log 2>&1; tail -3 gpurun_out/r05j_pytest_gemm.log
timeout 400 python tools/bench_gemm_x3.py --ring gpurun_out/r05j_gemm_x3_ring_sweep.json > gpurun_out/r05j_ring_new.log 2>&1; tail -30 gpurun_out/r05j_ring_new.log
timeout 400 python tools/bench_gemm_x3.py --ring --lib tools/experiments/_build/libpsalm_hip_prepipe.so gpurun_out/r05j_gemm_x3_ring_sweep_prepipe.json > gpurun_out/r05j_ring_old.log 2>&1; tail -30 gpurun_out/r05j_ring_old.log
timeout 300 python bench.py --no-side-modes --no-cpu-baseline --no-varied --breakdown gpurun_out/r05j_bench_breakdown.json > gpurun_out/r05j_bench_quick.json 2> gpurun_out/r05j_bench_quick.err; tail -1 gpurun_out/r05j_bench_quick.json | cut -c1-300; tail -2 gpurun_out/r05j_bench_quick.err
