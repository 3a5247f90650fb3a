# r03 call 6: full GPU test tier, smoke, default bench line, BASELINE configs 3 / 5 in the qualifying mode
set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03h_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/r03h_pytest_gpu.log
cp -f gpurun_out/parity_report.jsonl gpurun_out/r03h_parity_report.jsonl 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 600 python bench.py --breakdown gpurun_out/r03h_breakdown.json > gpurun_out/r03h_bench.json 2> gpurun_out/r03h_bench.err; tail -1 gpurun_out/r03h_bench.json | cut -c1-900
timeout 900 python tools/bench_configs.py --json gpurun_out/r03h_configs.json > gpurun_out/r03h_configs.log 2>&1; cut -c1-700 gpurun_out/r03h_configs.log | tail -6
timeout 600 python tools/bench_gemm_x3.py gpurun_out/r03h_gemm_x3_sweep.json > gpurun_out/r03h_gemm_x3_sweep.log 2>&1; tail -25 gpurun_out/r03h_gemm_x3_sweep.log
