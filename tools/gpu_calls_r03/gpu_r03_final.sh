# r03 final pass: full GPU test tier, smoke, profile pass (bench + kernel trace + PMC), BASELINE configs 3 / 5
set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/r03_pytest_gpu.log
cp -f gpurun_out/parity_report.jsonl gpurun_out/r03_parity_report.jsonl 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -4 | tee gpurun_out/r03_smoke.txt
bash tools/gpu_profile.sh r03 > gpurun_out/r03_profile.log 2>&1; tail -1 gpurun_out/r03_bench.json | cut -c1-500; head -12 gpurun_out/r03_kernel_stats.txt
timeout 900 python tools/bench_configs.py --json gpurun_out/r03_configs.json > gpurun_out/r03_configs.log 2>&1; cut -c1-300 gpurun_out/r03_configs.log | grep "^{" | tail -4
