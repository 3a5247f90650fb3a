# Full verification pass: per-kernel files first (each in its own process), then whole-model tests, smoke, bench.
mkdir -p gpurun_out/verify
for f in test_1_ops test_2_gemm test_3_msda test_7_builder test_9_e2e_gpu; do
  timeout 600 python -m pytest tests/$f.py -m gpu -q -x -p no:cacheprovider > gpurun_out/verify/$f.log 2>&1
  echo "$f rc=$? $(grep -E 'passed|failed|error' gpurun_out/verify/$f.log | tail -1)"
done
timeout 200 python __graft_entry__.py --smoke > gpurun_out/verify/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/verify/smoke.log
timeout 600 python bench.py > gpurun_out/verify/bench.log 2>gpurun_out/verify/bench.err; echo "bench rc=$?"; tail -1 gpurun_out/verify/bench.log | cut -c1-1500
