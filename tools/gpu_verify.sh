# Full verification pass, the way the driver runs it at round end: the whole gpu-marked suite in file order (kernel tests first), smoke,
# the default bench line.  Each test file in its own process, so a GPU fault names its file and the later files still run.
mkdir -p gpurun_out/verify
for f in $(ls tests/test_*.py | sort); do
  n=$(basename $f .py)
  timeout 900 python -m pytest $f -m gpu -q -x -p no:cacheprovider > gpurun_out/verify/$n.log 2>&1
  echo "$n rc=$? $(grep -E 'passed|failed|error|no tests ran|deselected' gpurun_out/verify/$n.log | tail -1)"
done
timeout 300 python __graft_entry__.py --smoke > gpurun_out/verify/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/verify/smoke.log
timeout 600 python bench.py > gpurun_out/verify/bench.log 2>gpurun_out/verify/bench.err; echo "bench rc=$?"; tail -1 gpurun_out/verify/bench.log | cut -c1-700
