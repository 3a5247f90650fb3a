# the whole gpu-marked suite file by file + smoke (dress rehearsal / final)
set -x
mkdir -p gpurun_out gpurun_out/verify
for f in $(ls tests/test_*.py | sort); do
  n=$(basename $f .py)
  timeout 1200 python -m pytest $f -m gpu -q -x -p no:cacheprovider --durations=6 > gpurun_out/verify/$n.log 2>&1
  echo "$n rc=$? $(grep -E 'passed|failed|error|no tests ran|deselected' gpurun_out/verify/$n.log | tail -1)"
done
timeout 300 python __graft_entry__.py --smoke > gpurun_out/verify/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/verify/smoke.log
