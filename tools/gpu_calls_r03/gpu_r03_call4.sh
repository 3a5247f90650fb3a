#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03e
mkdir -p $O
timeout 900 python -m pytest tests/test_2_gemm.py -x -q -m gpu > $O/test_gemm.log 2>&1; echo "rc=$?" >> $O/test_gemm.log
timeout 600 python tools/experiments/gemm_timeline.py $O/gemm_timeline.json > $O/gemm_timeline.log 2>&1
timeout 900 python bench.py --steps 20 --no-side-modes --breakdown $O/bench_breakdown.json > $O/bench.json 2> $O/bench.err
tail -3 $O/test_gemm.log; cat $O/bench.json | cut -c1-200
