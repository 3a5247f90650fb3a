#!/bin/bash
# round 3, GPU call 3: direct fp32 epilogue -- correctness on the hardware, time line, GEMM sweep, end-to-end
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03c
mkdir -p $O
timeout 900 python -m pytest tests/test_2_gemm.py -x -q -m gpu > $O/test_gemm.log 2>&1; echo "rc=$?" >> $O/test_gemm.log
timeout 600 python tools/experiments/gemm_timeline.py $O/gemm_timeline.json > $O/gemm_timeline.log 2>&1
timeout 600 python tools/bench_gemm_x3.py $O/gemm_x3_sweep.json > $O/gemm_x3_sweep.log 2>&1
timeout 900 python bench.py --steps 20 --no-side-modes --breakdown $O/bench_breakdown.json > $O/bench.json 2> $O/bench.err
tail -3 $O/test_gemm.log; cat $O/bench.json | cut -c1-600
