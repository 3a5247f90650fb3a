#!/bin/bash
# round 3, GPU call 1: diagnostics before any kernel work (all outputs under gpurun_out/r03a/)
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03a
mkdir -p $O
rocm-smi --showproductname 2>/dev/null | head -5 > $O/box.txt
# 1. fp32 causal attention: main vs the fragment-order branch build
for i in 1 2; do
  timeout 300 python tools/bench_attn.py >> $O/attn_ab.jsonl 2>> $O/attn_ab.err
  PSALM_LIB=tools/experiments/_build/libpsalm_hip_attnfrag.so timeout 300 python tools/bench_attn.py >> $O/attn_ab.jsonl 2>> $O/attn_ab.err
done
# 2. block-level time line of the GEMM kernel
timeout 600 python tools/experiments/gemm_timeline.py $O/gemm_timeline.json > $O/gemm_timeline.log 2>&1
# 3. two images in flight
timeout 600 python tools/exp_inflight.py 2 20 > $O/inflight2.json 2> $O/inflight2.err
# 4. cross terms in MX e4m3 / dropped: numerics over 3 seeds
timeout 1200 python tools/exp_fp8cross.py 3 1024 > $O/fp8cross.jsonl 2> $O/fp8cross.err
tail -3 $O/attn_ab.jsonl; cat $O/inflight2.json; tail -5 $O/fp8cross.jsonl
