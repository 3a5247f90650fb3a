# r05t: epilogue operands fetched before the K loop (64 x 128 / 128 x 128 fp32-output kernels) + the batched loads of the mask decoder's
# cross-attention: ring sweep against the previous gemm.hip build, GEMM tests, quick bench
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_2_gemm.py tests/test_1_ops.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r05t_pytest.log 2>&1; tail -2 gpurun_out/r05t_pytest.log
timeout 400 python tools/bench_gemm_x3.py --ring gpurun_out/r05t_gemm_x3_ring_sweep.json > gpurun_out/r05t_ring_new.log 2>&1
timeout 400 python tools/bench_gemm_x3.py --ring --lib tools/experiments/_build/libpsalm_hip_prepipe.so gpurun_out/r05t_gemm_x3_ring_sweep_head.json > gpurun_out/r05t_ring_old.log 2>&1
python - <<'PY'
import json
n = json.load(open("gpurun_out/r05t_gemm_x3_ring_sweep.json")); o = json.load(open("gpurun_out/r05t_gemm_x3_ring_sweep_head.json"))
tn = to = 0
for k in n:
    a, b = n[k]["auto"]["us"], o[k]["auto"]["us"]; tn += a; to += b
    print(f"{k:24s} {b:7.1f} -> {a:7.1f}  {100 * (a - b) / b:+5.1f} %")
print("sum auto: head", round(to, 1), "new", round(tn, 1))
PY
for i in 1 2; do timeout 300 python bench.py --no-side-modes --no-cpu-baseline --no-varied > gpurun_out/r05t_bench_quick.json 2> gpurun_out/r05t_bench_quick.err; tail -1 gpurun_out/r05t_bench_quick.json | cut -c1-230; done
