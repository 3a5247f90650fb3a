# r06e: automatic loader-wave selection in the model: GEMM suite on hardware, then quick bench lines with PSALM_TUNE_GEMM_MID off / on (twice each)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_2_gemm.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r06e_pytest_gemm.log 2>&1; tail -3 gpurun_out/r06e_pytest_gemm.log
B="python bench.py --no-side-modes --no-cpu-baseline --no-varied"
for t in off on off2 on2; do
  case $t in off*) TUNE="--tuning 2=0";; *) TUNE="";; esac
  timeout 300 $B $TUNE --breakdown gpurun_out/r06e_breakdown_$t.json > gpurun_out/r06e_bench_$t.json 2> gpurun_out/r06e_bench_$t.err
done
python - <<'PY'
import json
for t in ("off", "on", "off2", "on2"):
    try:
        b = json.loads(open(f"gpurun_out/r06e_bench_{t}.json").read().strip().splitlines()[-1])
        d = json.load(open(f"gpurun_out/r06e_breakdown_{t}.json"))
        print(t, "value", b["value"], "gpu_ms", b["gpu_ms_per_step"], {k: round(d[k]["ms_per_step"], 3) for k in ("psalm_gemm_x3_ln_split", "psalm_gemm_x3", "psalm_gemm_x3_split")})
        if t in ("off", "on"):
            for k, v in sorted(d["_gemm_shapes"].items(), key=lambda kv: -kv[1]["ms_per_step"])[2:22]:
                print("   ", round(v["ms_per_step"], 3), v.get("TFLOPs"), k[:120])
    except Exception as e:
        print(t, "failed", e)
PY
