# r06a: (1) the op / GEMM suites on hardware -- incl. the 12 cases r05 left self-skipped (gate deleted) and the new placement tests;
# (2) XCD placement A/B (psalm_set_tuning 0 / 1 = split-K GEMM slices / causal-attention heads per XCD): quick bench lines with per-kernel
# breakdowns, old placement vs new; (3) FETCH_SIZE / WRITE_SIZE passes of the new placement.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_1_ops.py tests/test_2_gemm.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r06a_pytest_ops.log 2>&1; tail -3 gpurun_out/r06a_pytest_ops.log
B="python bench.py --no-side-modes --no-cpu-baseline --no-varied"
timeout 300 $B --tuning 0=0,1=0 --breakdown gpurun_out/r06a_breakdown_old.json > gpurun_out/r06a_bench_old.json 2> gpurun_out/r06a_bench_old.err
timeout 300 $B --breakdown gpurun_out/r06a_breakdown_new.json > gpurun_out/r06a_bench_new.json 2> gpurun_out/r06a_bench_new.err
timeout 300 $B --tuning 0=0,1=0 --breakdown gpurun_out/r06a_breakdown_old2.json > gpurun_out/r06a_bench_old2.json 2> gpurun_out/r06a_bench_old2.err
timeout 300 $B --breakdown gpurun_out/r06a_breakdown_new2.json > gpurun_out/r06a_bench_new2.json 2> gpurun_out/r06a_bench_new2.err
python - <<'PY'
import json
for t in ("old", "new", "old2", "new2"):
    try:
        b = json.loads(open(f"gpurun_out/r06a_bench_{t}.json").read().strip().splitlines()[-1])
        d = json.load(open(f"gpurun_out/r06a_breakdown_{t}.json"))
        print(t, "value", b["value"], "gpu_ms", b["gpu_ms_per_step"], {k: round(d[k]["ms_per_step"], 3) for k in ("psalm_gemm_x3_ln_split", "psalm_causal_attention_f32_split", "psalm_gemm_x3", "psalm_gemm_x3_split")},
              (b.get("parity_vs_cpu_oracle") or {}).get("meets_north_star_bar"))
    except Exception as e:
        print(t, "failed", e)
PY
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-side-modes --no-varied --parity-seeds 0 --eager --no-overlap"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -- $CMD > $R/gpurun_out/r06a_prof_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -- $CMD > $R/gpurun_out/r06a_prof_write.log 2>&1
cd $R
python tools/rocpd_pmc.py gpurun_out/prof_fetch/*/*_results.db gpurun_out/prof_write/*/*_results.db --top 24 --json gpurun_out/r06a_pmc_hbm.json > gpurun_out/r06a_pmc_hbm.txt 2>&1
python tools/make_traffic_json.py gpurun_out/r06a_pmc_hbm.json gpurun_out/r06a_pmc_hbm_traffic.json "profiles/r06a_pmc_hbm.json"
rm -rf gpurun_out/prof_fetch gpurun_out/prof_write
head -30 gpurun_out/r06a_pmc_hbm.txt | cut -c1-220
