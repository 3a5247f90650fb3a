# r04h: HEAD re-check after the final pass -- the two tests edited since (seed-11 knife edge, bf16 golden bar), BASELINE config 5 with the
# word-scanning region-mask nonzero on the host, and the bench line with the parity leg widened to panoptic seeds 0-15 (default arithmetic AND
# the exact-fp32 control at HEAD).
set -x
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_9_e2e_gpu.py -m gpu -q -p no:cacheprovider -k "seed11 or golden_referring_384" > gpurun_out/r04h_pytest.log 2>&1; tail -3 gpurun_out/r04h_pytest.log
timeout 300 python tools/bench_configs.py --only 5 --skip-oracle --no-bf16 --json gpurun_out/r04h_config5.json > gpurun_out/r04h_config5.log 2>&1; tail -1 gpurun_out/r04h_config5.log | cut -c1-300
timeout 700 python bench.py --parity-seeds 16 > gpurun_out/r04h_bench_16seeds.json 2> gpurun_out/r04h_bench_16seeds.err; tail -1 gpurun_out/r04h_bench_16seeds.json | cut -c1-600
