set -x
mkdir -p gpurun_out
PARITY_FP32=1 PARITY_BATCH=4 timeout 600 python tools/parity_seeds.py referring 640 0:4 > gpurun_out/r03n_parity_referring_seed4_modes.jsonl 2> gpurun_out/r03n_parity.err; cut -c1-420 gpurun_out/r03n_parity_referring_seed4_modes.jsonl | tail -14
timeout 300 python tools/bench_semantic.py > gpurun_out/r03n_semantic_tile_order.jsonl 2>&1; cat gpurun_out/r03n_semantic_tile_order.jsonl | tail -5
