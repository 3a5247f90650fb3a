set -x
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.jsonl
python __graft_entry__.py --smoke 2>&1 | tail -4
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_e2e_gpu.py 2>&1 | tail -8
timeout 1500 python -m pytest tests/test_e2e_gpu.py -m gpu -q 2>&1 | tail -40
cat gpurun_out/parity_report.jsonl
timeout 900 python bench.py --steps 5 --warmup 2 --breakdown gpurun_out/breakdown_r1.json 2>&1 | tail -5
cat gpurun_out/breakdown_r1.json | head -80
