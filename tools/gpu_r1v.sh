set -x
timeout 900 python -m pytest tests/test_e2e_gpu.py -m gpu -q -x -k "video or fp8 or graph" 2>&1 | tail -3
grep video gpurun_out/parity_report.jsonl | cut -c1-400
