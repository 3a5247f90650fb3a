timeout 600 python -m pytest tests/test_gemm.py -m gpu -q -x --tb=short 2>&1 | grep -v "^$" | tail -30
