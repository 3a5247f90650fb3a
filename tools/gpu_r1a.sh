set -x
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.jsonl
python -c "import torch; print(torch.cuda.get_device_name(0)); import os; print(os.cpu_count())"
python __graft_entry__.py --smoke 2>&1 | tail -4
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15
cat gpurun_out/parity_report.jsonl
timeout 900 python bench.py --steps 5 --warmup 2 --breakdown gpurun_out/breakdown_r1a.json > gpurun_out/bench_r1a.log 2>&1; tail -3 gpurun_out/bench_r1a.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench_r1a -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench_r1a.log 2>&1
find $R/gpurun_out/prof_bench_r1a -name '*stats*' | head
