set -x
mkdir -p gpurun_out
python -c "import torch; print(torch.cuda.get_device_name(0), torch.cuda.device_count())"
rocminfo | grep -m3 -E "gfx|Compute Unit" 
nproc; free -g | head -2
python __graft_entry__.py --smoke 2>&1 | tail -3
python -m pytest tests -m gpu -q 2>&1 | tail -8
python tools/bench_msda.py 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_msda -- python $GRAFT_REPO_ROOT/tools/bench_msda.py > $GRAFT_REPO_ROOT/gpurun_out/prof_msda.log 2>&1
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof_msda | head -20
