# r04m: last call of the round: the open-vocabulary tests at HEAD, and the kernel trace of HEAD's bench command (row-walking resize included).
set -x
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 150 python -m pytest tests/test_9_e2e_gpu.py -m gpu -q -p no:cacheprovider -k "open_vocab" > gpurun_out/r04m_pytest.log 2>&1; tail -2 gpurun_out/r04m_pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-side-modes --eager --no-overlap > $R/gpurun_out/r04m_prof_kt.log 2>&1
cd $R
python tools/rocpd_stats.py gpurun_out/prof_kt/*/*_results.db 70 > gpurun_out/r04m_kernel_stats.txt
rm -rf gpurun_out/prof_kt
head -8 gpurun_out/r04m_kernel_stats.txt | cut -c1-170; grep -E "resize|panoptic_argmax|semantic_from" gpurun_out/r04m_kernel_stats.txt | cut -c1-170
