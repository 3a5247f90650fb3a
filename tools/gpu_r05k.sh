# r05k: where is the GPU idle inside one image?  Kernel trace of the GRAPH-REPLAY bench (the timed configuration: two streams, hipGraph) and of the
# eager single-stream form, idle gaps per hand-over (tools/rocpd_gaps.py)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_gr -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-side-modes --no-varied > $R/gpurun_out/r05k_prof_graph.log 2>&1
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_eg -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-side-modes --no-varied --eager --no-overlap > $R/gpurun_out/r05k_prof_eager.log 2>&1
cd $R
python tools/rocpd_gaps.py gpurun_out/prof_gr/*/*_results.db --images 6 --top 40 > gpurun_out/r05k_gaps_graph.txt 2>&1; head -50 gpurun_out/r05k_gaps_graph.txt | cut -c1-220
python tools/rocpd_gaps.py gpurun_out/prof_eg/*/*_results.db --images 3 --top 25 > gpurun_out/r05k_gaps_eager.txt 2>&1; head -12 gpurun_out/r05k_gaps_eager.txt | cut -c1-220
rm -rf gpurun_out/prof_gr gpurun_out/prof_eg
