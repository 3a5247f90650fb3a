# r06t: grid shapes of every launch of one eager image (2 steps): which launches leave the chip idle
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-side-modes --no-varied --eager --no-overlap > $R/gpurun_out/r06t_prof_kt.log 2>&1
cd $R
python tools/rocpd_grids.py gpurun_out/prof_kt/*/*_results.db 20 > gpurun_out/r06t_grids.txt 2>&1
rm -rf gpurun_out/prof_kt
head -5 gpurun_out/r06t_grids.txt
