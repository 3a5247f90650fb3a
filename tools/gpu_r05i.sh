mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_9_e2e_gpu.py -m gpu -q -p no:cacheprovider -k "stage_level or tiny_vs_oracle or graph_replay or golden_region or golden_video or golden_referring or config3_referring_640_batch4 or config5_region_1024_batch2" > gpurun_out/r05i_pytest.log 2>&1; tail -5 gpurun_out/r05i_pytest.log
timeout 300 python bench.py --no-side-modes --no-cpu-baseline --no-varied > gpurun_out/r05i_bench_quick.json 2> gpurun_out/r05i_bench_quick.err; tail -1 gpurun_out/r05i_bench_quick.json | cut -c1-260; tail -2 gpurun_out/r05i_bench_quick.err
timeout 300 python bench.py --no-side-modes --no-cpu-baseline --no-varied --eager --steps 10 > gpurun_out/r05i_bench_eager.json 2> gpurun_out/r05i_bench_eager.err; tail -1 gpurun_out/r05i_bench_eager.json | cut -c1-200
python - <<'PY'
# eager launch path: stage-level calls vs op-by-op Python launches (host time per image when nothing is captured)
import time, torch, sys
sys.path.insert(0, ".")
from psalm_amd.config import PsalmConfig
from psalm_amd.model import PSALM
from psalm_amd.synthetic import make_inputs, make_state_dict
cfg = PsalmConfig(seg_task="panoptic"); sd = make_state_dict(cfg, seed=0)
m = PSALM(cfg, sd, precision="f16x3", use_graphs=False)
inp = make_inputs(cfg, "panoptic", size=1024, batch=1, seed=0); inp["images"] = inp["images"].cuda()
for flag in (True, False, True, False):
    m.c_stages = flag
    for _ in range(3): m.eval_seg(**inp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m.eval_seg(**inp)
    torch.cuda.synchronize(); print("eager eval_seg, c_stages =", flag, round((time.perf_counter() - t0) * 100, 2), "ms per image", flush=True)
PY
