"""bench.py cannot run without a GPU, and a NameError in the code that assembles its `roofline` object would void the round's bench line.
This test executes exactly that block of bench.py (from `hbm_roof = None` to `if args.breakdown:`) on synthetic per-kernel sums, with the
module-level constants / helpers of bench.py and the committed counter files of profiles/ (HBM traffic, SQ fractions)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOMINANT = "gemm_bf16_glds_kernel<float, 256, 256, 2, 4, 2, false, 32, 4, 2, true, true>"


def _run_block(kern, hbm):
    src = open(os.path.join(ROOT, "bench.py")).read()
    lines = src.split("\n")
    start = [i for i, l in enumerate(lines) if l.strip() == "hbm_roof = None"]
    end = [i for i, l in enumerate(lines) if l.strip().startswith("if args.breakdown:")]
    assert len(start) == 1 and len(end) == 1 and start[0] < end[0]
    block = "\n".join(l[8:] for l in lines[start[0]:end[0]])
    ns = {"__file__": os.path.join(ROOT, "bench.py"), "__name__": "bench_extract"}
    exec(src.split("def main")[0], ns)                                   # constants, _profiled_name, file locations
    ns.update(dict(hbm=hbm, kern=kern, nprof=2, ev_over=0.005))
    exec(block, ns)
    return ns["roof"]


def test_roofline_object_is_assembled_for_the_default_kernel_and_reads_the_committed_counter_files():
    kern = {DOMINANT: [48, 7.38, 48 * 52.79e9], "gemm_bf16_glds_kernel<float, 64, 128, 2, 2, 2, false, 32, 0, 2, false>": [100, 2.7, 1.0e12],
            DOMINANT.replace("true, true", "false") + " + splitk_reduce_ln_kernel": [46, 6.2, 46 * 37.7e9]}
    hbm = {"panoptic_argmax_kernel (+ the call's small kernels)": [2, 0.340, 855638016], "resize_planes_rows_kernel<8>": [2, 0.193, 891289600],
           "semantic_from_masks_x3_pair_kernel": [2, 0.80, 2 * 977272832], "msda_fused8_kernel": [12, 0.64, 12 * 68812800]}
    roof = _run_block(kern, hbm)
    json.dumps(roof)                                                      # must serialise: it goes into the one JSON line
    assert roof["bound"] == "mfma" and roof["kernel"] == DOMINANT and roof["peak"] == 2500.0
    assert abs(roof["frac"] - roof["achieved"] / 2500.0) < 1e-3 and 0.0 < roof["frac"] < 1.0
    assert roof["mfma_issue"]["f16_product_equivalents"] == 3
    # the committed counter passes were taken with the <.., 32, 3, 2, ..> form of the same kernel: found under that name
    assert isinstance(roof["traffic"], int) and roof["traffic"] > 176e6
    sq = roof["sq_counters"]
    assert sq["kernel_in_the_pass"] == DOMINANT.replace("32, 4, 2", "32, 3, 2") and 0.4 < sq["matrix_pipe_busy"] < 0.8 and 1.0 < sq["clock_GHz"] < 2.6
    names = [h["kernel"] for h in roof["hbm_bound_kernels"]]
    assert set(names) == set(hbm)
    for h in roof["hbm_bound_kernels"]:
        assert h["bound"] == "hbm" and h["peak"] == 8000.0 and abs(h["frac"] - h["achieved"] / 8000.0) < 1e-3
        assert ("note" in h) == h["kernel"].startswith("panoptic_argmax")      # the one entry whose byte count is an upper bound says so
    sem = next(h for h in roof["hbm_bound_kernels"] if h["kernel"].startswith("semantic"))
    assert sem["traffic"] and abs(sem["traffic"] / sem["algorithmic_bytes_per_launch"] - 1.0) < 0.05


def test_profiled_name_maps_only_the_padding_variant():
    ns = {"__file__": os.path.join(ROOT, "bench.py"), "__name__": "bench_extract"}
    exec(open(os.path.join(ROOT, "bench.py")).read().split("def main")[0], ns)
    f = ns["_profiled_name"]
    assert f(DOMINANT) == DOMINANT.replace("32, 4, 2", "32, 3, 2")
    other = "gemm_bf16_glds_kernel<float, 128, 128, 2, 2, 2, false, 32, 0, 2, true, true>"
    assert f(other) == other
