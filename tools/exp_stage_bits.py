"""Experiment (not product): WHICH stage of eval_seg needs the wide GEMM operands?  tools/exp_bits.py rounds every GEMM operand of the
whole path to m mantissa bits; here only the GEMMs of ONE stage (Swin / projector+LLM / pixel decoder / masked decoder) are rounded
(m = 10 explicit bits ~ a single f16 operand with row scales, m = 7 ~ bf16) and every other stage stays exact fp32.  The question it
answers: could some stage drop from the 3-product split-f16 GEMM to a 1-product f16/bf16 GEMM and keep the north-star bar
(pooled mask IoU >= 0.999, >= 99.9 % identical labels vs the fp32 reference)?

  python tools/exp_stage_bits.py [size]     -> one JSON line per (stage, m)"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from psalm_amd.config import PsalmConfig
from psalm_amd.model import PSALM
from psalm_amd.synthetic import make_inputs, make_state_dict
from psalm_amd import hip_ops as H


def rnd(t, m):                                   # round to nearest at m explicit mantissa bits
    sh = 23 - m
    i = t.contiguous().view(torch.int32)
    i = (i + (1 << (sh - 1))) & ~((1 << sh) - 1)
    return i.view(torch.float32)


from tools._metrics import metrics, clone  # noqa

STAGES = {"swin": ("swin",), "llm": ("projector", "llm"), "pixel_decoder": ("pixel_decoder",), "predictor": ("predictor",)}


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    emu = os.environ.get("PSALM_EXP_EMU") == "1"          # dry run of this script on the host emulator (tiny model, no GPU)
    cfg = PsalmConfig.tiny("panoptic") if emu else PsalmConfig(seg_task="panoptic")
    sd = make_state_dict(cfg, seed=0)
    inputs = make_inputs(cfg, "panoptic", size=size, batch=1, seed=0)
    if emu:
        sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "emu")]
        from ops_backend import make_ops
        model = PSALM(cfg, sd, ops=make_ops("emu"), precision="fp32")
    else:
        inputs["images"] = inputs["images"].cuda()
        model = PSALM(cfg, sd, precision="fp32")
    ref = clone(model.eval_seg(**inputs)[0])
    ops = model.ops
    orig = ops.gemm
    state = {"stage": None, "on": (), "m": 23, "n": 0}

    def gemm_q(a, w, bias=None, residual=None, act=H.ACT_NONE, act_col_start=0, out=None, out_dtype=None):
        if state["stage"] in state["on"] and w.dtype == torch.float32 and a.dtype == torch.float32:
            state["n"] += 1
            return orig(rnd(a, state["m"]), rnd(w, state["m"]), bias, residual, act, act_col_start, out, out_dtype)
        return orig(a, w, bias, residual, act, act_col_start, out, out_dtype)

    def tag(name):
        fn = getattr(model, name)

        def wrapped(*a, **k):
            prev, state["stage"] = state["stage"], name
            try:
                return fn(*a, **k)
            finally:
                state["stage"] = prev
        setattr(model, name, wrapped)
    for names in STAGES.values():
        for n in names:
            tag(n)
    ops.gemm = gemm_q
    out = {}
    runs = [(s, 10) for s in STAGES] + [("all", 10), ("llm", 7), ("predictor", 7)]
    for s, m in runs:
        on = sum((STAGES[x] for x in (STAGES if s == "all" else s.split("+"))), ())
        state.update(on=on, m=m, n=0)
        got = clone(model.eval_seg(**inputs)[0])
        if not emu:
            torch.cuda.synchronize()
        r = metrics(got, ref)
        r["gemms_rounded"] = state["n"]
        out[f"{s}@m{m}"] = r
        print(s, m, json.dumps(r), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
