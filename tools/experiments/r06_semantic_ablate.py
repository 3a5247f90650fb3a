"""r06: what the fused semantic pass (semantic_from_masks_x3_pair_kernel, 100 queries x 133 classes x 1024^2 pixels) costs without its stores / matrix
instructions / sigmoids / loads / split: side libraries built with -DPSALM_SEM_ABL=<bits> (tools/experiments/build_side_lib.sh), each timed on the
same inputs.  Results of ablated builds are meaningless; only the durations are read.
    for b in 0 1 2 4 8 16 3 12 31; do tools/experiments/build_side_lib.sh semabl$b WORKTREE postproc -DPSALM_SEM_ABL=$b; done
    python tools/experiments/r06_semantic_ablate.py out.json"""
import glob, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from psalm_amd.hip_ops import Ops, get_ops


def main():
    base = get_ops()
    d = base.device
    Q, C, H = 100, 133, 1024
    g = torch.Generator().manual_seed(0)
    mask = (torch.randn(Q, H * H, generator=g) * 4).to(d)
    cls = torch.randn(Q, C + 1, generator=g).to(d)
    out = {}
    libs = sorted(glob.glob(os.path.join(ROOT, "tools/experiments/_build/libpsalm_hip_semabl*.so")), key=lambda p: int(p.split("semabl")[1].split(".")[0]))
    for lib in libs:
        ops = Ops(lib)
        bits = int(lib.split("semabl")[1].split(".")[0])
        probs, probsT, score, label = ops.class_softmax(cls, 128, probsT_dtype=torch.float32)
        fn = lambda: ops.semantic_from_masks(mask, probsT, want_mask_score=True)
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        n = 30
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        ev[0].record()
        for i in range(n):
            fn()
            ev[i + 1].record()
        torch.cuda.synchronize()
        ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n))
        out[str(bits)] = {"median_us": ts[n // 2], "min_us": ts[0]}
        print(bits, out[str(bits)], flush=True)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
