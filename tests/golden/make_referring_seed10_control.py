"""Control fixture for the second referring input of the wide parity set outside the flip margin (r05 / r06: referring 640^2 batch 4, weights seed 0,
inputs seed 10, image 2; 58 pixels, 2.2e-3 of the logit range; the exact-fp32 GPU mode of r04 had the same 58 pixels).

tools/exp_referring_controls.py (profiles/r06_referring_controls_seeds_10_11.jsonl): of sixteen controls of the fp32 CPU oracle, the one that evaluates
every GroupNorm (the pixel decoder's input projections and FPN layers, msdeformattn.py:196-254) in float64 -- MORE exact than the reference -- moves
this image by 2.217e-3 / 58 pixels, the product's pixels; with all of the arithmetic in float64 (`every64`) the oracle is back on its fp32 side.  And
tools/experiments/r06_seed10_stage_bisect.py (profiles/r06_referring_seed10_stage_bisect.jsonl, on the MI355X box): the ORACLE's own predictor, fed the
product's seg-query / SEG embeddings (1.96e-6 from its own) together with the product's pixel-decoder outputs (1.4 - 3.7e-6 from its own), returns the
product's result to the digit (8 low-resolution pixels, 2.352e-3), while either substitution alone, and the product's predictor on the oracle's
inputs, stay on the oracle's side: the reference's predictor turns an fp32-rounding-sized change of its inputs into 2.4e-3 here (the thresholded
attention-mask feedback, mask2former_transformer_decoder.py:754-760).  Bucketing is not involved: len_bucket 0 / 32 / 64 give the same words on the GPU
and `padL` is exact on the CPU (profiles/r06_referring_seed10_bucket_ab.jsonl).

This script stores WHERE the float64-GroupNorm control differs from the fp32 oracle: (query, y, x) of every flipped pixel of image 2.

    python tests/golden/make_referring_seed10_control.py        (CPU only, ~4 min; needs no /root/reference)"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import psalm_oracle as O  # noqa: E402
from psalm_amd.config import PsalmConfig  # noqa: E402
from psalm_amd.synthetic import make_inputs, make_state_dict  # noqa: E402

IMAGE = 2


class _F64GroupNorm:
    """torch.nn.functional with group_norm evaluated in float64 (the oracle reaches it as `F.group_norm`)"""
    def __getattr__(self, name):
        fn = getattr(F, name)
        if name != "group_norm":
            return fn
        return lambda x, G, w=None, b=None, eps=1e-5: fn(x.double(), G, None if w is None else w.double(), None if b is None else b.double(), eps).float()


def main():
    cfg = PsalmConfig(seg_task="referring")
    sd = make_state_dict(cfg, seed=0)
    inputs = make_inputs(cfg, "referring", size=640, batch=4, seed=10)
    torch.manual_seed(1234)
    want = O.eval_seg(sd, cfg, **inputs)[IMAGE]
    O.F = _F64GroupNorm()
    try:
        torch.manual_seed(1234)
        got = O.eval_seg(sd, cfg, **inputs)[IMAGE]
    finally:
        O.F = F
    gm, wm = got["mask_pred"] > 0, want["mask_pred"] > 0
    idx = torch.nonzero(gm != wm).to(torch.int16).numpy()                     # (n, 3): query, y, x
    rel = float((got["mask_pred"] - want["mask_pred"]).abs().max() / want["mask_pred"].abs().max())
    np.savez_compressed(os.path.join(HERE, "referring_640_seed10_image2_groupnorm64_control.npz"), flipped_qyx=idx, mask_logit_rel_err=np.float64(rel),
                        oracle_positive_pixels=np.int64(int(wm.sum())), oracle_threads=np.int64(torch.get_num_threads()))
    print("flipped pixels:", idx.shape[0], "mask logit rel err:", rel)


if __name__ == "__main__":
    main()
