"""Control fixture for the referring input of the wide parity set on which BOTH GPU arithmetics (three f16 products, exact fp32) leave the flip margin
by the same 214 pixels (r04 / r05: referring 640^2 batch 4, weights seed 0, inputs seed 11, image 1; 7.578e-3 of the logit range).

tools/exp_referring_controls.py (profiles/r06_referring_controls_seeds_10_11.jsonl) shows the fp32 CPU oracle tipping AGAINST ITSELF on this image,
to those same 214 pixels / 7.578e-3, under two controls that are each as exact as the reference's arithmetic or more: the same oracle on ONE host
thread (`threads1`: only the BLAS summation order changes) and the oracle with its three attention forms evaluated in float64 (`attn64`); with all of
the transformer arithmetic in float64 (`full64`) it is back on the fp32 oracle's side.  The reference's fp32 result is within rounding of a decision of
the thresholded attention-mask feedback (mask2former_transformer_decoder.py:754-760) here: two resting places 214 pixels apart.

This script stores WHERE the one-thread control differs from the multi-thread fp32 oracle -- (query, y, x) of every flipped pixel of image 1 -- so
that the gate (oracle/parity_gate.py) and tests/test_9_e2e_gpu.py can check that the pixels the product flips are either (almost) none or those.

    python tests/golden/make_referring_seed11_control.py        (CPU only, ~5 min; needs no /root/reference)"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import psalm_oracle as O  # noqa: E402
from psalm_amd.config import PsalmConfig  # noqa: E402
from psalm_amd.synthetic import make_inputs, make_state_dict  # noqa: E402

IMAGE = 1


def main():
    cfg = PsalmConfig(seg_task="referring")
    sd = make_state_dict(cfg, seed=0)
    inputs = make_inputs(cfg, "referring", size=640, batch=4, seed=11)
    nthr = torch.get_num_threads()
    torch.manual_seed(1234)
    want = O.eval_seg(sd, cfg, **inputs)[IMAGE]
    torch.set_num_threads(1)
    try:
        torch.manual_seed(1234)
        got = O.eval_seg(sd, cfg, **inputs)[IMAGE]
    finally:
        torch.set_num_threads(nthr)
    gm, wm = got["mask_pred"] > 0, want["mask_pred"] > 0
    idx = torch.nonzero(gm != wm).to(torch.int16).numpy()                     # (n, 3): query, y, x
    rel = float((got["mask_pred"] - want["mask_pred"]).abs().max() / want["mask_pred"].abs().max())
    np.savez_compressed(os.path.join(HERE, "referring_640_seed11_image1_threads1_control.npz"), flipped_qyx=idx, mask_logit_rel_err=np.float64(rel),
                        oracle_positive_pixels=np.int64(int(wm.sum())), oracle_threads=np.int64(nthr))
    print("flipped pixels:", idx.shape[0], "mask logit rel err:", rel, "oracle threads:", nthr)


if __name__ == "__main__":
    main()
