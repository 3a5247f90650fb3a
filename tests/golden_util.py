"""Helpers shared by the golden-vector tests (CPU oracle and GPU candidate)."""
import ast
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RNG_SEED_AT_CALL = 1234          # tests/golden/make_golden.py seeds the global RNG with this before eval_seg


def load_case(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    case = ast.literal_eval(str(z["meta_case"]))
    return case, z


def check_signature(z, key, t, rtol, atol_scale=1.0, what=""):
    """Compare tensor `t` with the stored signature of stage `key`.
    Tolerance: |a-b| <= rtol * absmax(stage)  (stage-relative, robust to near-zero entries).
    A signature is a SAMPLE, not the tensor: numel, mean, absmax and 256 values at fixed pseudo-random flat positions
    (tests/golden/make_golden.py::signature) -- a tripwire for a broken stage.  Element-wise equality with the reference is what the full
    arrays of the goldens (pred_masks_s4, sem_seg_argmax, panoptic_ids, the class / SEG / region logits) and the full-tensor oracle
    comparisons of tests/test_9_e2e_gpu.py::test_config* establish; a signature match alone does not."""
    t = t.detach().float().cpu().contiguous().view(-1)
    assert int(z[f"sig_{key}_numel"]) == t.numel(), f"{key}: numel {t.numel()} vs golden {int(z[f'sig_{key}_numel'])}"
    idx = torch.from_numpy(z[f"sig_{key}_idx"])
    ref = torch.from_numpy(z[f"sig_{key}_val"])
    scale = float(z[f"sig_{key}_absmax"]) * atol_scale
    err = (t[idx] - ref).abs().max().item()
    assert err <= rtol * scale, f"{what}{key}: max|d|={err:.3e} > {rtol:g} * absmax {scale:.3e}"
    mean_err = abs(float(t.double().mean()) - float(z[f"sig_{key}_mean"]))
    assert mean_err <= rtol * scale, f"{what}{key}: mean differs by {mean_err:.3e}"
    return err / max(scale, 1e-30)
