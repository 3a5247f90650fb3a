"""SURVEY §8 f1 -- the batch collator in front of eval_seg: psalm_amd.collate.DataCollatorForCOCODatasetV2 vs the golden written by
the REFERENCE's own class (tests/golden/make_collator_golden.py, psalm/train/train_datasets.py:968-1045) on the same seeded instances.
Integer / index work: exact equality.  The collated batch then feeds eval_seg's splice plan (vectorised == per-token form)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_collator_golden import make_instances  # noqa: E402  (pure-torch instance generator; the reference import lives in main())

from psalm_amd.collate import DataCollatorForCOCODatasetV2  # noqa: E402


@pytest.mark.parametrize("kind", ["panoptic", "ragged_images", "referring"])
def test_collator_equals_reference_golden(kind):
    z = np.load(os.path.join(HERE, "golden", "collator.npz"))
    tok = types.SimpleNamespace(pad_token_id=50256, model_max_length=48)
    instances = make_instances(7, kind)
    batch = DataCollatorForCOCODatasetV2(tokenizer=tok)(instances)
    want_keys = {k.split("/")[1] for k in z.files if k.startswith(kind + "/")}
    got_keys = {("images_list" if k == "images" and not torch.is_tensor(v) else "seg_info_keys" if k == "seg_info" else k) for k, v in batch.items()}
    assert got_keys == want_keys
    for k, v in batch.items():
        if torch.is_tensor(v):
            w = z[f"{kind}/{k}"]
            assert v.dtype == torch.from_numpy(w).dtype and np.array_equal(v.numpy(), w), k
        elif k in ("images", "token_refer_id"):
            name = "images_list" if k == "images" else k
            for j, t in enumerate(v):
                assert np.array_equal(t.numpy(), z[f"{kind}/{name}/{j}"])
        elif k == "seg_info":
            assert [",".join(sorted(d.keys())) for d in v] == z[f"{kind}/seg_info_keys"].tolist()
            assert all(a is b for a, b in zip(v, instances))                       # the instances themselves, mutated in place
        elif k == "dataset_type":
            assert list(v) == z[f"{kind}/dataset_type"].tolist()
    assert batch["input_ids"].shape[1] <= tok.model_max_length                   # truncation to model_max_length


def test_vectorised_splice_plan_equals_per_token_form():
    """PSALM._splice_plan (numpy, no per-token Python loop) == the straightforward per-token restatement of llava_phi.py:767-971 kept
    as PSALM._splice_plan_reference, on panoptic / ragged referring / region prompts."""
    from ops_backend import make_ops
    from psalm_amd.config import PsalmConfig
    from psalm_amd.model import PSALM
    from psalm_amd.synthetic import make_inputs, make_state_dict
    for task, batch in (("panoptic", 2), ("referring", 4), ("region", 3)):
        cfg = PsalmConfig.tiny(task)
        model = PSALM(cfg, make_state_dict(cfg, seed=1), ops=make_ops("emu"), precision="fp32")
        inp = make_inputs(cfg, task, size=96, batch=batch, seed=9, num_classes=9)
        n_regions = None
        if task == "region":
            n_regions = [int(s["instances"].region_masks.tensor.shape[0]) for s in inp["seg_info"]]
        args = (inp["input_ids"], inp["attention_mask"], 9, inp.get("class_name_ids"), inp.get("cls_indices"), inp.get("token_refer_id"),
                n_regions, "class_name_embedding_indices" in inp, "refer_embedding_indices" in inp)
        a, b = model._splice_plan(*args), model._splice_plan_reference(*args)
        assert a["L"] == b["L"] and a["lens"] == b["lens"] and a["n_cls"] == b["n_cls"]
        for k in ("sid", "srow", "kmask"):
            assert np.array_equal(a[k], b[k]), (task, k)
        for k in ("seg", "cls", "refer", "region"):
            assert (a[k] is None) == (b[k] is None)
            if a[k] is not None:
                assert np.array_equal(a[k][0], b[k][0]) and np.array_equal(a[k][1], b[k][1]), (task, k)


def test_region_mask_nonzero_fast_path_equals_torch_nonzero():
    """context_cluster.py:345-356 takes `mask.nonzero()` of every region mask on the host; the word-scanning form of it that
    PSALM.region_points uses returns the same (k, 2) int64 rows in the same (row-major) order on sparse blobs, scattered pixels, dense
    masks (falls back), empty masks, widths that are not a multiple of 8 (falls back), transposed views (falls back)."""
    from psalm_amd.model import _nonzero_2d
    g = torch.Generator().manual_seed(3)
    blob = torch.zeros(1024, 1024, dtype=torch.bool)
    blob[400:431, 299:316] = True
    blob[1023, 1023] = True
    blob[0, 0] = True
    cases = [blob, torch.zeros(64, 64, dtype=torch.bool), torch.ones(64, 64, dtype=torch.bool), torch.zeros(64, 64, dtype=torch.bool).t(),
             torch.ones(48, 40, dtype=torch.bool).t()]
    for shape, p in (((1024, 1024), 3e-4), ((640, 640), 0.01), ((640, 640), 0.2), ((37, 24), 0.5), ((5, 7), 0.4), ((3, 8), 0.1)):
        cases.append(torch.rand(shape, generator=g) < p)
    for m in cases:
        got, want = _nonzero_2d(m), m.nonzero()
        assert got.dtype == want.dtype and got.shape == want.shape and torch.equal(got, want)


def test_bucketed_splice_plan_is_the_plan_padded_like_a_ragged_batch():
    """PSALM._bucketed: the sequence length rounded up to `len_bucket` with zero-embedding, masked positions (what the shorter prompts of
    a ragged batch already are, LP:939-946), the CSR row sets re-based to the new row stride and padded behind their last offset -- so that
    one captured launch sequence serves every prompt length of a bucket (the graph key holds the bucketed length, not the exact one)."""
    from ops_backend import make_ops
    from psalm_amd.config import PsalmConfig
    from psalm_amd.model import PSALM
    from psalm_amd.synthetic import make_inputs, make_state_dict
    for task, batch in (("panoptic", 2), ("referring", 4), ("region", 3)):
        cfg = PsalmConfig.tiny(task)
        model = PSALM(cfg, make_state_dict(cfg, seed=1), ops=make_ops("emu"), precision="fp32")
        inp = make_inputs(cfg, task, size=96, batch=batch, seed=9, num_classes=9)
        n_regions = None
        if task == "region":
            n_regions = [int(s["instances"].region_masks.tensor.shape[0]) for s in inp["seg_info"]]
        a = model._splice_plan(inp["input_ids"], inp["attention_mask"], 9, inp.get("class_name_ids"), inp.get("cls_indices"),
                               inp.get("token_refer_id"), n_regions, "class_name_embedding_indices" in inp, "refer_embedding_indices" in inp)
        L = a["L"]
        for q in (0, 1, 8, 32, 64):
            model.len_bucket = q
            b = model._bucketed(a, batch)
            Lp = L if q <= 1 else (L + q - 1) // q * q
            assert b["L"] == Lp and b["lens"] == a["lens"] and b["n_cls"] == a["n_cls"]
            assert model._bucketed(a, batch) is b or q <= 1                               # cached with the plan
            for k, fill in (("sid", -1), ("srow", 0), ("kmask", 0)):
                assert b[k].shape == (batch, Lp) and np.array_equal(b[k][:, :L], a[k]) and (b[k][:, L:] == fill).all(), (task, q, k)
            for k in ("seg", "cls", "refer", "region"):
                assert (a[k] is None) == (b[k] is None)
                if a[k] is None:
                    continue
                (ao, ar), (bo, br) = a[k], b[k]
                assert np.array_equal(ao, bo)
                n = ar.shape[0]
                assert np.array_equal(br[:n] // Lp, ar // L) and np.array_equal(br[:n] % Lp, ar % L), (task, q, k)
                if q > 1 and k != "seg":
                    assert br.shape[0] % q == 0 and br.shape[0] >= max(n, q) and (br[n:] == 0).all()
                else:
                    assert br.shape[0] == n


@pytest.mark.parametrize("task,batch", [("panoptic", 1), ("referring", 3)])
def test_bucketed_length_gives_the_results_of_the_exact_length(task, batch):
    """Whole tiny model on the host emulator with len_bucket = 32 (default) vs 0: the padded positions change nothing a result reads --
    integer outputs identical, logits equal to fp32 summation-order noise (the GEMMs may pick another tile / split for the larger M)."""
    from ops_backend import make_ops
    from psalm_amd.config import PsalmConfig
    from psalm_amd.model import PSALM
    from psalm_amd.synthetic import make_inputs, make_state_dict
    cfg = PsalmConfig.tiny(task)
    sd = make_state_dict(cfg, seed=12)
    inp = make_inputs(cfg, task, size=96, batch=batch, seed=4, num_classes=9, pad=32 if task == "referring" else 0)
    outs = {}
    for q in (0, 32):
        model = PSALM(cfg, sd, ops=make_ops("emu"), precision="fp32")
        model.len_bucket = q
        outs[q] = model.eval_seg(**inp)
        blob, layout, meta = model._prepare(inp["input_ids"], inp["attention_mask"], inp["images"], inp["seg_info"], inp.get("class_name_ids"),
                                            inp.get("class_name_embedding_indices"), inp.get("cls_indices"), inp.get("token_refer_id"),
                                            inp.get("refer_embedding_indices"), None)
        assert meta["L"] % 32 == 0 if q else meta["L"] == max(meta["lens"])
    for a, b in zip(outs[0], outs[32]):
        rng = a["mask_pred"].abs().max()
        assert (a["mask_pred"] - b["mask_pred"]).abs().max() <= 2e-5 * rng
        assert torch.equal(a["instances"].pred_masks, b["instances"].pred_masks)
        if task == "panoptic":
            assert torch.equal(a["panoptic_seg"][0], b["panoptic_seg"][0]) and a["panoptic_seg"][1] == b["panoptic_seg"][1]
