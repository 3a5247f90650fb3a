"""SURVEY §8 f1 -- the text fields in front of eval_seg: psalm_amd.prompts vs the golden written by the REFERENCE's own dataset classes
(tests/golden/make_prompt_golden.py runs their `__getitem__` / `preprocess_llama2` / `tokenizer_special_tokens` / `preprocess_class_name` /
`preprocess_referring_instruction` under the `llava_phi` conversation template) with the same stub tokenizers.  Integer work: exact equality."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_prompt_golden import CASES, CharStub, ShortWordStub, WordStub, case_key  # noqa: E402  (pure-python stubs; the reference import lives in main())

from psalm_amd import prompts as P  # noqa: E402
from psalm_amd.config import CLS_TOKEN_INDEX, IMAGE_TOKEN_INDEX, REFER_TOKEN_INDEX, REGION_TOKEN_INDEX, SEG_TOKEN_INDEX  # noqa: E402

BUILD = {"panoptic": P.panoptic_sample, "semantic": P.semantic_sample, "instance": P.instance_sample, "region": P.region_sample,
         "referring": P.referring_sample}


@pytest.mark.parametrize("tok", [WordStub(), ShortWordStub(), CharStub()], ids=lambda t: type(t).__name__)
def test_text_fields_equal_the_reference_datasets(tok):
    z = np.load(os.path.join(HERE, "golden", "prompts.npz"))
    for i, (task, arg) in enumerate(CASES):
        key = case_key(i, task, tok)
        got = BUILD[task](arg, tok)
        want_keys = {f.split("/", 1)[1] for f in z.files if f.startswith(key + "/")}
        assert set(got) == want_keys, (key, set(got), want_keys)
        for k, v in got.items():
            w = z[f"{key}/{k}"]
            if torch.is_tensor(v):
                assert v.dtype == torch.int64 and tuple(v.shape) == w.shape and np.array_equal(v.numpy(), w), (key, k)
            else:
                assert str(w) == v


def test_label_mask_kept_when_prompt_exceeds_model_max_length():
    """the one case in which the reference keeps the answer-only label mask it computed; the golden must actually contain such a sample"""
    z = np.load(os.path.join(HERE, "golden", "prompts.npz"))
    lab = z[case_key(0, "panoptic", ShortWordStub()) + "/labels"]
    ids = z[case_key(0, "panoptic", ShortWordStub()) + "/input_ids"]
    assert (lab != P.IGNORE_INDEX).any() and (lab == P.IGNORE_INDEX).any()
    assert np.array_equal(lab[lab != P.IGNORE_INDEX], ids[lab != P.IGNORE_INDEX])
    assert (z[case_key(0, "panoptic", WordStub()) + "/labels"] == P.IGNORE_INDEX).all()          # ... and the usual case: mismatch -> all ignored


def test_sentinels_and_slot_counts():
    tok = WordStub()
    names = ["cat", "traffic light", "background"]
    s = P.panoptic_sample(names, tok)
    ids = s["input_ids"]
    assert int((ids == IMAGE_TOKEN_INDEX).sum()) == 1 and int((ids == SEG_TOKEN_INDEX).sum()) == 1
    assert int((ids == CLS_TOKEN_INDEX).sum()) == len(names) == int(s["class_name_embedding_indices"].sum())
    assert s["cls_indices"].tolist() == sorted(s["cls_indices"].tolist()) and int(s["cls_indices"].max()) == len(names) - 1
    assert s["class_name_ids"].numel() == s["cls_indices"].numel()
    r = P.region_sample(5, tok)
    assert int((r["input_ids"] == REGION_TOKEN_INDEX).sum()) == 5
    f = P.referring_sample(["the dog", "brown dog on the left"], tok)
    assert int((f["input_ids"] == REFER_TOKEN_INDEX).sum()) == 1 == int(f["refer_embedding_indices"].sum())
    assert f["token_refer_id"][-1] == tok.encode("[SEG]")[0]
    assert P.llava_phi_prompt(["q", "a"]).startswith("[INST] <<SYS>>\n") and P.llava_phi_prompt(["q", "a"]).endswith(" a " + P.PHI_SEP)


def test_samples_feed_the_collator_and_the_splice_plan():
    """prompts -> collate -> the splice pre-pass of eval_seg (host part only): the sentinel positions the builders emit are the ones the
    splice plan consumes (class groups, the refer span, region slots)."""
    import types
    from psalm_amd.collate import DataCollatorForCOCODatasetV2
    tok = WordStub()
    items = []
    for n in (3, 3):
        s = P.panoptic_sample([f"class {i}" for i in range(n)], tok)
        s["image"] = torch.zeros(3, 8, 8)
        items.append(s)
    batch = DataCollatorForCOCODatasetV2(tokenizer=types.SimpleNamespace(pad_token_id=tok.pad_token_id, model_max_length=tok.model_max_length))(items)
    assert batch["input_ids"].shape[0] == 2 and int((batch["input_ids"] == CLS_TOKEN_INDEX).sum()) == 6
    assert torch.equal(batch["class_name_embedding_indices"], (batch["input_ids"] == CLS_TOKEN_INDEX).long())
