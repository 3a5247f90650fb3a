"""`DataCollatorForCOCODatasetV2` — the step immediately in front of `eval_seg`: the reference's batch collator
(psalm/train/train_datasets.py:968-1045), same class name, constructor and output dict, so an evaluation DataLoader can switch to it
unchanged:

    input_ids / labels            right-padded with tokenizer.pad_token_id / IGNORE_INDEX, truncated to tokenizer.model_max_length
    attention_mask                input_ids != pad_token_id
    images                        stacked when every image has the same shape, else the list
    seg_info                      the instances themselves, with `input_ids`, `labels`, `image` REMOVED (the reference deletes them in place)
    class_name_ids / cls_indices  stacked when shapes agree, else right-padded with -1
    class_name_embedding_indices / refer_embedding_indices   right-padded with 0
    token_refer_id                list passthrough;  random_idx stacked;  dataset_type list

Host-side integer work (no kernel): one pre-sized buffer per key instead of `pad_sequence`'s per-row copies.  Pinned by a golden
produced by the reference's own class (tests/golden/make_collator_golden.py -> tests/golden/collator.npz)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Sequence

import torch

from .config import IGNORE_INDEX


def _pad_rows(rows, value):
    """torch.nn.utils.rnn.pad_sequence(rows, batch_first=True, padding_value=value) for 1-D rows, in one allocation."""
    n = max(int(r.shape[0]) for r in rows)
    out = torch.full((len(rows), n), value, dtype=rows[0].dtype)
    for i, r in enumerate(rows):
        out[i, : r.shape[0]] = r
    return out


def _stack_or_pad(rows, value):
    if any(r.shape != rows[0].shape for r in rows):
        return _pad_rows(rows, value)
    return torch.stack(rows, dim=0)


@dataclass
class DataCollatorForCOCODatasetV2(object):
    """Collate examples for segmentation inference (train_datasets.py:968-1045)."""

    tokenizer: object                     # needs .pad_token_id and .model_max_length

    def __call__(self, instances: Sequence[Dict]) -> Dict[str, torch.Tensor]:
        pad_id, max_len = self.tokenizer.pad_token_id, self.tokenizer.model_max_length
        input_ids = _pad_rows([ins["input_ids"] for ins in instances], pad_id)[:, :max_len]
        labels = _pad_rows([ins["labels"] for ins in instances], IGNORE_INDEX)[:, :max_len]
        batch = dict(input_ids=input_ids, labels=labels, attention_mask=input_ids.ne(pad_id))
        if "image" in instances[0]:
            images = [ins["image"] for ins in instances]
            if all(x is not None and x.shape == images[0].shape for x in images):
                batch["images"] = torch.stack(images)
            else:
                batch["images"] = images
        for ins in instances:
            for key in ("input_ids", "labels", "image"):
                del ins[key]
        batch["seg_info"] = [ins for ins in instances]
        first = instances[0]
        if "dataset_type" in first:
            batch["dataset_type"] = [ins["dataset_type"] for ins in instances]
        if "class_name_ids" in first:
            batch["class_name_ids"] = _stack_or_pad([ins["class_name_ids"] for ins in instances], -1)
        if "token_refer_id" in first:
            batch["token_refer_id"] = [ins["token_refer_id"] for ins in instances]
        if "cls_indices" in first:
            batch["cls_indices"] = _stack_or_pad([ins["cls_indices"] for ins in instances], -1)
        if "random_idx" in first:
            batch["random_idx"] = torch.stack([ins["random_idx"] for ins in instances], dim=0)
        if "class_name_embedding_indices" in first:
            batch["class_name_embedding_indices"] = _pad_rows([ins["class_name_embedding_indices"] for ins in instances], 0)
        if "refer_embedding_indices" in first:
            batch["refer_embedding_indices"] = _pad_rows([ins["refer_embedding_indices"] for ins in instances], 0)
        return batch
