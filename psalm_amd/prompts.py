"""Text side of one `eval_seg` sample: what the reference's dataset classes hand the collator next to the image --
`input_ids` with the sentinel ids of the splice (`<image>`, `<cls>`, `<seg>`, `<region>`, `<refer>`), `labels`, the class-name / referring
token strings and their 0/1 position maps.  Host string / integer work; the tokenizer is the caller's (any object with
`encode(text, add_special_tokens=False) -> list[int]`, `pad_token_id`, `model_max_length`), so the results equal the reference's for the
tokenizer it is run with.  Pinned to the reference's own methods by tests/golden/make_prompt_golden.py -> tests/test_8_prompts.py.

Reference (psalm/train/train_datasets.py unless noted):
  tokenizer_special_tokens :156-173 (:626-643 adds `<refer>`)      -> encode_with_sentinels
  conv_llava_phi + Conversation.get_prompt, LLAMA_2 style          -> llava_phi_prompt          (psalm/conversation.py:71-89, 374-385)
  preprocess_llama2 :91-154                                        -> conversation_ids_and_labels
  preprocess_class_name :175-184                                   -> class_name_tokens
  preprocess_referring_instruction :619-624, instruction join :680-682   -> referring_tokens
  the prompt texts of COCO_panoptic_dataset :210-220, COCO_semantic_dataset :589-599, COCO_instance_dataset :459-470,
  COCO_interactive_dataset :337-345, RefCOCO_dataset :677-684      -> *_sample()
The evaluation-only dataset classes reuse these texts: gRefcoco_Dataset (psalm/eval/eval_grefcoco.py:246-258) = referring_sample,
DAVIS_Dataset (psalm/eval/eval_davis.py:319-326) = region_sample, common_semantic_dataset for the open-vocabulary class lists
(psalm/eval/semantic_segmentation.py:339-373: the *panoptic* wording over ADE-150 / PC-459 / A-847 names) = panoptic_sample.
"""
from __future__ import annotations

import re
from typing import Dict, List, Sequence, Tuple

import torch

from .config import CLS_TOKEN_INDEX, IMAGE_TOKEN_INDEX, REFER_TOKEN_INDEX, REGION_TOKEN_INDEX, SEG_TOKEN_INDEX

IGNORE_INDEX = -100                                   # psalm/constants.py
SENTINELS = {"<image>": IMAGE_TOKEN_INDEX, "<seg>": SEG_TOKEN_INDEX, "<cls>": CLS_TOKEN_INDEX, "<region>": REGION_TOKEN_INDEX,
             "<refer>": REFER_TOKEN_INDEX}
_SPLIT = re.compile("(" + "|".join(re.escape(k) for k in SENTINELS) + ")")

# conv_llava_phi (psalm/conversation.py:374-385): the template every eval script selects (`version = 'llava_phi'`)
PHI_SYSTEM = ("You are a helpful language and vision assistant. You are able to understand the visual content that the user provides, "
              "and assist the user with a variety of tasks using natural language.")
PHI_SEP = "<|endoftext|>"

ANSWER = "\nSure, the segmentation result is <seg>"
REGION_ANSWER = "\n[SEG]<seg>"


def encode_with_sentinels(text: str, tokenizer) -> List[int]:
    """Token ids of `text` with every `<image>` / `<seg>` / `<cls>` / `<region>` / `<refer>` replaced by its negative sentinel id; the
    text between them goes through the tokenizer piece by piece, without special tokens."""
    ids: List[int] = []
    for piece in _SPLIT.split(text):
        if piece in SENTINELS:
            ids.append(SENTINELS[piece])
        else:
            ids.extend(tokenizer.encode(piece, add_special_tokens=False))
    return ids


def llava_phi_prompt(turns: Sequence[str]) -> str:
    """The conversation string of alternating user / assistant turns in the `llava_phi` template: the system text wrapped into the first
    user turn, user turns as `[INST] .. [/INST]`, every assistant turn closed by the end-of-text separator."""
    assert turns and turns[0], "first message should not be none"
    out = ""
    for i, msg in enumerate(turns):
        if not msg:
            continue
        if i == 0:
            msg = f"<<SYS>>\n{PHI_SYSTEM}\n<</SYS>>\n\n" + msg
        out += (PHI_SEP + f"[INST] {msg} [/INST]") if i % 2 == 0 else (" " + msg + " " + PHI_SEP)
    # (the reference strips the separator with str.lstrip(sep), i.e. as a character SET: every leading character that occurs in the
    #  separator goes -- the result starts with "[INST]", whose "[" is not in the set)
    return out.lstrip(PHI_SEP)


def conversation_ids_and_labels(user: str, assistant: str, tokenizer) -> Tuple[torch.Tensor, torch.Tensor]:
    """(input_ids, labels) of a one-round conversation.  Labels are the ids with everything but the assistant's answer set to
    IGNORE_INDEX, computed the way the reference computes it (lengths of re-tokenised pieces, the `- 2` of its LLaMA heritage), including
    its fall-back: when the pieces do not add up to the whole (`tokenization mismatch`), the sample's labels are all IGNORE_INDEX."""
    prompt = llava_phi_prompt([user, assistant])
    ids = torch.tensor(encode_with_sentinels(prompt, tokenizer), dtype=torch.long)
    labels = ids.clone()
    total = int(ids.ne(tokenizer.pad_token_id).sum())
    cur = 1
    labels[:cur] = IGNORE_INDEX
    marker = "[/INST] "
    for rnd in prompt.split(PHI_SEP):
        if rnd == "":
            break
        parts = rnd.split(marker)
        if len(parts) != 2:
            break
        round_len = len(encode_with_sentinels(rnd, tokenizer))
        instruction_len = len(encode_with_sentinels(parts[0] + marker, tokenizer)) - 2
        labels[cur:cur + instruction_len] = IGNORE_INDEX
        cur += round_len
    labels[cur:] = IGNORE_INDEX
    if cur < tokenizer.model_max_length and cur != total:
        labels[:] = IGNORE_INDEX
    return ids, labels


def class_name_tokens(names: Sequence[str], tokenizer, marker: str = "[SEG]") -> Tuple[torch.Tensor, torch.Tensor]:
    """(class_name_ids, cls_indices): the token ids of every class name, each followed by the FIRST token of `marker`, concatenated; and for
    every such token the index of its class (the groups `eval_seg` pools into one embedding per class, LP:552-565)."""
    tail = tokenizer.encode(marker, add_special_tokens=False)[0]
    groups = [list(tokenizer.encode(n, add_special_tokens=False)) + [tail] for n in names]
    ids = torch.tensor([t for g in groups for t in g])
    idx = torch.tensor([i for i, g in enumerate(groups) for _ in g])
    return ids, idx


def referring_tokens(sentences: Sequence[str], tokenizer, marker: str = "[SEG]") -> torch.Tensor:
    """token_refer_id: the sentences joined as ` {sentence}.` each, tokenised, followed by the first token of `marker`."""
    text = "".join(f" {s}." for s in sentences)
    return torch.tensor(list(tokenizer.encode(text, add_special_tokens=False)) + [tokenizer.encode(marker, add_special_tokens=False)[0]])


def _positions(ids: torch.Tensor, sentinel: int) -> torch.Tensor:
    m = torch.zeros_like(ids)
    m[ids == sentinel] = 1
    return m


def _class_prompt_sample(task_sentence: str, names: Sequence[str], tokenizer) -> Dict[str, torch.Tensor]:
    slots = "<cls>, " * (len(names) - 1) + "<cls>."
    ids, labels = conversation_ids_and_labels(f"This is an image <image>, Please do {task_sentence}."
                                              f"\nThis is all the candidate categories: {slots}\n", ANSWER, tokenizer)
    cids, cidx = class_name_tokens(names, tokenizer, "[SEG]")
    return {"input_ids": ids, "labels": labels, "class_name_ids": cids, "cls_indices": cidx,
            "class_name_embedding_indices": _positions(ids, CLS_TOKEN_INDEX)}


def panoptic_sample(class_names: Sequence[str], tokenizer) -> Dict[str, torch.Tensor]:
    """COCO_panoptic_dataset.__getitem__'s text fields (`class_names` as the dataset holds them: the 133 COCO names + 'background')."""
    return _class_prompt_sample("Panoptic Segmentation", class_names, tokenizer)


def instance_sample(class_names: Sequence[str], tokenizer) -> Dict[str, torch.Tensor]:
    """COCO_instance_dataset.__getitem__: the panoptic wording over the 80 thing classes + 'background'."""
    return _class_prompt_sample("Panoptic Segmentation", class_names, tokenizer)


def semantic_sample(class_names: Sequence[str], tokenizer) -> Dict[str, torch.Tensor]:
    """COCO_semantic_dataset.__getitem__."""
    return _class_prompt_sample("Semantic Segmentation", class_names, tokenizer)


def referring_sample(sentences: Sequence[str], tokenizer) -> Dict[str, torch.Tensor]:
    """RefCOCO_dataset.__getitem__: one `<refer>` slot that `eval_seg` fills with the embeddings of the sentence tokens (LP:972-978)."""
    ids, labels = conversation_ids_and_labels("This is an image <image>, Please doing Referring Segmentation according to the following "
                                              "instruction:\n<refer>", ANSWER, tokenizer)
    return {"input_ids": ids, "labels": labels, "token_refer_id": referring_tokens(sentences, tokenizer),
            "refer_embedding_indices": _positions(ids, REFER_TOKEN_INDEX), "dataset_type": "referring_coco"}


def region_sample(n_regions: int, tokenizer) -> Dict[str, torch.Tensor]:
    """COCO_interactive_dataset.__getitem__: one `<region>` slot per visual prompt (LP:302-307, 592-594 check the count)."""
    slots = " <region>," * (n_regions - 1) + " <region>."
    ids, labels = conversation_ids_and_labels("This is an image <image>, Please segment by given regions"
                                              f"\nThis is all regions: {slots}\n", REGION_ANSWER, tokenizer)
    return {"input_ids": ids, "labels": labels, "dataset_type": "region_coco"}
