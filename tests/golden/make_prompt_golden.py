"""Golden for psalm_amd.prompts: run the REFERENCE's own dataset classes (psalm/train/train_datasets.py: COCO_panoptic_dataset :43-234,
COCO_semantic_dataset :565-615, COCO_instance_dataset :356-486, COCO_interactive_dataset :236-354, RefCOCO_dataset :617-695 -- their
`__getitem__`, `preprocess_llama2`, `tokenizer_special_tokens`, `preprocess_class_name`, `preprocess_referring_instruction`, with
`conversation_lib.default_conversation = conv_templates['llava_phi']` as every eval script sets it) on stub records and store the text fields
they emit.  The datasets are built without their `__init__` (no annotation files here): only the attributes `__getitem__` reads are set, the
image processor is a stand-in that returns the record (images are not this golden's subject).  Authoring container only.

Two stub tokenizers (no vocabulary files here, and none needed: the subject is the splice / template / label arithmetic AROUND the
tokenizer): `WordStub` -- words and punctuation runs hashed to ids, so that encode(a + b) == encode(a) + encode(b) at word boundaries
(the reference's label arithmetic adds up) -- and `CharStub` -- one id per character plus a leading marker id per call (pieces do NOT add
up).  No stub emits a begin-of-text token, like the Phi tokenizer under add_special_tokens=False -- so the reference's `cur_len = 1`
start makes EVERY sample a `tokenization mismatch` (all labels IGNORE_INDEX; inference never reads them); `ShortWordStub` has a
model_max_length below the prompt lengths, the one case in which the reference keeps the mask it computed.

    python tests/golden/make_prompt_golden.py"""
import os
import re
import sys
import types
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


class WordStub:
    pad_token_id = 50256
    model_max_length = 2048

    def encode(self, text, add_special_tokens=False):
        return [5 + zlib.crc32(w.encode()) % 50000 for w in re.findall(r"\s*[A-Za-z0-9]+|\s*[^A-Za-z0-9\s]+|\s+", text)]


class ShortWordStub(WordStub):
    """model_max_length below every prompt length: the reference then KEEPS the label mask it computed (no mismatch fall-back)"""
    model_max_length = 8


class CharStub:
    pad_token_id = 0
    model_max_length = 4096

    def encode(self, text, add_special_tokens=False):
        return ([3] if text else []) + [10 + ord(c) % 500 for c in text]


COCO_THINGS = ["person", "bicycle", "car", "traffic light", "hot dog", "teddy bear"]
SENTENCES = [["the left zebra"], ["man in a red shirt", "guy on the right holding a cup"], ["a"]]
CASES = [("panoptic", COCO_THINGS + ["road", "sky-other-merged", "background"]), ("semantic", COCO_THINGS + ["wall-other-merged", "background"]),
         ("instance", COCO_THINGS + ["background"]), ("region", 1), ("region", 4), ("referring", SENTENCES[0]), ("referring", SENTENCES[1]),
         ("referring", SENTENCES[2])]


def case_key(i, task, tok):
    return f"{i}_{task}_{type(tok).__name__}"


def main():
    import ref_shim
    ref_shim.install()
    sys.path.insert(0, "/root/reference")
    stub = types.ModuleType("psalm.train.llava_trainer")       # (needs transformers 4.36 internals; the datasets do not use it)
    stub.LLaVATrainer = object
    sys.modules["psalm.train.llava_trainer"] = stub
    from psalm import conversation as conversation_lib
    from psalm.train import train_datasets as TD
    conversation_lib.default_conversation = conversation_lib.conv_templates["llava_phi"]      # psalm/eval/panoptic_segmentation.py:100

    class Processor:                                           # stands in for the dataset mappers: the record goes through untouched
        def preprocess(self, d, mask_format=None, region_mask_type=None):
            return d

    def build(cls, tok, **attrs):
        ds = object.__new__(cls)
        ds.tokenizer = tok
        ds.mask_format = "polygon"
        ds.data_args = types.SimpleNamespace(image_processor=Processor(), image_folder="/img", refcoco_image_folder="/img", region_mask_type=None)
        for k, v in attrs.items():
            setattr(ds, k, v)
        return ds

    store = {}
    for tok in (WordStub(), ShortWordStub(), CharStub()):
        for i, (task, arg) in enumerate(CASES):
            if task in ("panoptic", "semantic"):
                cls = TD.COCO_panoptic_dataset if task == "panoptic" else TD.COCO_semantic_dataset
                ds = build(cls, tok, coco_class_name=list(arg), coco_id_to_cont_id={}, panoptic_image_path="/i", panoptic_gt_path="/p",
                           semantic_gt_path="/s", data=[{"image_id": "7", "file_name": "000000000007.png", "segments_info": []}])
            elif task == "instance":
                ds = build(TD.COCO_instance_dataset, tok, coco_class_name=list(arg), coco_id_to_cont_id={},
                           data=[{"image": "a.jpg", "image_info": {"height": 4, "width": 6}, "new_img_id": 1, "anns": []}])
            elif task == "region":
                class RegionProcessor(Processor):
                    def preprocess(self, d, mask_format=None, region_mask_type=None, _n=arg):
                        d["instances"] = list(range(_n))
                        return d
                ds = build(TD.COCO_interactive_dataset, tok, coco_class_name=[], coco_id_to_cont_id={},
                           data=[{"image": "a.jpg", "image_info": {"height": 4, "width": 6}, "new_img_id": 1, "anns": []}])
                ds.data_args.image_processor = RegionProcessor()
            else:
                ds = build(TD.RefCOCO_dataset, tok, coco_class_name=[], coco_id_to_cont_id={},
                           data=[{"image_info": {"file_name": "a.jpg", "height": 4, "width": 6}, "new_img_id": 1, "anns": [],
                                  "instruction": [{"sent": s} for s in arg]}])
            item = ds[0]
            for k in ("input_ids", "labels", "class_name_ids", "cls_indices", "class_name_embedding_indices", "token_refer_id",
                      "refer_embedding_indices"):
                if k in item:
                    store[f"{case_key(i, task, tok)}/{k}"] = item[k].numpy()
            if "dataset_type" in item:
                store[f"{case_key(i, task, tok)}/dataset_type"] = np.array(item["dataset_type"])
    out = os.path.join(HERE, "prompts.npz")
    np.savez_compressed(out, **store)
    print(f"{len(store)} arrays -> {out}")


if __name__ == "__main__":
    main()
