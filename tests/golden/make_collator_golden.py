"""Golden for psalm_amd.collate: run the REFERENCE's DataCollatorForCOCODatasetV2 (psalm/train/train_datasets.py:968-1045) on seeded
synthetic instances and store its output.  Authoring container only (needs /root/reference + the import shims of ref_shim.py).

    python tests/golden/make_collator_golden.py"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def make_instances(seed, kind):
    """Dataset items as the reference datasets emit them (train_datasets.py:186-234, 644-695): ragged ids, sentinel ids, side tensors."""
    g = torch.Generator().manual_seed(seed)
    out = []
    n = 3
    for i in range(n):
        T = int(torch.randint(20, 60, (1,), generator=g))
        ids = torch.randint(5, 1000, (T,), generator=g)
        ids[3] = -200
        ins = {"input_ids": ids, "labels": ids.clone(), "image": torch.randn(3, 8, 8 if kind != "ragged_images" or i else 12, generator=g),
               "file_name": f"img_{seed}_{i}.jpg", "height": 480 + i, "width": 640, "dataset_type": "panoptic_coco"}
        if kind in ("panoptic", "ragged_images"):
            m = 40 if kind == "panoptic" else 30 + 5 * i                      # equal shapes -> stack; ragged -> pad with -1
            ins["class_name_ids"] = torch.randint(5, 1000, (m,), generator=g)
            ins["cls_indices"] = torch.randint(0, 9, (m,), generator=g)
            ins["class_name_embedding_indices"] = (ids == -200).long()
            ins["random_idx"] = torch.randperm(9, generator=g)
        else:
            ins["token_refer_id"] = torch.randint(5, 1000, (6 + 3 * i,), generator=g)
            ins["refer_embedding_indices"] = (torch.rand(T, generator=g) < 0.1).long()
        out.append(ins)
    return out


def main():
    import ref_shim
    ref_shim.install()
    sys.path.insert(0, "/root/reference")
    # psalm.train.llava_trainer needs transformers==4.36 internals (load_sharded_checkpoint ...) that the installed 5.x lacks; the
    # collator does not use it -- stub the one name train_datasets.py:19 imports from it
    stub = types.ModuleType("psalm.train.llava_trainer")
    stub.LLaVATrainer = object
    sys.modules["psalm.train.llava_trainer"] = stub
    from psalm.train.train_datasets import DataCollatorForCOCODatasetV2
    tok = types.SimpleNamespace(pad_token_id=50256, model_max_length=48)
    store = {}
    for kind in ("panoptic", "ragged_images", "referring"):
        batch = DataCollatorForCOCODatasetV2(tokenizer=tok)(make_instances(7, kind))
        for k, v in batch.items():
            if torch.is_tensor(v):
                store[f"{kind}/{k}"] = v.numpy()
            elif k == "images":
                for j, t in enumerate(v):
                    store[f"{kind}/images_list/{j}"] = t.numpy()
            elif k == "token_refer_id":
                for j, t in enumerate(v):
                    store[f"{kind}/token_refer_id/{j}"] = t.numpy()
            elif k == "seg_info":
                store[f"{kind}/seg_info_keys"] = np.array([",".join(sorted(d.keys())) for d in v])
            elif k == "dataset_type":
                store[f"{kind}/dataset_type"] = np.array(v)
    np.savez_compressed(os.path.join(HERE, "collator.npz"), **store)
    print(sorted(store))


if __name__ == "__main__":
    main()
