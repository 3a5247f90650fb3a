"""Generate the committed golden vectors by running the REFERENCE code itself on CPU.

Runs only in the authoring container (needs /root/reference; see ref_shim.py).  For each case it
  1. builds the seeded synthetic checkpoint (psalm_amd.synthetic.make_state_dict) and loads it into
     the reference `PSALM` with load_state_dict(strict=True) -- which also proves the checkpoint
     layout of the drop-in matches the reference's;
  2. runs the reference's own `PSALM.eval_seg` (fp32, CPU, MSDA through the reference's
     grid_sample fallback) on the seeded synthetic inputs (psalm_amd.synthetic.make_inputs);
  3. stores compact signatures of every stage-boundary tensor plus the small outputs in
     tests/golden/<case>.npz.

    python tests/golden/make_golden.py [case ...]

The fixtures are what pins oracle/psalm_oracle.py (tests/test_4_oracle_golden.py) and, on the GPU
box where /root/reference does not exist, the HIP path (tests/test_9_e2e_gpu.py).
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ref_shim  # noqa: E402
from psalm_amd.config import PsalmConfig  # noqa: E402
from psalm_amd.synthetic import make_inputs, make_state_dict  # noqa: E402

CASES = {
    # BASELINE.json configs[0]: single 512x512 panoptic forward on the reference CPU path, full model
    "panoptic_512": dict(task="panoptic", size=512, batch=1, layers=24, seed=0, pad=0),
    # ragged referring batch (exercises LP:874-948 padding) with a 2-layer LLM, non-zero padding_mask
    "referring_384_b2": dict(task="referring", size=384, batch=2, layers=2, seed=1, pad=32),
    # region / interactive prompt, k=1 region; RNG-dependent point sampling (CC:31-40)
    "region_384": dict(task="region", size=384, batch=1, layers=2, seed=2, pad=0),
    # the two remaining seg_task variants of eval_seg (LP:268-301): semantic (post-processing AFTER inference, so the padded
    # crop + resize acts on the 133-plane semantic map) and instance (top-k without the thing filter)
    "semantic_384": dict(task="semantic", size=384, batch=1, layers=2, seed=3, pad=32),
    "instance_384": dict(task="instance", size=384, batch=1, layers=2, seed=4, pad=32),
    # PSALMForDAVISEval.eval_video (LP:1845-1998): region prompts pooled from the previous frame (vp_images, vp_region_masks)
    "video_region_384": dict(task="region", size=384, batch=1, layers=2, seed=5, pad=0, video=True),
    # BASELINE.json configs[1] with what the reference's evaluation loop really feeds (r05): the full 24-layer model at
    # 1024 x 1024, a 480 x 640 original -> un-padded box 768 x 1024 inside the canvas (coco_panoptic_mapper.py:81-89), results cropped to the box
    # and resized to 480 x 640 (LP:1418-1429).  Mask logits kept at stride 8 (410 kB), label maps in full.
    "panoptic_1024_box": dict(task="panoptic", size=1024, batch=1, layers=24, seed=0, pad=0, geometry=(768, 1024, 480, 640), mask_stride=8),
}
RNG_SEED_AT_CALL = 1234


def case_config(c) -> PsalmConfig:
    return PsalmConfig(num_layers=c["layers"], seg_task=c["task"])


def signature(t: torch.Tensor, n=256):
    """shape, moments and n values at fixed pseudo-random flat positions."""
    t = t.detach().float().contiguous().view(-1)
    g = torch.Generator().manual_seed(t.numel() % 100003)
    idx = torch.randint(0, t.numel(), (n,), generator=g)
    return {"numel": np.int64(t.numel()), "mean": np.float64(t.double().mean()), "std": np.float64(t.double().std()),
            "absmax": np.float64(t.abs().max()), "idx": idx.numpy(), "val": t[idx].numpy()}


def build_reference(cfg: PsalmConfig, sd, video=False):
    PSALM, LlavaConfig = ref_shim.reference_classes()
    if video:
        from psalm.model.language_model.llava_phi import PSALMForDAVISEval as PSALM  # type: ignore
    mask_cfg = ref_shim.load_reference_mask_cfg(cfg.seg_task)
    hf = LlavaConfig(vocab_size=cfg.vocab_size + 2, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                     num_hidden_layers=cfg.num_layers, num_attention_heads=cfg.num_heads,
                     partial_rotary_factor=cfg.partial_rotary_factor, max_position_embeddings=cfg.max_position_embeddings)
    hf.mm_vision_tower = "swin"
    hf.mm_projector_type = "swin_conv"
    hf.projector_outdim = cfg.hidden_size
    hf.mm_input_embeds = 1024
    hf.mask_decode_train = True
    model = PSALM(hf, mask_decoder_cfg=mask_cfg)
    missing = model.load_state_dict(sd, strict=True)
    print("load_state_dict(strict=True):", missing)
    model = model.to(torch.float32).eval()
    return model


def run_case(name, c):
    cfg = case_config(c)
    t0 = time.time()
    sd = make_state_dict(cfg, seed=c["seed"], include_lm_head=True)
    print(f"[{name}] synthetic state dict: {len(sd)} tensors, {sum(v.numel() for v in sd.values())/1e9:.2f} B params, {time.time()-t0:.1f}s")
    model = build_reference(cfg, sd, video=c.get("video", False))
    inputs = make_inputs(cfg, task=c["task"], size=c["size"], batch=c["batch"], seed=c["seed"], pad=c["pad"], video=c.get("video", False),
                         **({"geometry": [c["geometry"]]} if c.get("geometry") else {}))

    stages = {}
    hooks = []
    hooks.append(model.model.vision_tower.register_forward_hook(
        lambda m, i, o: stages.update(res2=o[0], res3=o[1], res4=o[2], res5=o[3])))
    hooks.append(model.model.mm_projector.register_forward_hook(lambda m, i, o: stages.update(image_tokens=o)))
    hooks.append(model.model.register_forward_hook(lambda m, i, o: stages.update(hidden_states=o.last_hidden_state)))
    hooks.append(model.predictor.register_forward_hook(lambda m, i, o: stages.update(
        pred_masks=o["pred_masks"], pred_class_name_logits=o["pred_class_name_logits"],
        pred_SEG_logits=o["pred_SEG_logits"], pred_region_logits=o["pred_region_logits"])))
    ff = model.pixel_decoder.forward_features

    def ff_wrap(features):
        mf, enc, ms = ff(features)
        stages.update(mask_features=mf, ms0=ms[0], ms1=ms[1], ms2=ms[2])
        return mf, enc, ms
    model.pixel_decoder.forward_features = ff_wrap

    torch.manual_seed(RNG_SEED_AT_CALL)
    t0 = time.time()
    with torch.no_grad():
        out = model.eval_video(**inputs) if c.get("video") else model.eval_seg(**inputs)
    dt = time.time() - t0
    print(f"[{name}] reference eval_seg: {dt:.2f}s, returned {len(out)} result(s) (reference stops after image 0, LP:1472)")
    for h in hooks:
        h.remove()

    save = {"meta_case": np.array(repr(c)), "ref_seconds": np.float64(dt), "threads": np.int64(torch.get_num_threads())}
    for k, v in stages.items():
        if v is None:
            continue
        if isinstance(v, list):
            v = torch.cat([x.reshape(-1) for x in v])
        for kk, vv in signature(v).items():
            save[f"sig_{k}_{kk}"] = vv
    if stages.get("pred_class_name_logits") is not None:
        save["pred_class_name_logits"] = stages["pred_class_name_logits"].numpy()
    if stages.get("pred_SEG_logits") is not None:
        save["pred_SEG_logits"] = stages["pred_SEG_logits"].numpy()
    if stages.get("pred_region_logits") is not None:
        save["pred_region_logits"] = torch.cat([x.reshape(-1) for x in stages["pred_region_logits"]]).numpy()
    st_ = int(c.get("mask_stride", 4))
    save[f"pred_masks_s{st_}"] = stages["pred_masks"][:, :, ::st_, ::st_].contiguous().numpy()
    save["pred_masks_pos_frac"] = (stages["pred_masks"] > 0).float().mean((2, 3)).numpy()
    r = out[0]
    if "sem_seg" in r:
        save["sem_seg_argmax"] = r["sem_seg"].argmax(0).to(torch.uint8).numpy()
        for kk, vv in signature(r["sem_seg"]).items():
            save[f"sig_sem_seg_{kk}"] = vv
    if "panoptic_seg" in r:
        pan, info = r["panoptic_seg"]
        save["panoptic_ids"] = pan.to(torch.uint8).numpy()
        save["panoptic_info"] = np.array([[s["id"], int(s["isthing"]), s["category_id"]] for s in info], dtype=np.int64).reshape(-1, 3)
        print(f"[{name}] panoptic segments: {len(info)}")
    if "instances" in r:
        inst = r["instances"]
        save["inst_scores"] = inst.scores.numpy()
        if inst.has("pred_classes"):
            save["inst_classes"] = inst.pred_classes.numpy()
        save["inst_mask_area"] = inst.pred_masks.flatten(1).sum(1).numpy()
        save["inst_masks_s4"] = np.packbits(inst.pred_masks[:, ::4, ::4].numpy().astype(bool), axis=-1)
        print(f"[{name}] instances: {inst.pred_masks.shape}")
    if "gt" in r:
        for kk, vv in signature(r["gt"]).items():
            save[f"sig_gt_{kk}"] = vv
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **save)
    print(f"[{name}] wrote {path} ({os.path.getsize(path)/1e3:.0f} kB)")


if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    for n in names:
        run_case(n, CASES[n])
