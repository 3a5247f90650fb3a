"""load_pretrained_model (reference: psalm/model/builder.py:27-72) on a tiny synthetic HF-layout checkpoint written to disk.
CPU: the kernels run through the host emulation (`ops=` extension); the GPU variant uses the product library."""
import json
import types

import pytest
import torch

from ops_backend import make_ops
from psalm_amd.builder import ImagePreprocessor, load_pretrained_model
from psalm_amd.config import PsalmConfig
from psalm_amd.synthetic import make_inputs, make_state_dict


def _write_ckpt(tmp_path, cfg, sd):
    from safetensors.torch import save_file
    items = sorted(sd.items())
    half = len(items) // 2                                  # two shards, like a large HF checkpoint
    save_file({k: v.contiguous() for k, v in items[:half]}, str(tmp_path / "model-00001-of-00002.safetensors"))
    save_file({k: v.contiguous() for k, v in items[half:]}, str(tmp_path / "model-00002-of-00002.safetensors"))
    (tmp_path / "config.json").write_text(json.dumps({
        "model_type": "llava_phi", "vocab_size": cfg.vocab_size, "hidden_size": cfg.hidden_size,
        "intermediate_size": cfg.intermediate_size, "num_hidden_layers": cfg.num_layers, "num_attention_heads": cfg.num_heads,
        "partial_rotary_factor": 0.5, "projector_outdim": cfg.proj_planes, "swin_type": "base", "max_sequence_length": 1536}))


def _mask_yaml(tmp_path, cfg):
    (tmp_path / "base.yaml").write_text("MODEL:\n  MASK_FORMER:\n    NHEADS: %d\n    DIM_FEEDFORWARD: %d\n" % (cfg.md_heads, cfg.md_dim_ff))
    (tmp_path / "mask.yaml").write_text(
        "_BASE_: base.yaml\nMODEL:\n  SEM_SEG_HEAD:\n    TRANSFORMER_ENC_LAYERS: %d\n    MASK_DIM: %d\n"
        "  MASK_FORMER:\n    HIDDEN_DIM: %d\n    NUM_OBJECT_QUERIES: %d\n    DEC_LAYERS: %d\n"
        "  SWIN:\n    EMBED_DIM: %d\n    DEPTHS: [2, 2, 2, 2]\n    NUM_HEADS: [1, 2, 4, 8]\nINPUT:\n  IMAGE_SIZE: 96\n"
        % (cfg.md_enc_layers, cfg.md_mask_dim, cfg.md_hidden, cfg.md_queries, cfg.md_dec_layers + 1, cfg.swin_embed_dim))
    return str(tmp_path / "mask.yaml")


@pytest.mark.parametrize("kind", ["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def test_load_pretrained_model_roundtrip(tmp_path, kind):
    if kind == "hip" and not torch.cuda.is_available():
        pytest.skip("no GPU")
    ops = make_ops(kind)
    cfg = PsalmConfig.tiny("panoptic")
    sd = make_state_dict(cfg, seed=3)
    _write_ckpt(tmp_path, cfg, sd)
    args = types.SimpleNamespace(model_map_name="psalm", seg_task="panoptic")
    tok, model, proc, ctx = load_pretrained_model(str(tmp_path), None, "psalm", args, mask_config=_mask_yaml(tmp_path, cfg),
                                                  precision="fp32", use_graphs=False, ops=ops)
    assert tok is None and ctx == 1536 and set(proc) == {"panoptic", "instance", "semantic"}
    assert model.seg_task == "panoptic" and model.cfg.md_heads == cfg.md_heads and model.cfg.md_dim_ff == cfg.md_dim_ff
    # the tiny architecture differs from the dataclass defaults in fields config.json / the YAML do not carry
    model.cfg = cfg
    from psalm_amd.model import PSALM
    direct = PSALM(cfg, sd, ops=ops, precision="fp32")
    inputs = make_inputs(cfg, "panoptic", size=96, batch=1, seed=4, num_classes=9)
    a = PSALM(cfg, {k: v for k, v in __import__("psalm_amd.builder", fromlist=["read_checkpoint"]).read_checkpoint(str(tmp_path)).items()},
              ops=ops, precision="fp32").eval_seg(**inputs)[0]
    b = direct.eval_seg(**inputs)[0]
    assert torch.equal(a["mask_pred"], b["mask_pred"]) and torch.equal(a["sem_seg"], b["sem_seg"])
    with pytest.raises(ValueError):
        load_pretrained_model(str(tmp_path), None, "psalm", types.SimpleNamespace(model_map_name="llava"), ops=ops)
    _, mv, _, _ = load_pretrained_model(str(tmp_path), None, "psalm", types.SimpleNamespace(model_map_name="psalm_video", seg_task="region"),
                                        mask_config=_mask_yaml(tmp_path, cfg), precision="fp32", use_graphs=False, ops=ops)
    assert mv.seg_task == "region" and callable(mv.eval_video)                           # builder.py:45-49: the DAVIS evaluation class
    with pytest.raises(NotImplementedError):
        load_pretrained_model(str(tmp_path), None, "psalm", args, load_4bit=True, ops=ops)


def test_image_preprocessor_contract():
    p = ImagePreprocessor(64, [123.675, 116.28, 103.53], [58.395, 57.12, 57.375])
    img = (torch.arange(40 * 30 * 3) % 251).to(torch.uint8).view(40, 30, 3)
    d = p(img)
    assert d["image"].shape == (3, 64, 64) and d["padding_mask"].shape == (64, 64)
    assert (d["height"], d["width"]) == (40, 30)
    assert not d["padding_mask"][:64, :48].any() and d["padding_mask"][:, 48:].all()      # 40x30 -> 64x48, right side padded
    pad_val = (128.0 - 123.675) / 58.395
    assert torch.allclose(d["image"][0, :, 50], torch.full((64,), pad_val), atol=1e-5)
