"""Seam B2 as the reference's evaluation script uses it: the LITERAL call sequence of psalm/eval/panoptic_segmentation.py:14-21,96-143
(import path, positional/keyword arguments, `.to(dtype=, device=).eval()`, the eval_seg keywords, the output keys the two evaluators
read: panoptic_evaluation.py:114-145,179-222) against the drop-in.  CPU: the kernels run in the host emulator (the only patch is
which library `get_ops()` returns); `-m gpu`: the product library on the MI355X."""
import json
import types

import numpy as np
import pytest
import torch

from ops_backend import make_ops
from psalm_amd.config import PsalmConfig
from psalm_amd.synthetic import make_inputs, make_state_dict
from test_7_builder import _mask_yaml, _write_ckpt


@pytest.mark.parametrize("kind", ["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def test_reference_eval_script_call_sequence(tmp_path, monkeypatch, kind):
    if kind == "hip" and not torch.cuda.is_available():
        pytest.skip("no GPU")
    import psalm_amd.hip_ops as hip_ops
    ops = make_ops(kind)                              # resolved BEFORE get_ops is replaced (make_ops("hip") calls the real one)
    monkeypatch.setattr(hip_ops, "get_ops", lambda: ops)
    cfg = PsalmConfig.tiny("panoptic")
    sd = make_state_dict(cfg, seed=3)
    _write_ckpt(tmp_path, cfg, sd)
    (tmp_path / "config.json").write_text(json.dumps(dict(json.loads((tmp_path / "config.json").read_text()),
                                                            region_points=cfg.region_points)))
    from psalm_amd import dropin
    dropin.install()

    # ---- psalm/eval/panoptic_segmentation.py:14-21
    from psalm.model.builder import load_pretrained_model
    from psalm.model.language_model.llava_phi import PSALM, LlavaConfig
    from transformers import AutoConfig
    assert AutoConfig.from_pretrained(str(tmp_path)).__class__ is LlavaConfig            # llava_phi.py:2001

    data_args = types.SimpleNamespace(model_path=str(tmp_path), mask_config=_mask_yaml(tmp_path, cfg), model_map_name="psalm",
                                      seg_task="panoptic", version="llava_phi")
    model_path, model_name = data_args.model_path, "psalm"
    # ---- :96
    tokenizer, model, image_processor, context_len = load_pretrained_model(model_path, None, model_name, mask_config=data_args.mask_config,
                                                                           model_args=data_args)
    data_args.image_processor = image_processor
    assert isinstance(model, PSALM) and set(image_processor) == {"panoptic", "instance", "semantic"} and context_len == 1536
    assert model.get_vision_tower().image_processor is image_processor and model.config.model_type == "llava_phi"
    model.cfg = cfg                                   # (tiny test architecture: fields neither config.json nor the YAML carry)
    model.__init__(cfg, sd, ops=model.ops, precision=model.precision, use_graphs=False)
    batch = make_inputs(cfg, "panoptic", size=96, batch=1, seed=4, num_classes=9)
    is_thing = batch.pop("is_thing_list")
    eval_dataloader = [batch]
    seen = []
    # ---- :126-143
    device = 'cuda' if torch.cuda.is_available() and kind == "hip" else 'cpu'
    model.to(dtype=torch.float32, device=device).eval()
    with torch.no_grad():
        for idx, inputs in enumerate(eval_dataloader):
            inputs = {k: v.to(device) if torch.is_tensor(v) else v for k, v in inputs.items()}
            outputs = model.eval_seg(
                input_ids=inputs['input_ids'],
                attention_mask=inputs['attention_mask'],
                images=inputs['images'].float(),
                seg_info=inputs['seg_info'],
                class_name_embedding_indices=inputs['class_name_embedding_indices'],
                class_name_ids=inputs['class_name_ids'],
                cls_indices=inputs['cls_indices'],
                labels=inputs['labels'],
                is_thing_list=is_thing
            )
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            # what evaluator.process / sem_evaluator.process read (panoptic_evaluation.py:179-186, :124-126)
            for input, output in zip(inputs['seg_info'], outputs):
                panoptic_img, segments_info = output["panoptic_seg"]
                panoptic_img = panoptic_img.cpu().numpy()
                pred = np.array(output["sem_seg"].argmax(dim=0).to("cpu"), dtype=int)
                seen.append((panoptic_img.shape, pred.shape, [dict(s) for s in segments_info], len(output["instances"])))
    (pan_shape, sem_shape, segs, n_inst), = seen
    assert pan_shape == sem_shape == (96, 96) and all({"id", "isthing", "category_id"} <= set(s) for s in segs)
    # same numbers as the directly constructed model
    from oracle import psalm_oracle as O
    torch.manual_seed(0)
    want = O.eval_seg(sd, cfg, is_thing_list=is_thing, **batch)[0]
    assert (torch.as_tensor(pred) == want["sem_seg"].argmax(0)).float().mean() > 0.99


def test_to_rejects_a_device_the_kernels_do_not_run_on():
    from psalm_amd.hip_ops import PsalmHipError
    from psalm_amd.model import PSALM
    cfg = PsalmConfig.tiny("panoptic")
    m = PSALM(cfg, make_state_dict(cfg, seed=3), ops=make_ops("emu"), precision="fp32")
    assert m.to(torch.float32) is m and m.eval() is m and m.float() is m and m.get_model() is m
    with pytest.raises(PsalmHipError):
        m.to(device="cuda")                          # emulator-backed model lives on the host; a product model refuses "cpu" the same way
    with pytest.raises(NotImplementedError):
        m.train()


def _run_py(code, env_extra, cwd):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(cwd), root, os.path.join(root, "tests")]), **env_extra)
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=str(cwd), timeout=600)


def test_dropin_env_hook_and_reference_present_branch(tmp_path):
    """(1) PSALM_AMD_DROPIN=1 + `import psalm_amd` installs the drop-in; (2) with a reference `psalm` package importable, the
    parents are shell packages over the reference's directories: its psalm/model/__init__.py (which imports the CUDA model) never runs,
    sibling modules still come from the reference, the replaced entry points are ours."""
    ref = tmp_path / "psalm"
    (ref / "model" / "language_model").mkdir(parents=True)
    (ref / "__init__.py").write_text("")
    (ref / "model" / "__init__.py").write_text("raise ImportError('reference psalm/model/__init__.py executed (would import the CUDA model)')\n")
    (ref / "model" / "language_model" / "__init__.py").write_text("")
    (ref / "model" / "language_model" / "llava_phi.py").write_text("raise ImportError('reference llava_phi imported')\n")
    (ref / "model" / "sibling.py").write_text("X = 41\n")
    (ref / "constants.py").write_text("IMAGE_TOKEN_INDEX = -200\n")
    code = (
        "import psalm_amd, sys\n"
        "import psalm.model\n"
        "from psalm.model.sibling import X\n"
        "from psalm.constants import IMAGE_TOKEN_INDEX\n"
        "from psalm.model.builder import load_pretrained_model\n"
        "from psalm.model.language_model.llava_phi import PSALM, LlavaConfig\n"
        "from psalm.model import PSALM as P2\n"
        "import psalm_amd.builder, psalm_amd.model\n"
        "assert load_pretrained_model is psalm_amd.builder.load_pretrained_model and PSALM is psalm_amd.model.PSALM is P2\n"
        "assert X == 41 and IMAGE_TOKEN_INDEX == -200\n"
        "print('OK', psalm.model.__path__[0])\n")
    r = _run_py(code, {"PSALM_AMD_DROPIN": "1"}, tmp_path)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]
    assert str(ref / "model") in r.stdout
    # without the variable nothing is installed: the reference's own (here: raising) package init is what an import finds
    r = _run_py("import psalm_amd\nimport psalm.model\n", {"PSALM_AMD_DROPIN": "0"}, tmp_path)
    assert r.returncode != 0 and "reference psalm/model/__init__.py executed" in r.stderr
