"""`psalm_amd.dropin.install()` -- make the reference's import paths resolve to this package, so that an evaluation script written
against the reference (psalm/eval/panoptic_segmentation.py:14-21,96) runs with no edit besides calling `install()` first (or setting
`PSALM_AMD_DROPIN=1` and importing psalm_amd):

    from psalm.model.builder import load_pretrained_model              -> psalm_amd.builder.load_pretrained_model
    from psalm.model.language_model.llava_phi import PSALM, LlavaConfig -> psalm_amd.model.PSALM, psalm_amd.hf.LlavaConfig
    import MultiScaleDeformableAttention                                 -> the repo-root plugin module (seam B1)

Only the modules that DEFINE the replaced entry points are overridden (registered in `sys.modules` under the reference's names); every
other `psalm.*` module -- datasets, collators, evaluators, conversation templates -- is still the reference's own when its package is
on `sys.path`: the parents `psalm.model` / `psalm.model.language_model` are then registered as shell packages over the reference's
directories (same `__path__`), because the reference's own `psalm/model/__init__.py:1` imports its CUDA model (and with it flash-attn,
detectron2 ops, ...) -- which is exactly what the drop-in replaces.  Where the reference package is absent (tests, the GPU box) empty
parent packages are synthesised so the import statements themselves still work.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

_NAMES = ("psalm", "psalm.model", "psalm.model.language_model")


def _have_reference() -> bool:
    try:
        return importlib.util.find_spec("psalm") is not None and "psalm" not in _SYNTH
    except (ImportError, ValueError):
        return False


_SYNTH: set = set()


def install() -> None:
    from . import builder, hf, model
    from .config import CLS_TOKEN_INDEX, IGNORE_INDEX, IMAGE_TOKEN_INDEX, REFER_TOKEN_INDEX, REGION_TOKEN_INDEX, SEG_TOKEN_INDEX
    if not _have_reference():
        for name in _NAMES:                                   # empty parent packages
            if name not in sys.modules:
                m = types.ModuleType(name)
                m.__path__ = []                               # marks it as a package
                sys.modules[name] = m
                _SYNTH.add(name)
        for child in _NAMES[1:]:
            parent, _, leaf = child.rpartition(".")
            setattr(sys.modules[parent], leaf, sys.modules[child])
        const = types.ModuleType("psalm.constants")          # psalm/constants.py:7-12 (the sentinel-id wire format)
        const.IGNORE_INDEX, const.IMAGE_TOKEN_INDEX, const.SEG_TOKEN_INDEX = IGNORE_INDEX, IMAGE_TOKEN_INDEX, SEG_TOKEN_INDEX
        const.CLS_TOKEN_INDEX, const.REGION_TOKEN_INDEX, const.REFER_TOKEN_INDEX = CLS_TOKEN_INDEX, REGION_TOKEN_INDEX, REFER_TOKEN_INDEX
        sys.modules.setdefault("psalm.constants", const)
        sys.modules["psalm"].constants = sys.modules["psalm.constants"]
    else:
        import psalm                                           # (the reference's psalm/__init__.py is empty)
        base = os.path.dirname(os.path.abspath(psalm.__file__))
        for name in _NAMES[1:]:
            if name in sys.modules:                            # already imported by the caller: leave it
                continue
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(base, *name.split(".")[1:])]   # sibling modules (datasets_mapper, mask_decoder, ...) stay importable
            m.__package__ = name
            sys.modules[name] = m
            parent, _, leaf = name.rpartition(".")
            setattr(sys.modules[parent], leaf, m)
        if not hasattr(sys.modules["psalm.model"], "PSALM"):
            sys.modules["psalm.model"].PSALM = model.PSALM  # `from psalm.model import *` (psalm/model/__init__.py:1)
    b = types.ModuleType("psalm.model.builder")               # psalm/model/builder.py
    b.load_pretrained_model = builder.load_pretrained_model
    b.__doc__ = "psalm_amd drop-in for psalm/model/builder.py (see psalm_amd.builder)"
    sys.modules["psalm.model.builder"] = b
    lp = types.ModuleType("psalm.model.language_model.llava_phi")   # psalm/model/language_model/llava_phi.py
    lp.PSALM, lp.PSALMModel, lp.LlavaConfig = model.PSALM, model.PSALM, hf.LlavaConfig
    sys.modules["psalm.model.language_model.llava_phi"] = lp
    for parent, leaf, mod in (("psalm.model", "builder", b), ("psalm.model.language_model", "llava_phi", lp)):
        if parent in sys.modules:
            setattr(sys.modules[parent], leaf, mod)
    if "psalm.model" in _SYNTH:
        sys.modules["psalm.model"].PSALM = model.PSALM      # `from psalm.model import *` (psalm/model/__init__.py:1)
    import MultiScaleDeformableAttention  # noqa: F401  (repo root on sys.path: the B1 plugin, imported by name as the reference does)
