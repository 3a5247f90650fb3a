"""Hugging Face registration of the drop-in, mirroring psalm/model/language_model/llava_phi.py:34-35 and :2001-2002:

    class LlavaConfig(PhiConfig): model_type = "llava_phi"
    AutoConfig.register("llava_phi", LlavaConfig)
    AutoModelForCausalLM.register(LlavaConfig, <model class>)

so that `AutoConfig.from_pretrained(<PSALM checkpoint>)` resolves and `AutoModelForCausalLM.from_pretrained` reaches
`psalm_amd.model.PSALM.from_pretrained`.  Importing this module performs the registration (idempotent); it is imported by
`psalm_amd.builder` -- the same moment the reference registers (import of psalm.model).
"""
from __future__ import annotations

from transformers import AutoConfig, AutoModelForCausalLM, PhiConfig


class LlavaConfig(PhiConfig):                       # llava_phi.py:34-35
    model_type = "llava_phi"


def register(model_class) -> None:
    try:
        AutoConfig.register("llava_phi", LlavaConfig)
    except ValueError:                              # already registered (e.g. by the reference package in the same process)
        pass
    model_class.config_class = LlavaConfig
    try:
        AutoModelForCausalLM.register(LlavaConfig, model_class)
    except ValueError:
        pass
