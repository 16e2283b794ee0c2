"""Model registry of the MI355X engine (mirror of /root/reference/src/models/__init__.py)."""

from collections.abc import Callable

from ._api import MODELS, get_model, get_model_builder, get_model_info, get_models_info, register_model
from ._base import Model
from ._llava_hf import LLaVA
from ._qwen2_vl import Qwen2VL

__all__ = ["MODELS", "Model", "LLaVA", "Qwen2VL", "register_model", "get_model", "get_model_builder", "get_model_info",
           "get_models_info"]

# same keys as the reference's `custom-model` map (src/models/__init__.py) for the families on this path
MODEL_TYPES: dict[str, Callable] = {"llava": LLaVA, "qwen2-vl": Qwen2VL}


@register_model("custom-model")
def custom_model(model_type: str, model_name_or_path: str, **model_kwargs):
    model_cls = MODEL_TYPES.get(model_type)
    if model_cls is None:
        raise ValueError(f"Model type '{model_type}' not found.")
    return model_cls(model_name_or_path, **model_kwargs)
