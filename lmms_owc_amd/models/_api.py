"""Model registry (mirror of /root/reference/src/models/_api.py:18-73)."""

from __future__ import annotations

from collections.abc import Callable

from ..schema import ModelInfo

MODELS: dict[str, ModelInfo] = {}


def register_model(name: str | None = None) -> Callable:
    def decorator(model: Callable) -> Callable:
        key = name or model.__name__.lower()
        MODELS[key] = ModelInfo(name=key, builder_fn=model)
        return model

    return decorator


def get_model(model_id: str, **model_kwargs):
    return MODELS[model_id].builder_fn(**model_kwargs)


def get_model_builder(model_id: str) -> Callable | None:
    return MODELS[model_id].builder_fn


def get_model_info(model_id: str) -> ModelInfo:
    return MODELS[model_id]


def get_models_info() -> list[ModelInfo]:
    return list(MODELS.values())
