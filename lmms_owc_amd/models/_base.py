"""`Model` ABC the engine talks to (mirror of /root/reference/src/models/_base.py:56-338).

Differences, all deliberate: no accelerate / bitsandbytes dependency (one process per GPU is set up with
plain torch.distributed, backend "nccl" == RCCL on ROCm); `batch_size != 1` is allowed (the reference
raises at `_base.py:103-104`; 1 stays the default); `load_in_8bit/4bit` are accepted for CLI compatibility
and rejected with a clear error (bitsandbytes has no role on this path)."""

from __future__ import annotations

import os
from abc import ABC, abstractmethod
from typing import Any

import torch


class CacheHook:
    """Kept for interface parity; the reference always disables it (`_base.py:138`)."""

    def __init__(self, cache=None) -> None:
        self.dbdict = None

    def add_partial(self, attr: str, req, res) -> None:
        return None


class DistShim:
    """The few `accelerator.*` members the reference engine touches (`_engine.py:84-85,167-169,201,387`)."""

    def __init__(self) -> None:
        import torch.distributed as dist

        self._dist = dist if dist.is_available() and dist.is_initialized() else None
        self.num_processes = self._dist.get_world_size() if self._dist else 1
        self.process_index = self._dist.get_rank() if self._dist else 0
        self.local_process_index = int(os.environ.get("LOCAL_RANK", "0")) if self._dist else 0
        self.is_main_process = self.process_index == 0
        self.is_local_main_process = self.local_process_index == 0

    def wait_for_everyone(self) -> None:
        if self._dist:
            self._dist.barrier()

    def gather(self, tensor: torch.Tensor) -> torch.Tensor:
        if not self._dist:
            return tensor
        t = tensor.reshape(-1) if tensor.dim() else tensor.reshape(1)
        out = [torch.empty_like(t) for _ in range(self.num_processes)]
        self._dist.all_gather(out, t)
        return torch.cat(out)

    def unwrap_model(self, model):
        return model


class Model(ABC):
    _model: Any | None = None
    _processor: Any | None = None
    _tokenizer: Any | None = None

    def __init__(self, batch_size: int = 1, device_map: str = "auto", dtype: str | torch.dtype = "bfloat16",
                 load_in_8bit: bool = False, load_in_4bit: bool = False, distributed_types: list | None = None,
                 **kwargs) -> None:
        if len(kwargs) > 0:
            raise ValueError("kwargs are currently unsupported and unused in models.")
        if int(batch_size) < 1:
            raise ValueError("`batch_size` must be >= 1")
        if distributed_types is None:
            raise ValueError("`distributed_types` must be passed to the base Model constructor!")
        if load_in_8bit or load_in_4bit:
            raise ValueError("bitsandbytes quantisation is not part of the MI355X path (bf16 weights only)")
        if isinstance(dtype, str) and dtype != "auto":
            dtype = getattr(torch, dtype)
        if dtype not in (torch.bfloat16, "auto"):
            raise ValueError("the HIP path computes in bfloat16")
        self.accelerator = DistShim()
        self._rank = self.accelerator.process_index
        self._world_size = self.accelerator.num_processes
        local = self.accelerator.local_process_index
        if not torch.cuda.is_available():
            raise RuntimeError("lmms_owc_amd models run on an MI355X: no GPU visible (there is no CPU fallback)")
        self._device = torch.device("cuda", local)
        torch.cuda.set_device(self._device)
        self._device_map = f"cuda:{local}"
        self._dtype = torch.bfloat16
        self.apply_chat_template = False
        self.batch_size_per_gpu = int(batch_size)
        self.cache_hook = CacheHook(None)
        self.chat_template = None
        self.task_dict: dict = {}
        self.load_model()
        if self._model is None:
            raise ValueError("The `load_model` method must set the attribute `_model`!")

    # ---- properties used by the engine
    @property
    def batch_size(self) -> int:
        return self.batch_size_per_gpu

    @property
    def device(self) -> torch.device:
        return self._device

    @property
    def device_map(self) -> str:
        return self._device_map

    @property
    def dtype(self):
        return self._dtype

    @property
    def model(self):
        return self._model

    @property
    def processor(self):
        return self._processor

    @property
    def tokenizer(self):
        return self._tokenizer

    @property
    def tokenizer_name(self) -> str:
        return getattr(self._tokenizer, "name_or_path", "") or ""

    @property
    def eot_token_id(self) -> int:
        return -1 if self._tokenizer is None else self._tokenizer.eos_token_id

    @property
    def rank(self) -> int:
        return self._rank

    @property
    def world_size(self) -> int:
        return self._world_size

    def eval(self) -> None:
        return None

    def train(self) -> None:
        raise NotImplementedError("inference-only engine")

    @abstractmethod
    def load_model(self) -> None:
        raise NotImplementedError

    @abstractmethod
    def loglikelihood(self, requests: list) -> list[tuple[float, bool]]:
        raise NotImplementedError

    @abstractmethod
    def generate_until(self, requests: list) -> list[str]:
        raise NotImplementedError

    @abstractmethod
    def generate_until_multi_round(self, requests: list) -> list[str]:
        raise NotImplementedError


def sampling_from_gen_kwargs(gen_kwargs: dict, default_top_k: int = 50) -> dict | None:
    """The reference's generation switches -> the engine's `sampling` argument (None = greedy).

    Reference (src/models/_qwen2_vl.py:308-329, _llava_hf.py:355-376): `temperature` (default 0), `top_p` (default None),
    `num_beams` (default 1) from the request's gen_kwargs, `do_sample = temperature > 0`.  HF then samples through
    temperature -> top-k -> top-p with top_k from the checkpoint's generation_config.json (HF's own default is 50; Qwen2-VL's
    ships `top_k: 1`, which makes its sampling the argmax) - `default_top_k` carries that value.  Beam search is not built:
    num_beams != 1 raises (every task YAML of the reference is greedy, one beam).  The random stream is the library's documented
    Philox stream keyed by torch's seed (the reference seeds torch with `--seed`, eval_model.py), one stream per DOCUMENT."""
    if int(gen_kwargs.get("num_beams", 1) or 1) != 1:
        raise NotImplementedError("beam search is not implemented by the HIP decoder (num_beams must be 1; the reference's task "
                                  "configs are all greedy, one beam)")
    t = gen_kwargs.get("temperature", 0) or 0
    if not float(t) > 0:
        return None
    import torch

    return {"temperature": float(t), "top_p": gen_kwargs.get("top_p"), "top_k": int(default_top_k), "seed": int(torch.initial_seed())}

