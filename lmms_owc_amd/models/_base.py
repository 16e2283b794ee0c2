"""`Model` ABC the engine talks to (mirror of /root/reference/src/models/_base.py:56-338).

Differences, all deliberate: no accelerate / bitsandbytes dependency (one process per GPU is set up with
plain torch.distributed, backend "nccl" == RCCL on ROCm); `batch_size != 1` is allowed (the reference
raises at `_base.py:103-104`; 1 stays the default); `load_in_8bit/4bit` are accepted for CLI compatibility
and rejected with a clear error (bitsandbytes has no role on this path)."""

from __future__ import annotations

import os
from abc import ABC, abstractmethod
from typing import Any

import numpy as np
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


class PassPipeline:
    """Host pipeline shared by the HIP model plug-ins (`generate_until`): a preparation thread readies the requests in small units
    on a PIL worker pool, the launching thread assembles prepared units into engine passes adaptively, token ids come back through
    pinned buffers one pass later, stragglers of a pass finish inside the next (greedy runs).  A plug-in provides
    `_prepare_chunk(chunk) -> dict` (host stage of one unit: at least "n", "key", "max_new", "sampling", "doc_ids", optional pinned
    "groups"), `_merge_preps(preps) -> dict`, `_launch_chunk(prep, eos, pad, carry) -> (host ids, event)`, `engine_batch(max_new)`
    and `_tokenizer`."""

    def _collate_encode(self, text: str):
        return self._tokenizer.encode(text)

    def _reserve_kv(self, first_prep: dict, slots: int, hand_over: bool) -> None:
        """Before a task's first pass: grow the engine's K / V pair ONCE to what a full pass of this task needs (the ramp's passes
        are views of it), instead of once per pass size of the ramp.  Prompt lengths are those of the first prepared unit plus a
        tenth (requests are collated longest-first); a longer prompt later simply grows the pair again."""
        d, eng = self._dims, self._model
        if not hasattr(eng, "reserve_kv"):
            return
        longest = max([len(p) for p in first_prep["prompts"]] + [1])
        rows = (int(longest * 1.1) + int(first_prep["max_new"]) + 2 + 15) // 16 * 16
        if hand_over:
            # slots for sequences handed over between passes: as many as a pass has, if 40 % of the memory left beside one pass's
            # cache holds them twice (a carried sequence's rows exist in the export AND in the next pass's cache for a moment);
            # none - no hand-over in this task - when not even 64 fit (an MHA decoder at a batch that fills the memory)
            slot_bytes = 4 * d.n_layers * d.n_kv_heads * d.head_dim * rows                    # K and V, bf16
            free = torch.cuda.mem_get_info(self._device)[0] + torch.cuda.memory_reserved(self._device) - torch.cuda.memory_allocated(self._device)
            held = 4 * eng._kv[0].numel() if getattr(eng, "_kv", None) else 0                 # (the current pair is released when it grows)
            fit = int(0.4 * (free + held - slots * slot_bytes) / (2 * slot_bytes))
            self._carry_capacity = max(0, min((slots + 63) // 64 * 64, fit // 64 * 64))
            slots = slots + self._carry_capacity
        eng.reserve_kv(d.n_layers * slots * d.n_kv_heads * d.head_dim * rows)

    def _start_workers(self) -> None:
        import os
        from concurrent.futures import ThreadPoolExecutor

        # NUMA placement (round 6): `os.sched_setaffinity(0, ...)` moves the CALLING THREAD only, and every thread
        # started while it is in force inherits the mask for good - round 5 pinned the launch thread here and restored only that
        # thread later, so torch's intra-op workers (created lazily, whenever the first CPU op ran) kept the one-node mask and
        # bench.py's CPU baseline ran its 128 threads on one socket's share.  Now: the mask is applied to exactly the threads that
        # want it - the launch thread (kernel launches, first touch of the pinned staging buffers) and, by an `initializer`, the PIL
        # workers and the preparation thread - and `unpin_host_threads()` gives EVERY thread of the process (/proc/self/task) the
        # original mask back, whoever inherited what in between.
        self._cpu_affinity_before = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None
        self._cpu_affinity = pin_to_gpu_numa_node(getattr(self, "_device", None))   # before any worker thread or pinned buffer exists
        share = self._cpu_affinity

        def pin_worker() -> None:
            if share and hasattr(os, "sched_setaffinity"):
                try:
                    os.sched_setaffinity(0, share)
                except OSError:
                    pass

        # host preparation: `OWC_PREP_THREADS` PIL workers (JPEG round trip + bicubic resize release the GIL) behind ONE
        # preparation thread that runs up to two engine batches ahead of the GPU (`_generate_rows`)
        # 8 workers prepare ~900 images/s per rank (JPEG round trip of a 448x448 image ~ 8 ms per worker), several times the GPU's
        # rate; more workers only take the GIL away from the thread that launches the kernels (measured on the bench's PIL leg:
        # 4-8 workers 0.89 of the engine rate, 32 workers 0.85, 64 workers 0.80)
        ranks_here = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
        # (CPUs = what the cgroup lets this process use, not what the host has: `usable_cpus`)
        self._prep_threads = int(os.environ.get("OWC_PREP_THREADS", max(2, min(8, usable_cpus()[0] // ranks_here))))
        self._pool = ThreadPoolExecutor(max_workers=self._prep_threads, thread_name_prefix="owc-pil", initializer=pin_worker)
        # ONE unit in preparation at a time (OWC_PREP_UNITS): two raise one rank's ceiling by ~14 % on an idle host but cost 3 % at the
        # production rate with 8 ranks on a 256-core host (0.975 -> 0.946 of 8 x 205 images/s, profiles/r05_host_soak.txt) - and so did
        # nothing else tried there for the 8-rank ceiling (a 0.5 ms GIL switch interval, Pillow's block cache, 12 / 16 workers): it sits
        # at ~3100-3300 images/s = 2.0 x the consumption at Food-101 sizes whatever the knob, ~4 busy cores per rank
        self._prep_thread = ThreadPoolExecutor(max_workers=max(1, int(os.environ.get("OWC_PREP_UNITS", "1"))), thread_name_prefix="owc-prep",
                                               initializer=pin_worker)
        import threading

        self._pinned_free, self._pinned_lock = [], threading.Lock()
        self._host_allocator = keep_image_blocks_mapped()

    PINNED_POOL_BYTES = 4 << 30   # retained (idle) pinned staging per rank; buffers in flight are bounded by the look-ahead

    def _h2d_groups(self, groups: list) -> list:
        """The uint8 image stacks of a pass -> device.  Pinned stacks (what `_prepare_chunk` stages: 1.2 GB for 2048 images of
        448 x 448) cross on a COPY STREAM of their own: a pass is launched while the previous one is still computing, so its
        copy runs beside that pass's kernels instead of queueing behind them on the compute stream (round 4 measured 0.6 % of a
        pass); the compute stream waits for the copy's event, the device buffers are handed to it (`record_stream`), and the
        pinned stacks go back to the staging pool only after the pass's own event (`_run_passes.finish`).  Anything else
        (pageable numpy stacks of a direct call) takes the compute stream as before."""
        from .. import _lib

        if not groups or not all(isinstance(g, torch.Tensor) and g.is_pinned() for g in groups) or os_env_off("OWC_COPY_STREAM"):
            return [_lib.h2d(g, self._device) for g in groups]
        if getattr(self, "_copy_stream", None) is None:
            self._copy_stream = torch.cuda.Stream(device=self._device)
        cur = torch.cuda.current_stream(self._device)
        with torch.cuda.stream(self._copy_stream):
            out = [g.to(self._device, non_blocking=True) for g in groups]
            done = torch.cuda.Event()
            done.record(self._copy_stream)
        cur.wait_event(done)
        for t in out:
            t.record_stream(cur)
        return out

    def unpin_host_threads(self) -> int:
        """Give EVERY thread of this process the CPU affinity it had before `_start_workers` pinned the rank to its GPU's NUMA share:
        the launch thread, the workers, and whatever was started in between and inherited the mask (torch's OpenMP / oneDNN intra-op
        workers, tokenizer pools).  Called when generation is over and host-side work follows - the engine's scoring pass, bench.py's
        CPU baseline.  Returns the number of threads whose mask was reset (0 when nothing was pinned)."""
        import os

        before = getattr(self, "_cpu_affinity_before", None)
        if not before or not getattr(self, "_cpu_affinity", None) or not hasattr(os, "sched_setaffinity"):
            return 0
        n = 0
        try:
            tids = [int(t) for t in os.listdir("/proc/self/task")]
        except OSError:
            tids = [0]
        for tid in tids:
            try:
                os.sched_setaffinity(tid, before)
                n += 1
            except OSError:      # a thread that exited between the listing and the call
                pass
        self._cpu_affinity = None
        return n

    def release_host_resources(self) -> None:
        """Stop the worker pools, drop the pinned staging buffers and give every thread of the process its CPU affinity back (a process
        that goes on to do host-side work of its own after the last task: bench.py's CPU baseline)."""
        for pool in (getattr(self, "_pool", None), getattr(self, "_prep_thread", None)):
            if pool is not None:
                pool.shutdown(wait=True)
        self._pinned_free = []
        self.unpin_host_threads()

    def _pinned_take(self, shape: tuple) -> torch.Tensor:
        """Pinned staging buffer for one same-size image run: reused across chunks (page-locking a fresh GB per chunk costs
        more than copying into it)."""
        n = int(np.prod(shape))
        with self._pinned_lock:
            for i, t in enumerate(self._pinned_free):
                if t.numel() >= n:
                    return self._pinned_free.pop(i)[:n].view(shape)
        return torch.empty(n, dtype=torch.uint8, pin_memory=True).view(shape)

    def _pinned_give(self, bufs: list) -> None:
        with self._pinned_lock:
            for b in bufs:
                base = b._base if b._base is not None else b
                self._pinned_free.append(base.reshape(-1))
            self._pinned_free.sort(key=lambda t: t.numel())
            del self._pinned_free[:-6]   # keep the six largest ...
            while len(self._pinned_free) > 1 and sum(t.numel() for t in self._pinned_free) > self.PINNED_POOL_BYTES:
                del self._pinned_free[0]   # ... within a byte cap: page-locked host memory is per rank, eight ranks share a host

    def _run_passes(self, requests: list, default_max_new: int) -> list[np.ndarray]:
        """Token rows (cut at EOS) per request, in request order.

        Two-stage pipeline.  A preparation thread readies the requests in small UNITS (engine_batch / 16, at least 64 requests:
        image fetch + JPEG round trip + resize on the PIL pool, prompt ids, pinned staging) strictly in order and runs up to
        two engine batches ahead.  This thread assembles the prepared units into engine passes ADAPTIVELY: the first pass takes
        whatever is ready when the GPU is idle (so the GPU starts after one unit, not after a whole chunk's preparation), every
        later pass waits until 1.5x the previous pass's requests are ready - or a full engine batch, or everything that is left -
        unless the GPU runs dry first, in which case it takes what is there.  With a host that prepares faster than the GPU
        consumes the passes grow geometrically to `engine_batch` and stay there; with a slower host the GPU is fed as the units
        arrive.  (Round 3 cut the first chunk 1/4 + 3/4: on one rank with 448 x 448 images - 900 prepared images/s against 240 -
        that hides everything, but eight ranks on real image sizes prepare 320 images/s per rank against 205, a task is 1.5-3
        engine batches per rank, and the ramp was a fifth of the run: tools/soak_host_ranks.py.)  Tokens do not depend on how
        the requests are grouped into passes (batch invariance, tested bit for bit).  A pass's ids come back through a pinned
        buffer + event one pass later, so the stream always holds the next pass's work when the host waits."""
        import time
        from collections import deque

        def _collate(x):
            return -len(self._collate_encode(x[0])), x[0]

        from .. import utils

        reordered = utils.Collator([reg.args for reg in requests], _collate, grouping=True)
        max_new = max([int(r.args[1].get("max_new_tokens", default_max_new)) for r in requests] + [1])
        eb = self.engine_batch(max_new)
        beams = max([beams_from_gen_kwargs(r.args[1]) for r in requests] + [1])
        if beams > 1:   # a beam-search pass holds num_beams hypotheses (cache slots, decode rows) per request
            eb = max(self.batch_size, eb // beams)
        unit = eb if eb < 256 else max(64, eb // 16)
        units = list(reordered.get_batched(n=unit, batch_fn=None))

        def unit_key(u) -> tuple:
            """What `_prepare_chunk` will put in prep["key"] for this unit: its first request's generation length and sampling
            switches (one pass = one key; a straggler is only ever handed to a pass of ITS key)."""
            gk = u[0][1]
            smp = sampling_from_gen_kwargs(gk, getattr(self, "_default_top_k", 50))
            return pass_key(int(gk.get("max_new_tokens", default_max_new)), smp, beams_from_gen_kwargs(gk))

        unit_keys = [unit_key(u) for u in units]
        taken = 0                               # units launched so far = index of the next pass's first unit
        tok = self._tokenizer
        pad = tok.pad_token_id if tok.pad_token_id is not None else 0
        rows: dict[int, np.ndarray] = {}        # position in the collated order -> token row (cut at EOS)
        ahead, inflight = deque(), deque()      # futures of submitted units (in order); launched passes (host ids, event, groups, ...)
        nxt = 0
        ahead_n = 0                             # requests submitted for preparation and not launched yet
        carried = None                          # unfinished sequences of the previous pass (Qwen2VLEngine.generate `carry`)
        launched = 0                            # requests launched so far = position of the next pass's first request

        def cut(r) -> np.ndarray:
            stop = np.flatnonzero(r == tok.eos_token_id)
            return r[: stop[0]].copy() if len(stop) else r.copy()

        def finish(item) -> None:
            host, ev, groups, pos0, skip = item
            ev.synchronize()          # the pass's GPU work is complete: its staging buffers can be reused
            self._pinned_give(groups)
            for i, r in enumerate(host.numpy()):
                if i not in skip:     # (a straggler handed to the next pass: its row comes back with that pass)
                    rows[pos0 + i] = cut(r)

        def top_up() -> None:
            nonlocal nxt, ahead_n
            while nxt < len(units) and (ahead_n < 2 * eb or not ahead):
                ahead.append((len(units[nxt]), self._prep_thread.submit(self._prepare_chunk, units[nxt])))
                ahead_n += len(units[nxt])
                nxt += 1

        def ready_prefix() -> tuple[int, bool]:
            """Requests in the leading run of prepared units that one pass can take (same generation length, <= engine_batch), and
            whether the run ends at such a limit (then waiting for more units cannot make the pass larger)."""
            n, first = 0, None
            for size, fut in ahead:
                if not fut.done():
                    return n, False
                if n + size > eb:
                    return n, True
                mn = fut.result()["key"]       # one pass = one generation length and one set of sampling switches
                if first is None:
                    first = mn
                elif mn != first:
                    return n, True
                n += size
            return n, nxt >= len(units)

        t_begin = time.perf_counter()
        self.last_timing = {"chunks": 0, "pass_sizes": []}
        last_size, left = 0, len(requests)
        while left:
            top_up()
            gpu_busy = bool(inflight) and not inflight[-1][1].query()
            have, closed = ready_prefix()
            if have == 0 and not gpu_busy:
                t_wait = time.perf_counter()
                ahead[0][1].result()            # nothing ready and the GPU is (about to be) idle: stand and wait for the next unit
                key = "first_chunk_prep_s" if not self.last_timing["chunks"] else "prep_wait_s"
                self.last_timing[key] = self.last_timing.get(key, 0.0) + time.perf_counter() - t_wait
                continue
            want = min(eb, left, max(unit, int(1.5 * last_size)))
            if left <= eb and left - want < want // 2:
                want = left                     # no small pass at the end of a task: it would run the decoder far below its rate
            elif eb < left < eb + eb // 2:
                want = min(want, (left + 1) // 2)   # ... nor a full pass followed by a sliver: two halves
            if gpu_busy and have < want and not closed:
                if len(inflight) > 1:           # use the wait: collect the pass before the one that is running
                    finish(inflight.popleft())
                else:
                    time.sleep(0.002)
                continue
            preps = []
            while ahead and sum(p["n"] for p in preps) < have:
                size, fut = ahead.popleft()
                preps.append(fut.result())
                ahead_n -= size
                taken += 1
            prep = preps[0] if len(preps) == 1 else self._merge_preps(preps)
            top_up()
            # straggler hand-over: while another pass follows, this pass stops decoding once its own live sequences are few enough
            # (`hand_over_below`) and the rest ride along in the next pass's decode steps
            # A pass applies ONE generation length and ONE set of sampling switches to every row of its decode batch, carried-in rows
            # included: the hand-over is only between passes of the same key.  A pass whose successor has another key (requests
            # are grouped by gen_kwargs) runs its own stragglers to the end, like the last pass of a task.
            carry = None
            more = left - prep["n"] > 0 and unit_keys[taken] == preps[0]["key"]
            hand_over = (tok.eos_token_id is not None and tok.eos_token_id >= 0 and (more or carried is not None)
                         and not getattr(self, "_no_carry", False) and prep.get("num_beams", 1) == 1)   # (a beam-search pass runs to its end)
            if not self.last_timing["chunks"]:
                # slots for carried sequences are reserved once per task: when ANY pass of it can hand over to a successor of its key
                some = more or any(a == b for a, b in zip(unit_keys[taken:], unit_keys[taken + 1:]))
                can = tok.eos_token_id is not None and tok.eos_token_id >= 0 and some and not getattr(self, "_no_carry", False)
                self._carry_capacity = (max(eb // 8, 256) + 255) // 256 * 256 if can else 0   # (an engine without `reserve_kv`)
                self._reserve_kv(prep, min(eb, len(requests)), can)
            if hand_over and self._carry_capacity > 0:
                n_in = 0 if carried is None else len(carried["tags"])
                carry = {"in": carried, "below": hand_over_below(prep["n"], n_in, self._carry_capacity) if more else 0,
                         "slots": self._carry_capacity, "tags": list(range(launched, launched + prep["n"]))}
            host, ev = self._launch_chunk(prep, tok.eos_token_id, pad, carry)
            skip = set()
            if carry is not None:
                for tag, full in carry["finished"]:
                    rows[tag] = cut(full)
                carried, skip = carry["out"], set(carry["unfinished_rows"])
            inflight.append((host, ev, prep.get("groups", []), launched, skip))
            launched += prep["n"]
            last_size = prep["n"]
            left -= prep["n"]
            self.last_timing["chunks"] += 1
            self.last_timing["pass_sizes"].append(prep["n"])
            if len(inflight) > 2:
                finish(inflight.popleft())
        while inflight:
            finish(inflight.popleft())
        self.last_timing["total_s"] = time.perf_counter() - t_begin
        self.last_timing.setdefault("first_chunk_prep_s", 0.0)
        assert carried is None and len(rows) == len(requests)
        return reordered.get_original([rows[i] for i in range(len(requests))])



def usable_cpus(cgroup_root: str = "/sys/fs/cgroup", proc_cgroup: str = "/proc/self/cgroup") -> tuple[int, float | None]:
    """(CPUs this process can actually keep busy, the cgroup CPU quota in CPUs or None).  `os.cpu_count()` and the affinity mask
    count the host's logical CPUs; a container's bandwidth limit (cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us / cpu.cfs_period_us`) is
    invisible to both, and more runnable threads than the quota only get the whole group THROTTLED - every thread stopped for the
    rest of each 100 ms period.  Round 6 found the 1-GPU MI355X boxes at 16 of 256 (tools/probe_cpu_quota.py): a bf16 `nn.Linear`
    ran 3.1x faster on 16 intra-op threads than on 128, and 8 emulated ranks x 8 PIL workers shared two CPUs per rank - that, not
    the pipeline, was the "8-rank host ceiling" of rounds 4-5 (profiles/r06_host_soak.txt)."""
    import math
    import os
    from pathlib import Path

    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    # cgroup v2: the tightest `cpu.max` from this process's group up to the mounted root (a limit on any ancestor binds)
    rel = ""
    try:
        for line in Path(proc_cgroup).read_text().splitlines():
            if line.startswith("0::"):
                rel = line[3:].strip().lstrip("/")
    except OSError:
        pass
    node, seen_v2 = Path(cgroup_root) / rel, False
    while True:
        try:
            q, period = (node / "cpu.max").read_text().split()[:2]
            seen_v2 = True
            if q != "max":
                quota = min(quota, int(q) / int(period)) if quota is not None else int(q) / int(period)
        except (OSError, ValueError):
            pass
        if node == Path(cgroup_root) or Path(cgroup_root) not in node.parents:
            break
        node = node.parent
    if not seen_v2:
        try:
            q = int((Path(cgroup_root) / "cpu" / "cpu.cfs_quota_us").read_text())
            period = int((Path(cgroup_root) / "cpu" / "cpu.cfs_period_us").read_text())
            if q > 0 and period > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, math.ceil(quota)))
    return n, quota


def keep_image_blocks_mapped() -> dict:
    """Stop the C allocators from mapping and unmapping memory once per image (round 6, tools/probe_host_prep_scaling.py).

    A prepared image is a handful of 0.2-1 MB blocks (RGB copy, JPEG buffers, two resize outputs, the CHW array); glibc serves
    blocks above 128 KiB by `mmap` and returns them by `munmap`, so every image costs a few thousand page faults on fresh zeroed
    pages, and every `munmap` in a process with N running threads takes its address-space lock and interrupts the N cores for the
    TLB flush.  `mallopt(M_MMAP_THRESHOLD / M_TRIM_THRESHOLD)` keeps such blocks on the (per-thread) heaps, which then stop growing
    after the first few images; Pillow's own block cache does the same for its image arenas.  One process, 8 PIL threads, Food-101
    sizes on the 256-core host: 1070 -> 1590 prepared images/s (16 threads: 1180 -> 1900).  `OWC_MALLOC_KEEP=0` / `OWC_PILLOW_BLOCKS=0`
    switch the two off; what was set is returned (and kept in `_host_allocator`)."""
    import os

    done = {"mallopt": False, "pillow_blocks": 0}
    if not os_env_off("OWC_MALLOC_KEEP"):
        try:
            import ctypes

            libc = ctypes.CDLL("libc.so.6")
            m_trim_threshold, m_mmap_threshold = -1, -3              # <malloc.h>
            limit = 64 << 20                                          # blocks up to 64 MiB stay on the heap (a 4096 x 4096 RGB image is 48 MiB)
            done["mallopt"] = bool(libc.mallopt(m_mmap_threshold, limit)) and bool(libc.mallopt(m_trim_threshold, 4 * limit))
        except (OSError, AttributeError):
            pass
    blocks = int(os.environ.get("OWC_PILLOW_BLOCKS", "256") or 0)    # x 16 MiB arena blocks kept for reuse instead of freed
    if blocks > 0:
        try:
            from PIL import Image

            Image.core.set_blocks_max(blocks)
            done["pillow_blocks"] = blocks
        except Exception:  # noqa: BLE001 - an old Pillow without the arena API
            pass
    return done


def os_env_off(name: str) -> bool:
    import os

    return os.environ.get(name, "1") == "0"


def gpu_local_cpus(pci_bus_id: str, sysfs: str = "/sys/bus/pci/devices") -> list[int]:
    """CPUs of the NUMA node a PCI device hangs off (`local_cpulist`, e.g. "0-31,128-159"); [] when sysfs does not say."""
    try:
        from pathlib import Path

        text = (Path(sysfs) / pci_bus_id.lower() / "local_cpulist").read_text().strip()
    except OSError:
        return []
    cpus: list[int] = []
    for part in filter(None, text.split(",")):
        lo, _, hi = part.partition("-")
        cpus += list(range(int(lo), int(hi or lo) + 1))
    return cpus


def numa_share(cpus: list[int], k: int, n: int, allowed: set | None = None) -> list[int]:
    """The k-th of n shares of a node's `cpus` (restricted to `allowed`): the ranks whose GPUs share a NUMA node split its cores
    between them instead of all roaming the node.  Every contiguous run of the list is cut into n slices and share k takes its
    slice of EACH run: a `local_cpulist` of "0-31,128-159" lists the cores and then their SMT siblings, so a rank gets whole cores
    (0-7 + 128-135), not another rank's hyper-threads."""
    cpus = sorted(c for c in cpus if allowed is None or c in allowed)
    if not cpus or n <= 0:
        return []
    runs, cur = [], [cpus[0]]
    for c in cpus[1:]:
        if c == cur[-1] + 1:
            cur.append(c)
        else:
            runs.append(cur)
            cur = [c]
    runs.append(cur)
    if min(len(r) for r in runs) // n < 1:
        return cpus           # fewer CPUs per run than ranks: no split
    out: list[int] = []
    for r in runs:
        per = len(r) // n
        out += r[k * per:(k + 1) * per]
    return out


def pin_to_gpu_numa_node(device) -> list[int] | None:
    """One process per GPU on a two-socket host (the reference launches its ranks with `accelerate launch`,
    /root/reference/scripts/schedule_batch.sh:109-112, /root/reference/src/utils/_core_utils.py:53-69, and leaves placement to the
    kernel): this rank's launch thread, its PIL workers and its pinned staging buffers (first touch) belong on the socket its GPU
    hangs off - a rank that prepares images on the far socket pays the inter-socket link twice (JPEG bytes in, pinned uint8 out)
    and competes with the ranks that live there.  The process's affinity becomes this rank's share of its GPU's `local_cpulist`
    (the ranks of a node split it evenly, in local-rank order); threads started later inherit it.  Off with OWC_NUMA_PIN=0; a no-op
    without a CUDA device, without sysfs topology, inside a cpuset that excludes the node, or on a single-node host."""
    import os

    if os.environ.get("OWC_NUMA_PIN", "1") == "0" or device is None or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        import torch

        dev = torch.device(device)
        if dev.type != "cuda" or not torch.cuda.is_available():
            return None
        n_dev = torch.cuda.device_count()

        def bus_id(i: int) -> str:
            p = torch.cuda.get_device_properties(i)
            return f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"

        mine = gpu_local_cpus(bus_id(dev.index or 0))
        allowed = set(os.sched_getaffinity(0))
        if not mine or not (set(mine) & allowed) or set(mine) >= allowed:
            return None     # no topology, a cpuset that excludes the node, or one node that holds everything we may use anyway
        # the ranks of this job whose GPUs hang off the same node, in local-rank order (one rank per visible device)
        ranks_here = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
        same = [i for i in range(min(n_dev, ranks_here)) if set(gpu_local_cpus(bus_id(i))) == set(mine)] if ranks_here > 1 else [dev.index or 0]
        k = same.index(dev.index or 0) if (dev.index or 0) in same else 0
        share = numa_share(mine, k, max(len(same), 1), allowed)
        if not share:
            return None
        os.sched_setaffinity(0, share)
        return share
    except (OSError, RuntimeError, AttributeError, ValueError):
        return None


def hand_over_below(n_own: int, n_carried_in: int, capacity: int) -> int:
    """The straggler hand-over threshold of one pass (`Qwen2VLEngine.generate(carry={"below": ...})`): the pass stops decoding
    once at most this many of its OWN sequences are unfinished; they go on inside the next pass's decode steps.

    A decode step costs about t0 + c * rows (7B on MI355X: ~5 ms + ~12.5 us per row): every step a pass does NOT run with a
    shrinking batch saves t0, the tokens themselves cost c wherever they are decoded - so hand over as early as the slots allow:
    half of the pass, or what is left of the `capacity` (slots reserved for carried sequences, `PassPipeline._reserve_kv`) beside
    the sequences that came in with this pass.  Round 4 first handed over at 1/64 of a pass (32 of 2048): with classification
    answers (mean 8 tokens) a pass then ran ~35 decode steps instead of ~6, with chain-of-thought answers (mean 100, cap 256) all
    255 - DESIGN.md section 8.  The tokens do not depend on the threshold (tests/test_qwen2vl_gpu.py, test_plugin_gpu.py)."""
    return max(8, min(n_own // 2, capacity - n_carried_in))


def pass_key(max_new: int, sampling: dict | None, num_beams: int = 1) -> tuple:
    """What one engine pass is uniform in: generation length, sampling switches, beams."""
    smp = None if sampling is None else (sampling["temperature"], sampling["top_p"], sampling["top_k"])
    return (int(max_new), smp) if num_beams == 1 else (int(max_new), smp, int(num_beams))


def beams_from_gen_kwargs(gen_kwargs: dict) -> int:
    """`num_beams` of a request (reference src/models/_qwen2_vl.py:308-329, _llava_hf.py:355-376: default 1, handed to HF's generate)."""
    k = int(gen_kwargs.get("num_beams", 1) or 1)
    if k < 1:
        raise ValueError("num_beams must be >= 1")
    if k > 32:   # owc_beam_candidates returns at most 64 = 2 x num_beams candidates per row: refuse here, not inside the first decode step
        raise ValueError("num_beams > 32 is not supported by the HIP decoder (owc_beam_candidates: k = 2 * num_beams <= 64)")
    return k


def sampling_from_gen_kwargs(gen_kwargs: dict, default_top_k: int = 50) -> dict | None:
    """The reference's generation switches -> the engine's `sampling` argument (None = greedy).

    Reference (src/models/_qwen2_vl.py:308-329, _llava_hf.py:355-376): `temperature` (default 0), `top_p` (default None),
    `num_beams` (default 1) from the request's gen_kwargs, `do_sample = temperature > 0`.  HF then samples through
    temperature -> top-k -> top-p with top_k from the checkpoint's generation_config.json (HF's own default is 50; Qwen2-VL's
    ships `top_k: 1`, which makes its sampling the argmax) - `default_top_k` carries that value.  `num_beams` > 1 with temperature 0
    is beam search (`beams_from_gen_kwargs`, `Qwen2VLEngine.generate_beam`); with temperature > 0 (beam sampling) it raises.  The
    random stream is the library's documented Philox stream keyed by torch's seed (the reference seeds torch with `--seed`, eval_model.py), one stream per DOCUMENT."""
    t = gen_kwargs.get("temperature", 0) or 0
    if not float(t) > 0:
        return None
    if beams_from_gen_kwargs(gen_kwargs) != 1:
        raise NotImplementedError("beam SAMPLING (num_beams > 1 with temperature > 0) is not implemented by the HIP decoder; beam search "
                                  "(temperature 0) and one-beam sampling are")
    import torch

    return {"temperature": float(t), "top_p": gen_kwargs.get("top_p"), "top_k": int(default_top_k), "seed": int(torch.initial_seed())}

