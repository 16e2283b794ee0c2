"""Model-level parity on the GPU: HIP engine (through the C ABI) vs the numpy oracle, tiny Qwen2-VL
config with real structure (GQA, head_dim 128, mrope, vision head_dim 80), plus the golden vectors
HF produced for the same weights (tests/golden/qwen2vl_tiny.npz).

Tolerance (stated): both sides round to bf16 at the same module boundaries but accumulate in a
different order, so activations agree to a few bf16 ulps of their scale: max |diff| <= 3% of max |ref|
for logits; greedy tokens must match wherever the oracle's top-2 logit margin exceeds that bound.
"""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import qwen2vl_np as Q
from tests import recipes
from tests.util import to_np

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"


@pytest.fixture(scope="module")
def setup(gpu):
    from lmms_owc_amd.engine.qwen2vl import Qwen2VLDims, Qwen2VLEngine, Qwen2VLWeights

    cfg = recipes.tiny_cfg()
    w = recipes.qwen2vl_weights(cfg, 1234)
    dims = Qwen2VLDims(v_depth=2, v_embed=160, v_heads=2, v_mlp=640, n_layers=2, d_model=256, n_q_heads=2, n_kv_heads=1,
                       d_ff=512, vocab=512, tie_embeddings=False, image_token_id=500, max_positions=512, max_grid=64)
    eng = Qwen2VLEngine(Qwen2VLWeights.from_state_dict(dims, w, gpu), vit_chunk_tokens=64, prefill_chunk_tokens=40)
    return cfg, w, eng, np.load(GOLD / "qwen2vl_tiny.npz")


# model-level bounds: 2 % of max |ref| against the bf16 numpy oracle, 2.5 % against HF's bf16 / fp32 runs (observed on MI355X:
# 0.4-1.1 %; HF's own bf16-vs-fp32 gap on these goldens is 1.0-1.3 %)
def _close(got, ref, frac, tag=""):
    from tests.util import assert_rel_close

    assert_rel_close(got, ref, frac, tag or "model-level")


@pytest.mark.parametrize("case", ["a", "b"])
def test_vit_matches_oracle_and_hf(setup, gpu, case):
    cfg, w, eng, g = setup
    grid = [tuple(r) for r in g[f"{case}_grid"].tolist()]
    pix = recipes.pixel_values(grid, 7)
    out = to_np(eng.encode_images(torch.from_numpy(pix).to(torch.bfloat16).to(gpu), grid))
    _close(out, Q.vit_forward(w, cfg, pix, grid, bf16=True), 0.02)
    _close(out, g[f"{case}_bf16_vit"], 0.025)   # HF bf16 (CPU) run of the same weights
    _close(out, g[f"{case}_f32_vit"], 0.025)    # HF fp32


@pytest.mark.parametrize("case", ["a", "b"])
def test_generate_matches_oracle_and_hf(setup, gpu, case):
    cfg, w, eng, g = setup
    grid = [tuple(r) for r in g[f"{case}_grid"].tolist()]
    pix = recipes.pixel_values(grid, 7)
    ids = g[f"{case}_ids"]
    emb = eng.encode_images(torch.from_numpy(pix).to(torch.bfloat16).to(gpu), grid)
    toks, logits = eng.generate([ids], emb, [grid], 8, return_logits=True)
    toks, logits = to_np(toks)[0].astype(int), to_np(logits)[0]
    o_toks, o_logits = Q.generate(w, cfg, ids, pix, grid, 8, bf16=True, return_logits=True)
    _close(logits, o_logits[0], 0.02)
    _close(logits, g[f"{case}_bf16_logits"][0], 0.025)
    _close(logits, g[f"{case}_f32_logits"][0], 0.025)
    # free-running tokens: identical to HF's up to the first near-tie (after it the continuations may legitimately differ);
    # EVERY step is asserted under teacher forcing in tests/test_decode_parity_gpu.py
    ref = g[f"{case}_f32_logits"]
    margins = [np.sort(ref[j])[-1] - np.sort(ref[j])[-2] > 0.06 * np.abs(ref[j]).max() for j in range(8)]
    n_sure = margins.index(False) if False in margins else 8
    assert n_sure >= 1 and np.array_equal(toks[:n_sure], g[f"{case}_f32_tokens"][:n_sure]), (toks, g[f"{case}_f32_tokens"])


def test_batched_equals_single(setup, gpu):
    """Batching (packed varlen prefill, batched decode, chunking) must not change any sequence's tokens."""
    cfg, w, eng, g = setup
    cases = []
    for case in ("a", "b"):
        grid = [tuple(r) for r in g[f"{case}_grid"].tolist()]
        cases.append((g[f"{case}_ids"], grid, recipes.pixel_values(grid, 7)))
    cases.append((recipes.prompt_ids(cfg, [(1, 4, 4)], seed=99), [(1, 4, 4)], recipes.pixel_values([(1, 4, 4)], 8)))
    singles = []
    for ids, grid, pix in cases:
        emb = eng.encode_images(torch.from_numpy(pix).to(torch.bfloat16).to(gpu), grid)
        singles.append(to_np(eng.generate([ids], emb, [grid], 6))[0])
    pix_all = np.concatenate([c[2] for c in cases])
    grids_all = [gg for c in cases for gg in c[1]]
    emb = eng.encode_images(torch.from_numpy(pix_all).to(torch.bfloat16).to(gpu), grids_all)
    batch = to_np(eng.generate([c[0] for c in cases], emb, [c[1] for c in cases], 6))
    for b in range(len(cases)):
        assert np.array_equal(batch[b], singles[b]), (b, batch[b], singles[b])


def test_eos_pads_rest(setup, gpu):
    cfg, w, eng, g = setup
    grid = [(1, 4, 4)]
    pix = recipes.pixel_values(grid, 7)
    ids = g["a_ids"]
    emb = eng.encode_images(torch.from_numpy(pix).to(torch.bfloat16).to(gpu), grid)
    free = to_np(eng.generate([ids], emb, [grid], 6))[0].astype(int)
    eos = int(free[2])
    first = int(np.flatnonzero(free == eos)[0])
    got = to_np(eng.generate([ids], emb, [grid], 6, eos_token_id=eos, pad_token_id=0, stop_check_every=1))[0].astype(int)
    assert np.array_equal(got[:first + 1], free[:first + 1]) and (got[first + 1:] == 0).all()


def test_shared_prefix_is_bit_identical(setup, gpu):
    """Prefilling the prompts' common leading tokens once (shared-prefix segment + K/V broadcast) must give exactly
    the tokens and first-step logits of the plain per-prompt prefill."""
    from lmms_owc_amd.engine.qwen2vl import Qwen2VLEngine

    cfg, w, eng, g = setup
    plain = Qwen2VLEngine(eng.w, vit_chunk_tokens=64, prefill_chunk_tokens=4096, share_prefix=False)
    shared = Qwen2VLEngine(eng.w, vit_chunk_tokens=64, prefill_chunk_tokens=4096, share_prefix=True, min_shared_prefix=2)
    r = np.random.default_rng(3)
    head = r.integers(1, 400, 7)
    prompts, grids, pixs = [], [], []
    for i, grid in enumerate([[(1, 4, 4)], [(1, 6, 4)], [(1, 4, 8)], [(1, 4, 4)]]):
        n_img = grid[0][1] * grid[0][2] // 4
        prompts.append(np.concatenate([head, np.full(n_img, cfg.image_token_id), r.integers(1, 400, 3 + i)]))
        grids.append(grid)
        pixs.append(recipes.pixel_values(grid, 20 + i))
    pix = torch.from_numpy(np.concatenate(pixs)).to(torch.bfloat16).to(gpu)
    emb = eng.encode_images(pix, [gg for gs in grids for gg in gs])
    assert shared._common_prefix(prompts, 0, 4) == 7
    a, la = plain.generate(prompts, emb, grids, 6, return_logits=True)
    b, lb = shared.generate(prompts, emb, grids, 6, return_logits=True)
    assert torch.equal(a, b) and torch.equal(la, lb)
    # two chunks (the prefix is shared per chunk) and a chunk of one prompt (no sharing) agree as well
    shared2 = Qwen2VLEngine(eng.w, vit_chunk_tokens=64, prefill_chunk_tokens=40, share_prefix=True, min_shared_prefix=2)
    c = shared2.generate(prompts, emb, grids, 6)
    assert torch.equal(a, c)


def test_pruned_last_prefill_layer_is_bit_identical(setup, gpu):
    """owc_llm_prefill runs the last layer's attention / o-proj / MLP on the last-token rows only (K/V still for every row);
    the first-step logits and all generated tokens must equal the full last layer bit for bit, with and without the
    shared-prefix segment, for the bf16 decoder."""
    from lmms_owc_amd import _lib
    from lmms_owc_amd.engine.qwen2vl import Qwen2VLEngine

    cfg, w, eng, g = setup
    r = np.random.default_rng(5)
    head = r.integers(1, 400, 6)
    prompts, grids, pixs = [], [], []
    for i, grid in enumerate([[(1, 4, 4)], [(1, 6, 4)], [(1, 4, 8)]]):
        n_img = grid[0][1] * grid[0][2] // 4
        prompts.append(np.concatenate([head, np.full(n_img, cfg.image_token_id), r.integers(1, 400, 2 + 2 * i)]))
        grids.append(grid)
        pixs.append(recipes.pixel_values(grid, 40 + i))
    pix = torch.from_numpy(np.concatenate(pixs)).to(torch.bfloat16).to(gpu)
    emb = eng.encode_images(pix, [gg for gs in grids for gg in gs])
    lib = _lib.load()
    try:
        for share in (False, True):
            e = Qwen2VLEngine(eng.w, vit_chunk_tokens=64, prefill_chunk_tokens=4096, share_prefix=share, min_shared_prefix=2)
            assert lib.owc_tuning_set(b"prefill_prune_last", 0) == 0
            a, la = e.generate(prompts, emb, grids, 5, return_logits=True)
            assert lib.owc_tuning_set(b"prefill_prune_last", 1) == 0
            b, lb = e.generate(prompts, emb, grids, 5, return_logits=True)
            assert torch.equal(la, lb) and torch.equal(a, b)
    finally:
        lib.owc_tuning_set(b"prefill_prune_last", 1)


def test_graph_replayed_decode_equals_eager(setup, gpu):
    """owc_llm_decode_step with a device-side step state is a fixed launch sequence: steps 2.. replayed from ONE captured hipGraph
    give exactly the eager loop's tokens (also with EOS stopping and pads)."""
    from lmms_owc_amd.engine.qwen2vl import Qwen2VLEngine

    cfg, w, eng, g = setup
    eager = Qwen2VLEngine(eng.w, graph_decode=False)
    graph = Qwen2VLEngine(eng.w, graph_decode=True)
    r = np.random.default_rng(11)
    prompts = [r.integers(1, 400, 9 + 2 * i) for i in range(5)]
    a = eager.generate(prompts, None, [[] for _ in prompts], 12)
    b = graph.generate(prompts, None, [[] for _ in prompts], 12)
    assert torch.equal(a, b)
    eos = int(a[0, 3])  # a token sequence 0 emits at step 3: from there on it must be padded in both
    a2 = eager.generate(prompts, None, [[] for _ in prompts], 12, eos_token_id=eos, pad_token_id=0, stop_check_every=2)
    b2 = graph.generate(prompts, None, [[] for _ in prompts], 12, eos_token_id=eos, pad_token_id=0, stop_check_every=2)
    assert torch.equal(a2, b2) and int((a2[0, 4:] != 0).sum()) == 0


def _ragged_forced(rng, B, T, eos, vocab, mean_len, cap_frac=0.02):
    """Seeded per-sequence continuations: geometric stop lengths (a few forced to the cap), EOS at the stop column."""
    lens = np.minimum(rng.geometric(1.0 / mean_len, B), T)
    lens[rng.random(B) < cap_frac] = T + 1                       # never stops inside the cap
    forced = rng.integers(1, vocab, (B, T))
    forced[forced == eos] = eos - 1
    for b in range(B):
        if lens[b] <= T:
            forced[b, lens[b] - 1] = eos
    return forced.astype(np.int32), lens


def test_row_compaction_is_bit_identical(setup, gpu):
    """EOS-aware row compaction of the decode batch (reference: every image is its own `generate` call that stops at its own
    EOS, src/models/_qwen2_vl.py:319-337): dropping the finished rows from the following steps changes no token of any row -
    forced (seeded ragged answer lengths) and free-running, with image prompts of unequal length, across the kernel switches
    the shrinking batch goes through (256x256 / ring / skinny GEMMs, both decode-attention forms)."""
    cfg, w, eng, g = setup
    r = np.random.default_rng(77)
    B, T, eos = 150, 40, 9
    prompts = [r.integers(10, 400, 6 + (i % 9)).astype(np.int64) for i in range(B)]
    none = [[] for _ in prompts]
    forced, lens = _ragged_forced(r, B, T, eos, 400, 6)
    st_a, st_b = {}, {}
    a = to_np(eng.generate(prompts, None, none, T, eos_token_id=eos, pad_token_id=0, forced_tokens=forced, compact_rows=False, stats=st_a))
    b = to_np(eng.generate(prompts, None, none, T, eos_token_id=eos, pad_token_id=0, forced_tokens=forced, compact_rows=True, stats=st_b))
    assert np.array_equal(a, b)
    for i in range(B):                                         # a row is padded behind the column its forced continuation ends at
        assert (a[i, min(lens[i], T):] == 0).all() and (lens[i] > T or a[i, :lens[i]].min() >= 0)
    la, lb = st_a["live_rows_per_step"], st_b["live_rows_per_step"]
    assert max(la) == B and lb[0] == B and min(lb) < B // 8 and sum(lb) < 0.5 * sum(la), (la, lb)
    assert all(x >= y for x, y in zip(lb, lb[1:]))
    # every `stop_check_every` gives the same tokens (rows only leave later)
    c = to_np(eng.generate(prompts, None, none, T, eos_token_id=eos, pad_token_id=0, forced_tokens=forced, stop_check_every=5))
    assert np.array_equal(a, c)
    # free-running with a frequent token as EOS, image prompts
    grids = [[(1, 4, 4)], [(1, 6, 4)], [(1, 4, 8)], [(1, 4, 4)], [(1, 8, 4)]] * 4
    ps, pixs = [], []
    for i, grid in enumerate(grids):
        n_img = grid[0][1] * grid[0][2] // 4
        ps.append(np.concatenate([r.integers(1, 400, 3 + i % 4), np.full(n_img, cfg.image_token_id), r.integers(1, 400, 2 + i % 3)]))
        pixs.append(recipes.pixel_values(grid, 60 + i))
    emb = eng.encode_images(torch.from_numpy(np.concatenate(pixs)).to(torch.bfloat16).to(gpu), [gg for gs in grids for gg in gs])
    free = to_np(eng.generate(ps, emb, grids, 16))
    vals, counts = np.unique(free[:, 1:12], return_counts=True)
    eos2 = int(vals[np.argmax(counts)])
    s2 = {}
    x = to_np(eng.generate(ps, emb, grids, 16, eos_token_id=eos2, pad_token_id=0, compact_rows=False))
    y = to_np(eng.generate(ps, emb, grids, 16, eos_token_id=eos2, pad_token_id=0, compact_rows=True, stats=s2))
    assert np.array_equal(x, y)
    for i in range(len(ps)):                                   # and both equal the free run up to the first EOS, pad behind it
        hit = np.flatnonzero(free[i] == eos2)
        k = int(hit[0]) + 1 if len(hit) else 16
        assert np.array_equal(x[i, :k], free[i, :k]) and (x[i, k:] == 0).all()
    assert min(s2["live_rows_per_step"]) < len(ps)


def test_straggler_carry_over_is_bit_identical(setup, gpu):
    """Straggler hand-over between the passes of a task (`generate(..., carry=)`): a pass stops decoding once its OWN live
    sequences are few, the unfinished ones - K / V rows, pending token, position, remaining budget - join the NEXT pass's decode
    steps as extra rows, the last pass runs everything to its end.  Every sequence's tokens equal those of its pass run alone to
    completion: forced (seeded ragged answer lengths with sequences that never stop inside the cap), free-running, and sampled."""
    cfg, w, eng, g = setup
    r = np.random.default_rng(123)
    eos = 9
    sizes = (70, 50, 33, 21)
    caps = (40, 24, 40, 32)      # passes of different generation lengths (requests grouped by gen_kwargs): a carried sequence keeps ITS cap
    passes = [[r.integers(10, 400, 6 + (i % 9)).astype(np.int64) for i in range(n)] for n in sizes]
    forced = [_ragged_forced(r, n, T, eos, 400, 6, cap_frac=0.08)[0] for n, T in zip(sizes, caps)]

    def run(with_forced: bool, eos_id: int, below: int, sampled: bool = False):
        smp = [None if not sampled else {"temperature": 0.9, "top_k": 30, "seed": 11, "stream_ids": 1000 * k + np.arange(len(p))}
               for k, p in enumerate(passes)]
        ref = [to_np(eng.generate(p, None, [[] for _ in p], caps[k], eos_token_id=eos_id, pad_token_id=0, sampling=smp[k],
                                  forced_tokens=forced[k] if with_forced else None)) for k, p in enumerate(passes)]
        got = {}
        state, handed = None, 0
        for k, p in enumerate(passes):
            c = {"in": state, "below": below if k + 1 < len(passes) else 0, "tags": [(k, i) for i in range(len(p))]}
            out = to_np(eng.generate(p, None, [[] for _ in p], caps[k], eos_token_id=eos_id, pad_token_id=0, sampling=smp[k],
                                     forced_tokens=forced[k] if with_forced else None, carry=c))
            for i in range(len(p)):
                if i not in c["unfinished_rows"]:
                    got[(k, i)] = out[i]
            for tag, full in c["finished"]:
                got[tag] = full
            state = c["out"]
            handed += 0 if state is None else len(state["tags"])
        assert state is None and len(got) == sum(sizes)
        for k, n in enumerate(sizes):
            for i in range(n):
                assert np.array_equal(got[(k, i)], ref[k][i]), (with_forced, k, i, got[(k, i)], ref[k][i])
        return handed

    assert run(True, eos, 6) >= 6                    # sequences really travelled (some through more than one pass)
    free = to_np(eng.generate(passes[0], None, [[] for _ in passes[0]], caps[0]))
    vals, counts = np.unique(free[:, 2:20], return_counts=True)
    assert run(False, int(vals[np.argmax(counts)]), 40) >= 10   # (free-running: many sequences never emit that token)
    # sampled: a carried sequence keeps its random stream and goes on counting its own steps
    assert run(False, int(vals[np.argmax(counts)]), 40, sampled=True) >= 10


def test_finished_rows_never_write_past_their_cache_slot(setup, gpu):
    """Round 4's ADVICE: a finished row stays in the decode batch until >= n / 64 rows are finished, and every step advances its
    cache write index.  The last pass of a task with 130 carried-in sequences (38 tokens still to go each) and ONE own long prompt
    with a short generation length: the own row's budget ends at step 3, one finished row of 131 is below the compaction
    threshold (2), so it used to ride along for 35 more steps and write K / V rows past its slot - into slot 1, the first carried
    sequence's first keys.  Now the host drops such a row before it reaches the end of its slot (and the kernel refuses the
    write): every carried sequence's tokens equal those of its own pass run alone; also with no spare slot at all
    (`slots` = the carried sequences) and with `stop_check_every` > 1."""
    cfg, w, eng, g = setup
    r = np.random.default_rng(5)
    first = [r.integers(10, 400, 10).astype(np.int64) for _ in range(130)]      # identical prompt lengths
    last = [r.integers(10, 400, 60).astype(np.int64)]
    none = [[] for _ in first]
    free = to_np(eng.generate(first, None, none, 40))
    eos = int(np.setdiff1d(np.arange(10, 400), np.concatenate([free.ravel(), to_np(eng.generate(last, None, [[]], 4)).ravel()]))[0])
    ref0 = to_np(eng.generate(first, None, none, 40, eos_token_id=eos, pad_token_id=0))   # nobody emits `eos`: 40 tokens each
    ref1 = to_np(eng.generate(last, None, [[]], 4, eos_token_id=eos, pad_token_id=0))
    assert np.array_equal(ref0, free)
    for slots, every in ((None, 1), (130, 1), (130, 3)):
        c0 = {"in": None, "below": 130, "tags": [(0, i) for i in range(130)]}
        if slots:
            c0["slots"] = slots
        eng.generate(first, None, none, 40, eos_token_id=eos, pad_token_id=0, carry=c0)
        assert c0["out"] is not None and len(c0["out"]["tags"]) == 130 and int(np.min(c0["out"]["remaining"])) >= 30
        c1 = {"in": c0["out"], "below": 0, "tags": [(1, 0)]}
        if slots:
            c1["slots"] = slots
        st = {}
        out1 = to_np(eng.generate(last, None, [[]], 4, eos_token_id=eos, pad_token_id=0, carry=c1, stop_check_every=every, stats=st))
        assert c1["out"] is None and np.array_equal(out1, ref1)
        assert min(st["live_rows_per_step"]) == 130            # the own row did leave the batch although 1 < 131 // 64
        got = dict(c1["finished"])
        for i in range(130):
            assert np.array_equal(got[(0, i)], ref0[i]), (slots, every, i, got[(0, i)], ref0[i])


def test_sampled_generation_is_a_function_of_seed_and_stream_only(setup, gpu):
    """`generate(..., sampling=)` - HF's do_sample path (reference src/models/_qwen2_vl.py:319-329): a sequence's sampled tokens
    depend on (weights, prompt, seed, its stream id) only: the same alone and inside a batch, with and without row compaction,
    run after run; top_k = 1 reproduces the greedy tokens; another seed gives another continuation."""
    cfg, w, eng, g = setup
    r = np.random.default_rng(31)
    prompts = [r.integers(10, 400, 7 + (i % 5)).astype(np.int64) for i in range(40)]
    none = [[] for _ in prompts]
    sp = {"temperature": 0.9, "top_k": 20, "top_p": 0.95, "seed": 77, "stream_ids": np.arange(100, 140)}
    a = to_np(eng.generate(prompts, None, none, 12, sampling=sp))
    assert np.array_equal(a, to_np(eng.generate(prompts, None, none, 12, sampling=sp)))
    for i in (0, 17, 39):
        solo = to_np(eng.generate([prompts[i]], None, [[]], 12, sampling={**sp, "stream_ids": [100 + i]}))
        assert np.array_equal(solo[0], a[i]), i
    greedy = to_np(eng.generate(prompts, None, none, 12))
    # top_k = 1 keeps the maximal VALUE (ties included, as HF's `scores < kth value` rule does): teacher-forced on the greedy tokens,
    # every drawn token must carry its step's maximal logit (the tiny model's bf16 logits tie at the top in ~10 % of the steps)
    t1, l1 = eng.generate(prompts, None, none, 12, sampling={"temperature": 1.3, "top_k": 1, "seed": 5}, forced_tokens=greedy,
                          return_step_logits=True)
    l1 = l1.float()                                           # [T, B, V]
    picked = l1.gather(2, t1.long().T[:, :, None])[:, :, 0]
    assert torch.equal(picked, l1.max(dim=2).values)
    b = to_np(eng.generate(prompts, None, none, 12, sampling={**sp, "seed": 78}))
    assert (a != b).mean() > 0.3 and (a != greedy).mean() > 0.3
    vals, counts = np.unique(a[:, 1:8], return_counts=True)
    eos = int(vals[np.argmax(counts)])
    x = to_np(eng.generate(prompts, None, none, 12, sampling=sp, eos_token_id=eos, compact_rows=False))
    st = {}
    y = to_np(eng.generate(prompts, None, none, 12, sampling=sp, eos_token_id=eos, compact_rows=True, stats=st))
    assert np.array_equal(x, y) and min(st["live_rows_per_step"]) < len(prompts)


def test_rmsnorm_fused_into_the_skinny_gemm_is_bit_identical(setup, gpu):
    """Decode at 1-2 sequences (the kernel takes up to 4: the knob's maximum is exercised here): the qkv / gate-up projections
    normalise their own activations (gemm_bf16_skinny_norm_kernel, one launch less per RMSNorm).  Same bits as rmsnorm + GEMM - for
    every step's logits with the knob on and off, and therefore across the switch back to the separate kernels (a sequence alone
    == inside a batch of 6)."""
    from lmms_owc_amd import _lib

    cfg, w, eng, g = setup
    r = np.random.default_rng(21)
    prompts = [r.integers(1, 490, 11 + 2 * i).astype(np.int64) for i in range(6)]
    lib = _lib.load()
    for B in (1, 3, 4):
        try:
            assert lib.owc_tuning_set(b"decode_norm_fuse", 0) == 0
            ta, la = eng.generate(prompts[:B], None, [[] for _ in range(B)], 6, return_step_logits=True)
            assert lib.owc_tuning_set(b"decode_norm_fuse", 4) == 0
            tb, lb = eng.generate(prompts[:B], None, [[] for _ in range(B)], 6, return_step_logits=True)
        finally:
            lib.owc_tuning_set(b"decode_norm_fuse", -1)
        assert torch.equal(la, lb) and torch.equal(ta, tb), B
    big = eng.generate(prompts, None, [[] for _ in prompts], 6)                    # 6 rows: the separate kernels
    for i in (0, 5):
        assert torch.equal(eng.generate([prompts[i]], None, [[]], 6)[0], big[i]), i   # 1 row: the fused kernel


@pytest.mark.parametrize("k,T", [(2, 6), (4, 8), (5, 7)])
def test_beam_search_equals_the_oracle_driven_by_the_engines_own_logits(setup, gpu, k, T):
    """`generate_beam` (device: decode step + `owc_beam_candidates`; host: `BeamSearcher`; the KV cache following the hypotheses by
    slot) against `oracle/beam_np.beam_search` - HF's bookkeeping, pinned on HF in tests/test_oracle_beam.py - whose logits come from
    the engine itself, every continuation recomputed FROM THE PROMPT under teacher forcing (no cache reuse between steps).  The
    engine computes a row independently of its neighbours, so both sides see the same bf16 logits and the tokens must be EQUAL: what
    this binds is the slot bookkeeping (a hypothesis reading another one's K / V rows changes its logits) and the candidate kernel.
    Prompts of different lengths, with and without an image, searched in one batch; EOS ids the searches meet, so prompts end at
    different steps."""
    from oracle import beam_np as BM

    cfg, w, eng, g = setup
    grid = [(1, 4, 4)]
    emb = eng.encode_images(torch.from_numpy(recipes.pixel_values(grid, 7)).to(torch.bfloat16).to(gpu), grid)
    n_img = emb.shape[0]
    rng = np.random.default_rng(100 * k + T)
    prompts = [np.asarray(g["a_ids"]), rng.integers(1, 480, 9).astype(np.int64), np.asarray(g["a_ids"])[:-2], rng.integers(1, 480, 23)]
    grids = [grid if (p == 500).any() else [] for p in prompts]
    rows = [np.arange(n_img) if (p == 500).any() else np.zeros(0, np.int64) for p in prompts]

    def logits_fn_of(b):
        def fn(conts):
            gl = len(conts[0])
            forced = np.zeros((len(conts), gl + 1), np.int64)
            forced[:, :gl] = np.asarray(conts, np.int64).reshape(len(conts), gl)
            _, lg = eng.generate([prompts[b]] * len(conts), emb, [grids[b]] * len(conts), gl + 1, img_rows=[rows[b]] * len(conts),
                                 forced_tokens=forced, return_step_logits=True)
            return lg[gl].float().cpu().numpy()
        return fn

    free = to_np(eng.generate(prompts, emb, grids, T, img_rows=rows)).astype(int)
    for eos in (int(free[0, 2]), int(free[1, min(4, T - 1)]), 511):       # 511: (almost surely) never met - the length limit ends it
        want = [BM.beam_search(logits_fn_of(b), len(prompts[b]), k, T, eos, 0) for b in range(len(prompts))]
        got, scores = eng.generate_beam(prompts, emb, grids, T, k, eos_token_id=eos, pad_token_id=0, img_rows=rows, return_scores=True)
        got = to_np(got).astype(int)
        for b in range(len(prompts)):
            assert np.array_equal(got[b], want[b][0]), (k, T, eos, b, got[b], want[b][0])
            assert abs(scores[b] - want[b][1]) < 2e-3 * max(1.0, abs(want[b][1]))
        # one prompt alone = the same prompt in the batch
        alone = to_np(eng.generate_beam(prompts[2:3], emb, grids[2:3], T, k, eos_token_id=eos, pad_token_id=0, img_rows=rows[2:3])).astype(int)
        assert np.array_equal(alone[0], got[2])
    # and the search is not the greedy path: with beams the best hypothesis differs from the argmax chain on at least one prompt ... or
    # scores at least as well as it (the greedy chain is one of the candidates while it stays among the k best)
    assert got.shape == (len(prompts), T)


def _penalised_replay(step_logits, prompt_ids, tokens, penalty, eos=None):
    """The greedy loop replayed on the host from the engine's own RAW step logits [T, V] (bf16 values): every token must be the
    lowest-index argmax of `oracle.repetition_penalty_scores` over the prompt ids + the tokens fed before it."""
    hist = [int(t) for t in prompt_ids]
    for j in range(step_logits.shape[0]):
        scores = Q.repetition_penalty_scores(step_logits[j], hist, penalty)
        want = int(np.argmax(scores))
        assert int(tokens[j]) == want, (j, int(tokens[j]), want)
        if eos is not None and want == eos:
            return j + 1
        hist.append(want)
    return step_logits.shape[0]


def test_repetition_penalty_matches_hf_processor_semantics(setup, gpu):
    """Round 6.  HF's RepetitionPenaltyLogitsProcessor is in force in the reference whenever the checkpoint's generation_config.json has
    `repetition_penalty` != 1 (tests/test_oracle_sampling.py pins that HF merges it into the reference's greedy `generate` call, and
    pins the oracle's restatement on HF's processor).  Engine: `owc_llm_set_repetition_penalty` - a seen-token bitmap per cache slot
    marked from the prompt rows at prefill and from the fed token at every decode step, applied in fp32 inside the argmax.
    Asserted: every token of every row == the host replay of the penalised greedy loop on the engine's own raw step logits (exact);
    the penalty really decides tokens on this model; batched == alone; shared-prefix prefill == plain prefill; row compaction and
    EOS change nothing; the option does not leak into the next (un-penalised) call."""
    from lmms_owc_amd.engine.qwen2vl import Qwen2VLEngine

    cfg, w, eng, g = setup
    T, p = 10, 1.3
    cases = []
    for case in ("a", "b"):
        grid = [tuple(r) for r in g[f"{case}_grid"].tolist()]
        cases.append((g[f"{case}_ids"], grid, recipes.pixel_values(grid, 7)))
    cases.append((recipes.prompt_ids(cfg, [(1, 4, 4)], seed=99), [(1, 4, 4)], recipes.pixel_values([(1, 4, 4)], 8)))
    pix_all = np.concatenate([c[2] for c in cases])
    grids_all = [gg for c in cases for gg in c[1]]
    emb = eng.encode_images(torch.from_numpy(pix_all).to(torch.bfloat16).to(gpu), grids_all)
    prompts, grids = [c[0] for c in cases], [c[1] for c in cases]
    plain = to_np(eng.generate(prompts, emb, grids, T)).astype(int)
    toks, sl = eng.generate(prompts, emb, grids, T, repetition_penalty=p, return_step_logits=True)
    toks, sl = to_np(toks).astype(int), to_np(sl)
    for b in range(len(cases)):
        _penalised_replay(sl[:, b], prompts[b], toks[b], p)
    assert (toks != plain).any(), "the penalty decided no token on this model: the test would not see a missing penalty"
    assert np.array_equal(to_np(eng.generate(prompts, emb, grids, T)).astype(int), plain), "the option leaked into the next call"
    # alone == in the batch
    off = 0
    for b, (ids, grid, pix) in enumerate(cases):
        n = sum(t * h * w_ // 4 for t, h, w_ in grid)
        alone = to_np(eng.generate([ids], emb[off:off + n], [grid], T, repetition_penalty=p))[0].astype(int)
        off += n
        assert np.array_equal(alone, toks[b]), (b, alone, toks[b])
    # the engine attribute is the default (what a plug-in sets from generation_config.json)
    eng.repetition_penalty = p
    try:
        assert np.array_equal(to_np(eng.generate(prompts, emb, grids, T)).astype(int), toks)
    finally:
        eng.repetition_penalty = 1.0
    # shared-prefix prefill (prefix rows marked in every slot of the launch) == plain prefill
    r = np.random.default_rng(5)
    head = r.integers(10, 400, 7)
    text = [np.concatenate([head, r.integers(10, 400, 3 + i)]).astype(np.int64) for i in range(5)]
    none = [[] for _ in text]
    shared = Qwen2VLEngine(eng.w, vit_chunk_tokens=64, prefill_chunk_tokens=4096, share_prefix=True, min_shared_prefix=2)
    unshared = Qwen2VLEngine(eng.w, vit_chunk_tokens=64, prefill_chunk_tokens=4096, share_prefix=False)
    a = to_np(shared.generate(text, None, none, T, repetition_penalty=p))
    b_ = to_np(unshared.generate(text, None, none, T, repetition_penalty=p))
    assert np.array_equal(a, b_)
    # EOS + row compaction: the bitmap lives with the cache SLOT, so dropping finished rows changes nothing
    eos = int(a[0, 3])
    kw = dict(eos_token_id=eos, pad_token_id=0, repetition_penalty=p)
    c1 = to_np(unshared.generate(text, None, none, T, compact_rows=False, **kw))
    c2 = to_np(unshared.generate(text, None, none, T, compact_rows=True, **kw))
    assert np.array_equal(c1, c2)
    for i in range(len(text)):
        stop = np.flatnonzero(a[i] == eos)
        n_keep = (stop[0] + 1) if len(stop) else T
        assert np.array_equal(c1[i, :n_keep], a[i, :n_keep]) and (c1[i, n_keep:] == 0).all()
    # the hipGraph decode path replays the captured step: the mark / penalised-argmax launches are part of it
    gr = Qwen2VLEngine(eng.w, vit_chunk_tokens=64, prefill_chunk_tokens=4096, share_prefix=False, graph_decode=True)
    assert np.array_equal(to_np(gr.generate(text, None, none, T, repetition_penalty=p)), b_)
    # sampled requests: penalised values written back as bf16 in front of the draw - runs, is a function of the seed
    smp = {"temperature": 0.8, "top_k": 20, "top_p": 0.9, "seed": 7}
    s1 = to_np(unshared.generate(text, None, none, T, sampling=smp, repetition_penalty=p))
    s2 = to_np(unshared.generate(text, None, none, T, sampling=smp, repetition_penalty=p))
    assert np.array_equal(s1, s2) and s1.min() >= 0 and s1.max() < 512


def test_repetition_penalty_against_hf_generate_golden(setup, gpu):
    """The engine against HF ITSELF under a repetition penalty (tests/golden/qwen2vl_tiny_rep.npz: HF's generate with 1.3 on the
    model's generation config and the reference's argument list; prompts on which the seeded model loops without the penalty).
    Teacher-forced on HF's bf16 tokens, so every step is comparable: the engine's RAW logits stay within the model-level bound of
    HF's raw logits (2.5 % of max |logit|), and its token - the argmax of its own penalised scores over the same history - equals
    HF's wherever HF's PROCESSED top-2 margin exceeds twice that bound; on a near-tie it must still be one of the candidates.  The
    un-penalised engine run differs (it loops like HF's un-penalised run)."""
    import json

    cfg, w, eng, _ = setup
    g = np.load(GOLD / "qwen2vl_tiny_rep.npz")
    meta = json.loads((GOLD / "qwen2vl_tiny_rep.json").read_text())
    p, T = meta["repetition_penalty"], 12
    decisive = 0
    for name, case in meta["cases"].items():
        grid = [tuple(r) for r in case["grid"]]
        pix = recipes.pixel_values(grid, 7)
        ids = g[f"{name}_ids"]
        emb = eng.encode_images(torch.from_numpy(pix).to(torch.bfloat16).to(gpu), grid)
        hf_toks, hf_raw, hf_proc = g[f"{name}_bf16_tokens"], g[f"{name}_bf16_logits"], g[f"{name}_bf16_scores"]
        toks, sl = eng.generate([ids], emb, [grid], T, repetition_penalty=p, forced_tokens=hf_toks[None], return_step_logits=True)
        toks, sl = to_np(toks)[0].astype(int), to_np(sl)[:, 0]
        for j in range(T):
            scale = np.abs(hf_raw[j]).max()
            assert np.abs(sl[j] - hf_raw[j]).max() <= 0.025 * scale, (name, j, np.abs(sl[j] - hf_raw[j]).max() / scale)
            top2 = np.sort(hf_proc[j])[-2:]
            if top2[1] - top2[0] > 0.05 * scale:
                assert toks[j] == int(hf_toks[j]), (name, j, toks[j], int(hf_toks[j]))
                decisive += 1
            else:
                assert hf_proc[j][toks[j]] >= top2[1] - 0.05 * scale, (name, j)
        free = to_np(eng.generate([ids], emb, [grid], T))[0].astype(int)
        if not np.array_equal(g[f"{name}_bf16_tokens"], g[f"{name}_bf16_tokens_without_penalty"]):
            assert not np.array_equal(free, toks), name      # without the penalty the engine does what HF does without it: something else
    assert decisive >= 12, decisive
