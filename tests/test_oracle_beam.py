"""Pin oracle/beam_np.py on HF's OWN beam search: `generate(num_beams=k, do_sample=False)` of a tiny seeded Qwen2 decoder (fp32, CPU)
against `beam_np.beam_search` fed by that model's forward - the call the reference makes with `num_beams` from a request's gen_kwargs
(/root/reference/src/models/_qwen2_vl.py:308-329).  Cases: 2 / 3 / 4 beams, an EOS id the beams do reach (hypotheses finish early
and compete length-penalised), one they never reach (every hypothesis ends at the length limit)."""
import numpy as np
import pytest

from oracle import beam_np as BM


@pytest.fixture(scope="module")
def tiny_lm():
    import torch
    from transformers import Qwen2Config, Qwen2ForCausalLM

    torch.manual_seed(7)
    cfg = Qwen2Config(vocab_size=97, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, max_position_embeddings=128, tie_word_embeddings=False)
    m = Qwen2ForCausalLM(cfg).eval().float()
    with torch.no_grad():
        for p in m.parameters():          # wider logits than the default init: beams of different quality
            p.mul_(3.0)
    return m


def _hf_beams(m, prompt, k, T, eos, pad, **kw):
    import torch

    with torch.no_grad():
        out = m.generate(input_ids=torch.tensor([prompt]), attention_mask=torch.ones(1, len(prompt), dtype=torch.long), num_beams=k,
                         do_sample=False, max_new_tokens=T, eos_token_id=eos, pad_token_id=pad, use_cache=True, **kw)
    new = out[0, len(prompt):].tolist()
    return np.array(new + [pad] * (T - len(new)))


@pytest.mark.parametrize("k,T,eos_rank", [(2, 6, None), (3, 8, 0), (4, 8, 1), (3, 10, 2), (2, 5, 0)])
def test_beam_search_matches_hf_generate(tiny_lm, k, T, eos_rank):
    import torch

    m = tiny_lm
    r = np.random.default_rng(100 * k + T)
    pad = 0
    for trial in range(4):
        prompt = r.integers(1, 97, 5 + trial).tolist()

        def logits_fn(conts):
            with torch.no_grad():
                ids = torch.tensor([prompt + c for c in conts])
                return m(input_ids=ids).logits[:, -1, :].float().numpy()

        # an EOS id the search meets: the eos_rank-th most frequent token of an EOS-free beam run (None: an id that never wins)
        free, _ = BM.beam_search(logits_fn, len(prompt), k, T, -1, pad)
        if eos_rank is None:
            eos = 96 if 96 not in free else 95
        else:
            vals, counts = np.unique(free, return_counts=True)
            eos = int(vals[np.argsort(-counts, kind="stable")][min(eos_rank, len(vals) - 1)])
        want = _hf_beams(m, prompt, k, T, eos, pad)
        got, _ = BM.beam_search(logits_fn, len(prompt), k, T, eos, pad)
        assert np.array_equal(got, want), (k, T, eos, trial, got, want)


def test_product_bookkeeping_equals_the_oracle_on_a_batch():
    """`lmms_owc_amd.engine.beam.BeamSearcher` (the product's host side: B prompts at once, fed per running row with the log-sum-exp
    and the 2 x num_beams best logits - what `owc_beam_candidates` returns) against `beam_np.beam_search` prompt by prompt, on a
    synthetic "model" (logits = a seeded function of the sequence), with EOS ids the search meets: same tokens, same scores - also
    for prompts whose search ends steps before the others'."""
    from lmms_owc_amd.engine.beam import BeamSearcher

    V, B, T = 61, 7, 9
    rng = np.random.default_rng(5)
    table = rng.normal(size=(V, V, V)).astype(np.float32) * 2.5

    def logits_of(prompt_seed: int, seq: list) -> np.ndarray:
        a = (prompt_seed * 7 + len(seq)) % V
        b = seq[-1] if seq else prompt_seed % V
        c = seq[-2] if len(seq) > 1 else (prompt_seed * 3) % V
        return table[a, b] + 0.5 * table[c, a]

    for k, eos in ((2, 11), (3, 40), (4, 7)):
        want = [BM.beam_search(lambda conts, s=s: np.stack([logits_of(s, c) for c in conts]), 0, k, T, eos, 0) for s in range(B)]
        bs = BeamSearcher(B, k, T, eos, 0)
        seqs = [[[] for _ in range(k)] for _ in range(B)]
        more, steps = True, 0
        while more:
            lg = np.stack([np.stack([logits_of(s, seqs[s][j]) for j in range(k)]) for s in range(B)])       # [B, k, V]
            order = np.lexsort((np.broadcast_to(np.arange(V), lg.shape), -lg.astype(np.float64)), axis=-1)[..., : 2 * k]
            top_val = np.take_along_axis(lg, order, -1)
            m = lg.max(-1)
            logz = m + np.log(np.exp(lg - m[..., None]).sum(-1, dtype=np.float32))
            parent, token, more = bs.step(logz, top_val, order.astype(np.int32))
            seqs = [[seqs[s][parent[s, j]] + [int(token[s, j])] for j in range(k)] for s in range(B)]
            steps += 1
        toks, scores = bs.result()
        for s in range(B):
            assert np.array_equal(toks[s], want[s][0]), (k, eos, s, toks[s], want[s][0])
            assert abs(scores[s] - want[s][1]) < 1e-4
        assert steps <= T


@pytest.mark.parametrize("k,T,lp,es", [(3, 8, 0.5, False), (3, 8, 2.0, False), (4, 9, 1.0, True), (2, 7, 1.5, True), (3, 8, 1.0, "never")])
def test_length_penalty_and_early_stopping_match_hf(tiny_lm, k, T, lp, es):
    """`generate_beam` exposes HF's `length_penalty` and `early_stopping` (the reference leaves both at HF's defaults): the oracle and
    the product's `BeamSearcher` follow HF for other values as well - finished hypotheses scored by length ** penalty, the search ended
    as soon as num_beams hypotheses are finished (True), by the attainable-score heuristic (False) or its optimistic form ("never")."""
    import torch

    from lmms_owc_amd.engine.beam import BeamSearcher

    m = tiny_lm
    r = np.random.default_rng(7 * k + T)
    pad = 0
    for trial in range(3):
        prompt = r.integers(1, 97, 6 + trial).tolist()

        def logits_fn(conts):
            with torch.no_grad():
                return m(input_ids=torch.tensor([prompt + c for c in conts])).logits[:, -1, :].float().numpy()

        free, _ = BM.beam_search(logits_fn, len(prompt), k, T, -1, pad)
        vals, counts = np.unique(free, return_counts=True)
        eos = int(vals[np.argsort(-counts, kind="stable")][0])
        want = _hf_beams(m, prompt, k, T, eos, pad, length_penalty=lp, early_stopping=es)
        got, score = BM.beam_search(logits_fn, len(prompt), k, T, eos, pad, length_penalty=lp, early_stopping=es)
        assert np.array_equal(got, want), (k, T, lp, es, trial, got, want)
        # the product's bookkeeping, fed with the same logits the way the device feeds it
        bs = BeamSearcher(1, k, T, eos, pad, lp, es)
        seqs, more = [[] for _ in range(k)], True
        while more:
            lg = logits_fn(seqs)[None]                                                        # [1, k, V]
            order = np.lexsort((np.broadcast_to(np.arange(lg.shape[-1]), lg.shape), -lg.astype(np.float64)), axis=-1)[..., : 2 * k]
            mx = lg.max(-1)
            logz = mx + np.log(np.exp(lg - mx[..., None]).sum(-1, dtype=np.float32))
            parent, token, more = bs.step(logz, np.take_along_axis(lg, order, -1), order.astype(np.int32))
            seqs = [seqs[parent[0, j]] + [int(token[0, j])] for j in range(k)]
        toks, scores = bs.result()
        assert np.array_equal(toks[0], want) and abs(scores[0] - score) < 1e-4


@pytest.mark.parametrize("k,T,lp,trial,eos_rank", [(2, 10, 1.0, 3, 0), (2, 10, 3.0, 3, 1), (2, 12, 2.0, 2, 0), (3, 10, 1.0, 1, 0), (3, 10, 2.0, 3, 0)])
def test_early_stopping_never_prices_running_beams_at_the_longest_length(tiny_lm, k, T, lp, trial, eos_rank):
    """HF's `_check_early_stop_heuristic` divides the best running score by (max_length - prompt_len) ** length_penalty
    when `early_stopping == "never"` and length_penalty > 0, by the CURRENT length otherwise; the oracle and `BeamSearcher` used the
    current length for both and round 5's test cases did not separate the two.  These cases do (found by scanning: HF's "never"
    result differs from its `early_stopping=False` result), and both implementations must give HF's "never" result."""
    import torch

    from lmms_owc_amd.engine.beam import BeamSearcher

    m, pad = tiny_lm, 0
    r = np.random.default_rng(1000 * k + 10 * T + int(lp))
    for t in range(trial + 1):
        prompt = r.integers(1, 97, 5 + t).tolist()

    def logits_fn(conts):
        with torch.no_grad():
            return m(input_ids=torch.tensor([prompt + c for c in conts])).logits[:, -1, :].float().numpy()

    free, _ = BM.beam_search(logits_fn, len(prompt), k, T, -1, pad)
    vals, counts = np.unique(free, return_counts=True)
    eos = int(vals[np.argsort(-counts, kind="stable")][min(eos_rank, len(vals) - 1)])
    want = _hf_beams(m, prompt, k, T, eos, pad, length_penalty=lp, early_stopping="never")
    other = _hf_beams(m, prompt, k, T, eos, pad, length_penalty=lp, early_stopping=False)
    assert not np.array_equal(want, other), "this case no longer separates the two heuristics"
    got, score = BM.beam_search(logits_fn, len(prompt), k, T, eos, pad, length_penalty=lp, early_stopping="never")
    assert np.array_equal(got, want), (got, want)
    bs = BeamSearcher(1, k, T, eos, pad, lp, "never")
    seqs, more = [[] for _ in range(k)], True
    while more:
        lg = logits_fn(seqs)[None]
        order = np.lexsort((np.broadcast_to(np.arange(lg.shape[-1]), lg.shape), -lg.astype(np.float64)), axis=-1)[..., : 2 * k]
        mx = lg.max(-1)
        logz = mx + np.log(np.exp(lg - mx[..., None]).sum(-1, dtype=np.float32))
        parent, token, more = bs.step(logz, np.take_along_axis(lg, order, -1), order.astype(np.int32))
        seqs = [seqs[parent[0, j]] + [int(token[0, j])] for j in range(k)]
    toks, scores = bs.result()
    assert np.array_equal(toks[0], want) and abs(scores[0] - score) < 1e-4
