"""oracle/sampling_np.py against the installed transformers' own logits warpers (CPU): the same kept set and probabilities for
temperature / top-k / top-p, except inside a run of equal logits at the top-p cut (stated in the oracle's header)."""
import numpy as np
import pytest
import torch

from oracle import sampling_np as S


@pytest.mark.parametrize("T,k,p", [(0.7, 0, None), (1.3, 50, None), (1.0, 0, 0.9), (0.6, 40, 0.8), (2.0, 5, 0.5), (0.01, 1, 0.001)])
def test_distribution_matches_hf_warpers(T, k, p):
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper

    r = np.random.default_rng(int(T * 100) + k)
    for _ in range(4):
        logits = (r.standard_normal(4096) * 3).astype(np.float32)   # distinct values: no ties at the cuts
        scores = torch.from_numpy(logits)[None].clone()
        ids = torch.zeros((1, 1), dtype=torch.long)
        scores = TemperatureLogitsWarper(T)(ids, scores)
        if k:
            scores = TopKLogitsWarper(top_k=k)(ids, scores)
        if p is not None:
            scores = TopPLogitsWarper(top_p=p)(ids, scores)
        want = torch.softmax(scores[0].double(), -1).numpy()
        got = S.sampling_probs(logits, T, k, p)
        assert np.array_equal(got > 0, want > 0)
        assert np.abs(got - want).max() <= 1e-6


def test_top_k_keeps_ties_like_hf():
    from transformers.generation.logits_process import TopKLogitsWarper

    logits = np.array([1.0, 3.0, 3.0, 2.0, 3.0, 0.0], np.float32)
    want = torch.softmax(TopKLogitsWarper(top_k=2)(torch.zeros((1, 1), dtype=torch.long), torch.from_numpy(logits)[None].clone())[0].double(), -1)
    got = S.sampling_probs(logits, 1.0, 2, None)
    assert np.allclose(got, want.numpy()) and (got > 0).sum() == 3


def test_top_p_cut_inside_a_run_of_equal_logits_keeps_hf_count_and_mass():
    """Where the top-p cut falls inside a run of EQUAL logits, HF's TopPLogitsWarper keeps a prefix of the run in torch.sort's order
    (which of the equal tokens: unspecified).  The restatement keeps the same NUMBER of tokens - hence the same kept mass and the same
    probability for every kept token - and takes the lowest ids of the run."""
    from transformers.generation.logits_process import TopKLogitsWarper, TopPLogitsWarper

    ids = torch.zeros((1, 1), dtype=torch.long)
    r = np.random.default_rng(3)
    cases = [(np.array([2.0, 0.0, 2.0, 2.0, 2.0, 1.0, 2.0, -1.0], np.float32), 0, 0.5),      # five tied maxima, cut inside them
             (np.array([2.0, 0.0, 2.0, 2.0, 2.0, 1.0, 2.0, -1.0], np.float32), 0, 0.001),    # exactly one survivor
             (np.array([3.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.0], np.float32), 0, 0.8),       # cut inside the second run
             (np.array([1.0, 3.0, 3.0, 2.0, 3.0, 0.0], np.float32), 2, 0.6),                 # top-k keeps 3 ties, top-p cuts inside them
             (np.round(r.standard_normal(512) * 2).astype(np.float32), 40, 0.7)]             # integer logits: many runs
    for logits, k, p in cases:
        scores = torch.from_numpy(logits)[None].clone()
        if k:
            scores = TopKLogitsWarper(top_k=k)(ids, scores)
        scores = TopPLogitsWarper(top_p=p)(ids, scores)
        want = torch.softmax(scores[0].double(), -1).numpy()
        got = S.sampling_probs(logits, 1.0, k, p)
        assert (got > 0).sum() == (want > 0).sum(), (logits, k, p)
        assert np.allclose(np.sort(got), np.sort(want), atol=1e-9)            # the same multiset of probabilities
        for v in np.unique(logits):                                            # per value: the same count kept, ours = the lowest ids
            idx = np.flatnonzero(logits == v)
            n_kept = int((want[idx] > 0).sum())
            assert np.array_equal(np.flatnonzero(got[idx] > 0), np.arange(n_kept)), (v, n_kept)
    # top_k = 1 with top_p < 1 (Qwen2-VL's generation_config.json: top_k 1, top_p 0.001): one survivor = the argmax (lowest id among ties)
    logits = np.array([0.0, 5.0, 1.0, 5.0, 5.0], np.float32)
    got = S.sampling_probs(logits, 0.01, 1, 0.001)
    assert np.array_equal(got, np.array([0.0, 1.0, 0.0, 0.0, 0.0]))
    # top_k = 1 ALONE keeps every tied maximum: HF's own behaviour (`scores < k-th value` is what it removes)
    hf = TopKLogitsWarper(top_k=1)(ids, torch.from_numpy(logits)[None].clone())[0]
    assert torch.isfinite(hf).sum().item() == 3 and (S.sampling_probs(logits, 1.0, 1, None) > 0).sum() == 3


def test_repetition_penalty_restatement_matches_hf_processor_and_generate_merges_the_config():
    """(1) oracle `repetition_penalty_scores` == transformers' RepetitionPenaltyLogitsProcessor on float32 scores (duplicates in the
    history count once; negative scores are multiplied, positive ones divided).  (2) The claim the HIP path's default rests on: a
    `generate(do_sample=False, temperature=0, top_p=None, num_beams=1, ...)` call - the reference's call,
    /root/reference/src/models/_qwen2_vl.py:319-329 - on a model whose generation_config carries repetition_penalty != 1 DOES get the
    processor (HF merges the config fields the call does not pass), checked through `_get_logits_processor` of a tiny random GPT-2."""
    from transformers.generation.logits_process import RepetitionPenaltyLogitsProcessor

    from oracle import qwen2vl_np as Q

    r = np.random.default_rng(0)
    for penalty in (1.05, 1.3, 0.8):
        logits = (r.standard_normal(977) * 4).astype(np.float32)
        hist = r.integers(0, 977, 60)
        hist[10:20] = hist[0]                                   # duplicates
        want = RepetitionPenaltyLogitsProcessor(penalty=penalty)(torch.from_numpy(hist)[None], torch.from_numpy(logits)[None].clone())[0].numpy()
        got = Q.repetition_penalty_scores(logits, hist, penalty)
        assert np.array_equal(got, want)
        untouched = np.setdiff1d(np.arange(977), hist)
        assert np.array_equal(got[untouched], logits[untouched])
    from transformers import GenerationConfig, GPT2Config, GPT2LMHeadModel

    torch.manual_seed(3)
    m = GPT2LMHeadModel(GPT2Config(n_layer=1, n_head=2, n_embd=16, vocab_size=50, n_positions=32, bos_token_id=0, eos_token_id=49)).eval()
    m.generation_config = GenerationConfig(do_sample=True, temperature=0.01, top_p=0.001, top_k=1, repetition_penalty=1.3,
                                           eos_token_id=49, pad_token_id=0)
    ids = torch.tensor([[3, 4, 5, 3]])
    kw = dict(do_sample=False, temperature=0, top_p=None, num_beams=1, max_new_tokens=8, eos_token_id=49, pad_token_id=0)
    with torch.no_grad():
        base = m.generate(ids, **kw, output_logits=True, return_dict_in_generate=True)
        plain = m.generate(ids, **kw, repetition_penalty=1.0, output_logits=True, return_dict_in_generate=True)
    # replay the greedy loop by hand from the RAW logits of the penalised run: each chosen token is the argmax of the PENALISED scores
    seq = base.sequences[0].tolist()
    hist = ids[0].tolist()
    flips = 0
    for j, raw in enumerate(base.logits):
        scores = Q.repetition_penalty_scores(raw[0].float().numpy(), hist, 1.3)
        assert int(np.argmax(scores)) == seq[len(ids[0]) + j]
        flips += int(np.argmax(raw[0].float().numpy()) != seq[len(ids[0]) + j])
        hist.append(seq[len(ids[0]) + j])
        if seq[len(ids[0]) + j] == 49:
            break
    # the penalty decided tokens (this seeded model repeats itself when left alone), and passing 1.0 explicitly switches it off
    assert flips >= 1 and plain.sequences[0].tolist() != seq
