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
