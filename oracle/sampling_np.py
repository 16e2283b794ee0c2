"""numpy restatement of HF's sampling distribution (TEST INFRASTRUCTURE ONLY).

What `GenerationMixin._sample` draws from when the reference passes `do_sample = temperature > 0`
(/root/reference/src/models/_qwen2_vl.py:319-329, _llava_hf.py:365-376): the logits go to float32, then
TemperatureLogitsWarper (scores / T), TopKLogitsWarper (scores below the k-th largest VALUE are removed: ties at the cut stay),
TopPLogitsWarper (descending cumulative softmax mass; everything behind the point where it reaches top_p is removed, at least one
token stays), softmax, one multinomial draw (transformers generation/logits_process.py).  transformers is third-party and absent
from /root/reference; the three warpers are restated from its published behaviour and PINNED against the installed transformers'
own warpers in tests/test_oracle_sampling.py.  The draw itself is random: parity is the DISTRIBUTION (and the support), not
torch's random stream.  Ties: top-k keeps a run of equal logits at its cut whole, exactly as HF does (`scores < k-th value` is
what it removes).  Where the top-p cut falls inside a run of EQUAL logits HF keeps a prefix of the run in torch.sort's order,
which is unspecified among equal values; this restatement (and the HIP kernel) keep the same NUMBER of them and take the lowest
token ids - pinned on the kept COUNT and the kept mass in tests/test_oracle_sampling.py.
"""

from __future__ import annotations

import numpy as np


def sampling_probs(logits: np.ndarray, temperature: float, top_k: int = 0, top_p: float | None = None) -> np.ndarray:
    """float [V] logits -> float64 [V] probabilities of the next token."""
    s = np.asarray(logits, np.float32).astype(np.float64) / float(temperature)
    if top_k and top_k < len(s):
        kth = np.sort(s)[-int(top_k)]
        s = np.where(s < kth, -np.inf, s)
    if top_p is not None and 0.0 < top_p < 1.0:
        p = np.exp(s - s.max())
        p /= p.sum()
        order = np.argsort(-s, kind="stable")
        cum = np.cumsum(p[order])
        first = int(np.searchsorted(cum, top_p, side="left"))      # the token at which the cumulative mass reaches top_p stays
        keep = np.zeros(len(s), bool)
        keep[order[:min(first, len(s) - 1) + 1]] = True             # stable order: lowest ids first inside a run of equal logits
        s = np.where(keep, s, -np.inf)
    p = np.exp(s - s.max())
    return p / p.sum()
