"""numpy restatement of beam search as the reference reaches it (TEST INFRASTRUCTURE ONLY; see oracle/__init__.py).

The reference hands `num_beams=gen_kwargs["num_beams"]` to HF's `generate` (/root/reference/src/models/_qwen2_vl.py:308-329,
_llava_hf.py:365-376; batch size 1, `do_sample=False` when temperature is 0, HF defaults `length_penalty=1.0`, `early_stopping=False`,
one returned sequence).  The arithmetic is transformers' `GenerationMixin._beam_search` (pinned 4.47.0 in the reference's uv.lock;
5.15.0 in the build container, the vectorised rewrite with the same semantics); `GM:` names its helper functions:

* every step the `num_beams` running hypotheses are extended by one token: log_softmax of the fp32 logits + the running score,
  the best 2 x num_beams continuations over ALL beams are kept (`GM:_get_top_k_continuations`);
* a continuation that ends in EOS or reaches the length limit "hits the stopping criteria": it cannot run on (its score gets -1e9
  for the choice of the next running beams, `GM:_get_running_beams_for_next_iteration`) and, IF it is among the first `num_beams` of
  the 2 x num_beams, it competes for the `num_beams` finished slots with its score divided by (generated length ** length_penalty)
  (`GM:_update_finished_beams`);
* after the step: unless a running beam's best attainable score (its score / generated length ** length_penalty, the
  `early_stopping=False` heuristic) still beats the worst finished one - or a finished slot is still empty - nothing can improve
  and the search ends (`GM:_check_early_stop_heuristic`, `GM:_beam_search_has_unfinished_sequences`); it also ends when every
  continuation hit the stopping criteria (the length limit);
* the answer is the best finished hypothesis, padded with `pad_token_id`.

Pinned by tests/test_oracle_beam.py: HF's own `generate(num_beams=k)` on a tiny seeded model, with this function fed by that
model's forward."""

from __future__ import annotations

import numpy as np

NEG = -1.0e9


def log_softmax(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, np.float32)
    m = x.max(-1, keepdims=True)
    return (x - m - np.log(np.exp(x - m).sum(-1, keepdims=True, dtype=np.float32))).astype(np.float32)


def topk_desc(v: np.ndarray, k: int) -> np.ndarray:
    """Indices of the k largest values, descending, lowest index first among equals."""
    return np.lexsort((np.arange(len(v)), -np.asarray(v, np.float64)))[:k]


def beam_search(logits_fn, prompt_len: int, num_beams: int, max_new_tokens: int, eos_token_id: int, pad_token_id: int,
                length_penalty: float = 1.0, early_stopping=False) -> tuple[np.ndarray, float]:
    """One prompt.  `logits_fn(list of num_beams generated-token lists) -> float32 [num_beams, vocab]` = the next-token logits of
    prompt + each continuation.  Returns (generated tokens of the best hypothesis padded to max_new_tokens, its score)."""
    k = num_beams
    running = [[] for _ in range(k)]
    run_scores = np.full(k, NEG, np.float32)
    run_scores[0] = 0.0
    fin = [[] for _ in range(k)]
    fin_scores = np.full(k, NEG, np.float32)
    is_fin = np.zeros(k, bool)
    unsat = True
    g = 0                                        # tokens generated so far (HF: cur_len - decoder_prompt_len)
    while True:
        logp = log_softmax(logits_fn(running)) + run_scores[:, None]
        V = logp.shape[1]
        top = topk_desc(logp.reshape(-1), 2 * k)
        c_scores = logp.reshape(-1)[top].astype(np.float32)
        c_seqs = [running[i // V] + [int(i % V)] for i in top]
        hits = np.array([s[-1] == eos_token_id or g + 1 >= max_new_tokens for s in c_seqs])
        # next running beams: the best k continuations that did not stop
        r_scores = (c_scores + hits.astype(np.float32) * np.float32(NEG)).astype(np.float32)
        nxt = topk_desc(r_scores, k)
        # finished slots: only the first k of the 2k may enter, length-penalised
        did = hits & (np.arange(2 * k) < k)
        f = (c_scores / np.float32((g + 1) ** length_penalty)).astype(np.float32)
        if bool(is_fin.all()) and early_stopping is True:
            f = f + np.float32(NEG)
        if not unsat:
            f = f + np.float32(NEG)
        f = f + (~did).astype(np.float32) * np.float32(NEG)
        m_scores = np.concatenate([fin_scores, f])
        m_seqs = fin + c_seqs
        m_fin = np.concatenate([is_fin, did])
        keep = topk_desc(m_scores, k)
        fin, fin_scores, is_fin = [m_seqs[i] for i in keep], m_scores[keep].astype(np.float32), m_fin[keep]
        running, run_scores = [c_seqs[i] for i in nxt], r_scores[nxt]
        g += 1
        # HF `_check_early_stop_heuristic`: early_stopping == "never" with length_penalty > 0 uses max_length - prompt_len, else the current length
        hyp_len = max_new_tokens if (early_stopping == "never" and length_penalty > 0.0) else g
        best_possible = run_scores[0] / np.float32(hyp_len ** length_penalty)
        worst_fin = np.where(is_fin, fin_scores.min(), np.float32(NEG))
        unsat = unsat and bool((best_possible > worst_fin).any())
        open_beam = not (bool(is_fin.all()) and early_stopping is True)
        if not (unsat and open_beam and not bool(hits.all())):
            break
    out = np.full(max_new_tokens, pad_token_id, np.int64)
    out[: len(fin[0])] = fin[0]
    return out, float(fin_scores[0])
