"""Beam-search bookkeeping for the batched decoder (host side; the arithmetic - logits, log-sum-exp, per-row top candidates - is
in libowc_hip.so: `owc_llm_decode_step`, `owc_beam_candidates`).

The reference passes `num_beams` from a request's gen_kwargs to HF's `generate` (/root/reference/src/models/_qwen2_vl.py:308-329,
_llava_hf.py:365-376) with everything else at HF's defaults: `do_sample=False`, `length_penalty=1.0`, `early_stopping=False`, one
returned sequence.  What that means per step, for every prompt on its own (`GenerationMixin._beam_search`):

* the `num_beams` running hypotheses are extended by one token; of all beams x vocabulary continuations the best 2 x num_beams by
  accumulated log-probability survive (per beam its own best 2 x num_beams suffice to find them);
* a continuation that emits EOS or reaches `max_new_tokens` stops: it does not run on, and if it ranks among the first `num_beams`
  it competes for one of `num_beams` finished slots with score = accumulated log-probability / generated length ** length_penalty;
* the best `num_beams` continuations that did not stop are the next running beams (each names its parent: the KV cache follows);
* the search of a prompt ends when no running beam can still beat the worst finished hypothesis (its score / current length **
  length_penalty is not better) with all finished slots taken, or when every continuation stopped; the answer is the best
  finished hypothesis.

`BeamSearcher` keeps that state for B prompts at once (numpy, [B, num_beams]); a prompt whose search has ended keeps its rows in
the decode batch (fed the pad token) until all are done."""

from __future__ import annotations

import numpy as np

NEG = np.float32(-1.0e9)


class BeamSearcher:
    def __init__(self, n_prompts: int, num_beams: int, max_new_tokens: int, eos_token_id: int, pad_token_id: int,
                 length_penalty: float = 1.0, early_stopping=False):
        B, k, T = n_prompts, num_beams, max_new_tokens
        self.B, self.k, self.T = B, k, T
        self.eos, self.pad = eos_token_id, pad_token_id
        self.lp, self.early = float(length_penalty), early_stopping
        self.running = np.full((B, k, T), pad_token_id, np.int64)
        self.run_scores = np.full((B, k), NEG, np.float32)
        self.run_scores[:, 0] = 0.0                       # only beam 0 counts at the first step: the k beams start identical
        self.fin = np.full((B, k, T), pad_token_id, np.int64)
        self.fin_scores = np.full((B, k), NEG, np.float32)
        self.is_fin = np.zeros((B, k), bool)
        self.unsat = np.ones(B, bool)
        self.active = np.ones(B, bool)
        self.g = 0                                         # tokens generated so far

    @staticmethod
    def _order(scores: np.ndarray, tie: np.ndarray, n: int) -> np.ndarray:
        """Per row: the n best columns, descending score, lowest `tie` first among equals."""
        return np.lexsort((tie, -scores.astype(np.float64)), axis=-1)[..., :n]

    def step(self, logz: np.ndarray, top_val: np.ndarray, top_idx: np.ndarray):
        """One step.  For every running row (prompt b, beam j): `logz[b, j]` = log-sum-exp of its logits, `top_val / top_idx[b, j, :]`
        its 2 x num_beams largest logits (descending, lowest token id first among equals) and their token ids.  Returns
        (parent [B, k], token [B, k], more): running beam j of prompt b continues beam parent[b, j] with token[b, j]."""
        B, k, g = self.B, self.k, self.g
        K2 = 2 * k
        vocab_tie = top_idx.astype(np.int64) + (np.arange(k, dtype=np.int64)[None, :, None] << 32)     # flat (beam, token) order
        acc = (self.run_scores[:, :, None] + (top_val.astype(np.float32) - logz.astype(np.float32)[:, :, None])).astype(np.float32)
        flat = acc.reshape(B, k * K2)
        pick = self._order(flat, vocab_tie.reshape(B, k * K2), K2)                               # [B, 2k] into the k x 2k candidates
        rows = np.arange(B)[:, None]
        c_scores = flat[rows, pick]
        c_beam = pick // K2
        c_tok = top_idx.reshape(B, k * K2)[rows, pick].astype(np.int64)
        hits = (c_tok == self.eos) | (g + 1 >= self.T)
        # the next running beams: the best k continuations that did not stop
        r_scores = (c_scores + hits.astype(np.float32) * NEG).astype(np.float32)
        nxt = self._order(r_scores, np.broadcast_to(np.arange(K2), (B, K2)), k)
        # finished slots: the first k of the 2k only, length-penalised; closed once the prompt's search has ended
        did = hits & (np.arange(K2)[None, :] < k)
        f = (c_scores / np.float32((g + 1) ** self.lp)).astype(np.float32)
        full = self.is_fin.all(1) & (self.early is True)
        f = f + (full | ~self.unsat | ~self.active).astype(np.float32)[:, None] * NEG
        f = (f + (~did).astype(np.float32) * NEG).astype(np.float32)
        c_seqs = self.running[rows, c_beam].copy()                                                  # [B, 2k, T]
        c_seqs[:, :, g] = c_tok
        m_scores = np.concatenate([self.fin_scores, f], 1)
        m_seqs = np.concatenate([self.fin, c_seqs], 1)
        m_fin = np.concatenate([self.is_fin, did], 1)
        keep = self._order(m_scores, np.broadcast_to(np.arange(3 * k), (B, 3 * k)), k)
        act = self.active
        self.fin = np.where(act[:, None, None], m_seqs[rows, keep], self.fin)
        self.fin_scores = np.where(act[:, None], m_scores[rows, keep], self.fin_scores).astype(np.float32)
        self.is_fin = np.where(act[:, None], m_fin[rows, keep], self.is_fin)
        parent = np.where(act[:, None], c_beam[rows, nxt], np.arange(k)[None, :])
        token = np.where(act[:, None], c_tok[rows, nxt], self.pad)
        self.running = np.where(act[:, None, None], c_seqs[rows, nxt], self.running)
        self.run_scores = np.where(act[:, None], r_scores[rows, nxt], self.run_scores).astype(np.float32)
        self.g = g + 1
        # HF `_check_early_stop_heuristic`: early_stopping == "never" with a positive length penalty prices the running beam at the
        # LONGEST length it may reach (max_new_tokens), every other setting at its current length
        hyp_len = self.T if (self.early == "never" and self.lp > 0.0) else self.g
        best_possible = self.run_scores[:, 0] / np.float32(hyp_len ** self.lp)
        worst_fin = np.where(self.is_fin, self.fin_scores.min(1, keepdims=True), NEG)
        self.unsat = self.unsat & (best_possible[:, None] > worst_fin).any(1)
        open_beam = ~(self.is_fin.all(1) & (self.early is True))
        self.active = act & self.unsat & open_beam & ~hits.all(1)
        return parent, token, bool(self.active.any())

    def result(self) -> tuple[np.ndarray, np.ndarray]:
        """(tokens [B, T] of the best finished hypothesis per prompt, padded; their scores [B])."""
        return self.fin[:, 0, :].copy(), self.fin_scores[:, 0].copy()
