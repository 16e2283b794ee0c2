"""Model-level parity of the fp8 decoder (BASELINE.json config #5), tiny Qwen2-VL config with the real structure.

Redefined parity (no reference fp8 path):
* HIP fp8 engine vs the numpy fp8 oracle (same quantiser, same weights): logits within 10 % of max |logit|, greedy tokens
  equal wherever the oracle's top-2 margin exceeds twice that bound.  (Looser than the bf16 model's 3 %: a one-ulp bf16
  difference in an activation can flip its e4m3 code, a 6 % step on that element, so upstream summation-order noise is
  amplified; the op-level tests in tests/test_fp8_gpu.py hold the kernels to exact codes / 1 bf16 ulp on identical inputs.)
* fp8 engine vs the bf16 engine on the same weights: logits within 15 % of max |logit| (mean <= 3 %; measured 9.4 % / 1.8 %: two
  e4m3 operands per product, 7 projections x 2 layers) and top-1 agreement >= 75 % (measured 91 %) over 32 random prompts x 4 steps on this random-weight miniature
  (random logits are near-ties far more often than a trained model's).
"""
import numpy as np
import pytest
import torch

from oracle import fp8_np as F
from oracle import qwen2vl_np as Q
from tests import recipes
from tests.util import to_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(gpu):
    from lmms_owc_amd.engine.qwen2vl import Qwen2VLDims, Qwen2VLEngine, Qwen2VLWeights

    cfg = recipes.tiny_cfg()
    w = recipes.qwen2vl_weights(cfg, 1234)
    kw = dict(v_depth=2, v_embed=160, v_heads=2, v_mlp=640, n_layers=2, d_model=256, n_q_heads=2, n_kv_heads=1, d_ff=512, vocab=512,
              tie_embeddings=False, image_token_id=500, max_positions=512, max_grid=64)
    e8 = Qwen2VLEngine(Qwen2VLWeights.from_state_dict(Qwen2VLDims(**kw, decoder_dtype="fp8"), w, gpu))
    e16 = Qwen2VLEngine(Qwen2VLWeights.from_state_dict(Qwen2VLDims(**kw), w, gpu))
    return cfg, w, e8, e16


def test_fp8_generate_matches_fp8_oracle(setup, gpu):
    cfg, w, e8, _ = setup
    fp8 = F.quantize_decoder(w, Q.T, cfg.text.num_hidden_layers)
    grid = [(1, 4, 4)]
    pix = recipes.pixel_values(grid, 7)
    ids = recipes.prompt_ids(cfg, grid, seed=3)
    emb = e8.encode_images(torch.from_numpy(pix).to(torch.bfloat16).to(gpu), grid)
    toks, logits = e8.generate([ids], emb, [grid], 6, return_logits=True)
    toks, logits = to_np(toks)[0].astype(int), to_np(logits)[0]
    o_toks, o_logits = Q.generate(w, cfg, ids, pix, grid, 6, bf16=True, return_logits=True, fp8=fp8)
    assert np.abs(logits - o_logits[0]).max() <= 0.10 * np.abs(o_logits[0]).max()
    assert np.abs(logits - o_logits[0]).mean() <= 0.02 * np.abs(o_logits[0]).max()
    # free-running tokens up to the first near-tie; every step under teacher forcing: tests/test_decode_parity_gpu.py
    margins = [np.sort(o_logits[j])[-1] - np.sort(o_logits[j])[-2] > 0.20 * np.abs(o_logits[j]).max() for j in range(6)]
    n_sure = margins.index(False) if False in margins else 6
    assert np.array_equal(toks[:n_sure], o_toks[:n_sure]), (toks, o_toks)


def test_fp8_vs_bf16_agreement(setup, gpu):
    cfg, w, e8, e16 = setup
    r = np.random.default_rng(0)
    prompts = [r.integers(1, 490, 12 + (i % 7)).astype(np.int64) for i in range(32)]
    t8, l8 = e8.generate(prompts, None, [[] for _ in prompts], 4, return_logits=True)
    t16, l16 = e16.generate(prompts, None, [[] for _ in prompts], 4, return_logits=True)
    l8, l16 = to_np(l8), to_np(l16)
    print('fp8 vs bf16: max rel', np.abs(l8 - l16).max() / np.abs(l16).max(), 'mean rel', np.abs(l8 - l16).mean() / np.abs(l16).max())
    assert np.abs(l8 - l16).max() <= 0.15 * np.abs(l16).max(), np.abs(l8 - l16).max() / np.abs(l16).max()
    assert np.abs(l8 - l16).mean() <= 0.03 * np.abs(l16).max()
    agree = (to_np(t8)[:, 0] == to_np(t16)[:, 0]).mean()   # first token: same context on both sides
    print('top-1 agreement', agree)
    assert agree >= 0.75, agree
    # batching invariance holds for the fp8 decoder too (per-token scales do not see the batch)
    single = to_np(e8.generate([prompts[5]], None, [[]], 4))
    assert np.array_equal(single[0], to_np(t8)[5])


def test_fp8_pruned_last_prefill_layer_is_bit_identical(setup, gpu):
    """Per-token quantisation is row-local, so running the last prefill layer on the last-token rows only must not change a bit
    of the fp8 decoder's logits or tokens either."""
    from lmms_owc_amd import _lib

    cfg, w, e8, _ = setup
    r = np.random.default_rng(11)
    prompts = [r.integers(1, 490, 9 + 3 * i).astype(np.int64) for i in range(5)]
    lib = _lib.load()
    try:
        assert lib.owc_tuning_set(b"prefill_prune_last", 0) == 0
        a, la = e8.generate(prompts, None, [[] for _ in prompts], 4, return_logits=True)
        assert lib.owc_tuning_set(b"prefill_prune_last", 1) == 0
        b, lb = e8.generate(prompts, None, [[] for _ in prompts], 4, return_logits=True)
        assert torch.equal(la, lb) and torch.equal(a, b)
    finally:
        lib.owc_tuning_set(b"prefill_prune_last", 1)
