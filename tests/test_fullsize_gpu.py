"""Parity at BASELINE's full widths: a 2-block slice of the real vision tower (1280 x 16 heads x 5120) over
16 images of 448x448 (16384 patch rows: the 256x256 GEMM, fused-RoPE epilogue and hd-80 attention at their
production shapes) and a 2-layer slice of the Qwen2-VL-2B decoder (1536 / 12 q / 2 kv heads / 8960, S = 286)
against the bf16 numpy oracle, plus size-independent properties on the full 2B model."""
import numpy as np
import pytest
import torch

from oracle import qwen2vl_np as Q
from tests import recipes
from tests.util import to_np

pytestmark = pytest.mark.gpu


def _dims(**kw):
    from lmms_owc_amd.engine.qwen2vl import Qwen2VLDims

    base = dict(v_depth=2, v_embed=1280, v_heads=16, v_mlp=5120, n_layers=2, d_model=1536, n_q_heads=12, n_kv_heads=2,
                d_ff=8960, vocab=4096, tie_embeddings=False, image_token_id=4000, max_positions=1024, max_grid=64)
    base.update(kw)
    return Qwen2VLDims(**base)


@pytest.fixture(scope="module")
def slice_model(gpu):
    from lmms_owc_amd.engine.qwen2vl import Qwen2VLEngine, Qwen2VLWeights

    cfg = Q.Cfg(vision=Q.VisionCfg(depth=2, embed_dim=1280, num_heads=16, mlp_ratio=4.0, hidden_size=1536),
                text=Q.TextCfg(hidden_size=1536, num_hidden_layers=2, num_attention_heads=12, num_key_value_heads=2,
                               intermediate_size=8960, vocab_size=4096, tie_word_embeddings=False), image_token_id=4000)
    w = recipes.qwen2vl_weights(cfg, 4321)
    eng = Qwen2VLEngine(Qwen2VLWeights.from_state_dict(_dims(), w, gpu))
    return cfg, w, eng


def test_vision_slice_full_width(slice_model, gpu):
    cfg, w, eng = slice_model
    grid = [(1, 32, 32)] * 16
    pix = recipes.pixel_values(grid, 3)
    out = to_np(eng.encode_images(torch.from_numpy(pix).to(torch.bfloat16).to(gpu), grid))
    # the oracle on 2 of the 16 images (numpy time); images are independent
    for i in (0, 15):
        ref = Q.vit_forward(w, cfg, pix[i * 1024:(i + 1) * 1024], grid[:1], bf16=True)
        got = out[i * 256:(i + 1) * 256]
        assert np.abs(got - ref).max() <= 0.02 * np.abs(ref).max(), (i, np.abs(got - ref).max(), np.abs(ref).max())
        assert np.abs(got - ref).mean() <= 0.004 * np.abs(ref).max()


def test_vision_tower_bits_with_and_without_the_persistent_gemm(slice_model, gpu):
    """The vision tower's GEMMs at 16 384 patch rows (960 / 320 / 1280 output tiles: qkv with the rotary epilogue, proj and fc2 adding
    into the residual stream in place, fc1 with quick-GELU) run the persistent ping-pong kernel: the embeddings equal those of the
    one-tile-per-block kernel bit for bit, five times (the race screen at model level)."""
    from lmms_owc_amd import _lib

    cfg, w, eng = slice_model
    lib = _lib.load()
    grid = [(1, 32, 32)] * 16
    pix = torch.from_numpy(recipes.pixel_values(grid, 11)).to(torch.bfloat16).to(gpu)
    try:
        assert lib.owc_tuning_set(b"gemm_persist", 0) == 0
        want = eng.encode_images(pix, grid).clone()
        assert lib.owc_tuning_set(b"gemm_persist", 1) == 0
        for i in range(5):
            got = eng.encode_images(pix, grid)
            assert torch.equal(got, want), (i, (got != want).sum().item())
    finally:
        lib.owc_tuning_set(b"gemm_persist", -1)


def test_qwen25_vision_slice_full_width(gpu):
    """Qwen2.5-VL's vision tower at its real widths (registry names qwen2.5-vl-7b / -3b, /root/reference/src/models/_qwen2_vl.py:
    106-115, 635-648): hidden 1280 x 16 heads, gated MLP of 3420 (zero-padded to 3456 = 27 x 128 at load, so its down projection
    runs the ping-pong GEMM), 112-pixel windows, a 3-block slice whose MIDDLE block is a full-attention block, merger 5120 -> 2048
    (the 3B decoder's width).  18 300 patch rows in one launch group: twelve 448 x 448 images, two 36 x 28-patch images (504 x 392:
    both sides end in HALF windows) and one cap-size non-square image (54 x 74 patches = 756 x 1036 pixels: 6 3/4 x 9 1/4 windows, 3996
    keys in the full-attention block) - against oracle/qwen25vl_np.py (pinned on HF's Qwen2_5_VLForConditionalGeneration) under the
    2 % bound of the Qwen2-VL slice."""
    from lmms_owc_amd.engine.qwen2vl import Qwen2VLDims, Qwen2VLEngine, Qwen2VLWeights
    from oracle import qwen25vl_np as Q25

    cfg = Q25.Cfg25(vision=Q25.Vision25Cfg(depth=3, embed_dim=1280, num_heads=16, intermediate_size=3420, hidden_size=2048,
                                           window_size=112, fullatt_block_indexes=(1,)),
                    text=Q.TextCfg(hidden_size=2048, num_hidden_layers=1, num_attention_heads=16, num_key_value_heads=2,
                                   intermediate_size=256, vocab_size=512, tie_word_embeddings=False), image_token_id=500)
    w = recipes.qwen25vl_weights(cfg, 97)
    dims = Qwen2VLDims(v_variant=1, v_depth=3, v_embed=1280, v_heads=16, v_mlp=3420, v_fullatt=(1,), n_layers=1, d_model=2048, n_q_heads=16,
                       n_kv_heads=2, d_ff=256, vocab=512, tie_embeddings=False, image_token_id=500, max_positions=512, max_grid=128)
    eng = Qwen2VLEngine(Qwen2VLWeights.from_state_dict(dims, w, gpu))
    assert eng.w.vit.mlp_hidden == 3456
    grid = [(1, 32, 32)] * 6 + [(1, 36, 28)] + [(1, 32, 32)] * 6 + [(1, 54, 74), (1, 36, 28)]
    pix = recipes.pixel_values(grid, 8)
    out = to_np(eng.encode_images(torch.from_numpy(pix).to(torch.bfloat16).to(gpu), grid))
    starts = np.concatenate([[0], np.cumsum([g[1] * g[2] for g in grid])])
    assert starts[-1] == 18300 and out.shape == (18300 // 4, 2048)
    for i in (0, 6, 13, 14):   # a 448 x 448 image, both half-window images (mid-batch and last), the cap-size image
        ref = Q25.vit_forward(w, cfg, pix[starts[i]:starts[i + 1]], grid[i:i + 1], bf16=True)
        got = out[starts[i] // 4:starts[i + 1] // 4]
        assert np.abs(got - ref).max() <= 0.02 * np.abs(ref).max(), (i, grid[i], np.abs(got - ref).max(), np.abs(ref).max())
        assert np.abs(got - ref).mean() <= 0.004 * np.abs(ref).max(), (i, grid[i])


def test_decoder_slice_full_width(slice_model, gpu):
    cfg, w, eng = slice_model
    grid = [(1, 32, 32)]
    pix = recipes.pixel_values(grid, 5)
    r = np.random.default_rng(9)
    ids = np.concatenate([r.integers(1, 3900, 14), np.full(256, cfg.image_token_id), r.integers(1, 3900, 16)])
    emb = eng.encode_images(torch.from_numpy(pix).to(torch.bfloat16).to(gpu), grid)
    from tests.util import check_forced_steps

    o_toks, o_logits = Q.generate(w, cfg, ids, pix, grid, 4, bf16=True, return_logits=True)
    toks, logits = eng.generate([ids], emb, [grid], 4, forced_tokens=o_toks[None], return_step_logits=True)
    check_forced_steps(to_np(logits)[:, 0], to_np(toks)[0].astype(int), o_logits, o_toks, 0.02, "2b-width slice, S = 286")


def test_full_2b_properties(gpu):
    """Full Qwen2-VL-2B (random weights): batch invariance + determinism of greedy tokens, finite logits."""
    from lmms_owc_amd.engine.qwen2vl import DIMS, Qwen2VLEngine, Qwen2VLWeights

    d = DIMS["qwen2-vl-2b"]
    eng = Qwen2VLEngine(Qwen2VLWeights.random(d, gpu, seed=5))
    g = torch.Generator(device=gpu).manual_seed(1)
    n = 6
    pix = torch.randn((n * 1024, 1176), generator=g, device=gpu, dtype=torch.bfloat16)
    r = np.random.default_rng(2)
    ids = [np.concatenate([r.integers(1000, 150000, 10 + i), np.full(256, d.image_token_id), r.integers(1000, 150000, 12)])
           for i in range(n)]
    grids = [[(1, 32, 32)]] * n
    emb = eng.encode_images(pix, [(1, 32, 32)] * n)
    assert bool(torch.isfinite(emb.float()).all())
    batch, logits = eng.generate(ids, emb, grids, 6, return_logits=True)
    assert bool(torch.isfinite(logits.float()).all())
    again = eng.generate(ids, emb, grids, 6)
    assert torch.equal(batch, again)  # deterministic
    for b in (0, n - 1):
        single = eng.generate([ids[b]], emb[b * 256:(b + 1) * 256].contiguous(), [grids[b]], 6)
        assert torch.equal(single[0], batch[b]), (b, single[0].tolist(), batch[b].tolist())


# ---------------------------------------------------------------- LLaVA at CLIP-L/336 + Llama-7B widths
def test_llava_clip_slice_full_width(gpu):
    """2 encoder layers of the real CLIP ViT-L/14-336 geometry (1024 x 16 heads x 4096, 577 tokens per view, 4 views
    so the 256x256 GEMM and the head_dim-64 attention run at production shapes) + the 1024 -> 4096 projector, against
    the bf16 numpy oracle."""
    from lmms_owc_amd.engine.llava import LlavaDims, LlavaEngine, LlavaWeights
    from oracle import llava_np as L

    cfg = L.LlavaCfg(vision=L.ClipCfg(num_hidden_layers=3), text=Q.TextCfg(hidden_size=4096, num_hidden_layers=1, num_attention_heads=32,
                                                                           num_key_value_heads=32, intermediate_size=1024, vocab_size=1024,
                                                                           rms_norm_eps=1e-5, rope_theta=10000.0, tie_word_embeddings=False),
                     image_token_id=1000)
    w = recipes.llava_weights(cfg, 777)
    dims = LlavaDims(v_layers=3, n_layers=1, d_ff=1024, vocab=1024, image_token_id=1000, max_positions=1024)
    eng = LlavaEngine(LlavaWeights.from_state_dict(dims, w, gpu))
    pix = recipes.clip_pixels(4, 336, seed=3)
    p = pix.reshape(4, 3, 24, 14, 24, 14).transpose(0, 2, 4, 1, 3, 5).reshape(4 * 576, 588)
    p = np.concatenate([p, np.zeros((p.shape[0], dims.patch_k - 588), np.float32)], 1)
    out = to_np(eng.encode_views(torch.from_numpy(p).to(torch.bfloat16).to(gpu))).reshape(4, 577, 4096)[:, 1:]
    for i in (0, 3):  # views are independent; the oracle on two of them (numpy time)
        ref = L.project(w, L.clip_features(w, cfg, pix[i:i + 1], bf16=True), bf16=True)[0]
        assert np.abs(out[i] - ref).max() <= 0.02 * np.abs(ref).max(), (i, np.abs(out[i] - ref).max(), np.abs(ref).max())
        assert np.abs(out[i] - ref).mean() <= 0.004 * np.abs(ref).max()


def test_full_llava_15_7b_properties(gpu):
    """Full LLaVA-1.5-7B (random weights): batch invariance + determinism of greedy tokens, finite features."""
    from lmms_owc_amd.engine.llava import DIMS, LlavaEngine, LlavaWeights

    d = DIMS["llava-1.5-7b"]
    eng = LlavaEngine(LlavaWeights.random(d, gpu, seed=5))
    g = torch.Generator(device=gpu).manual_seed(1)
    n = 5
    u8 = torch.randint(0, 256, (n, 3, 336, 336), generator=g, device=gpu, dtype=torch.uint8)
    feats = eng.encode_views(eng.patchify(u8, (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)))
    assert feats.shape == (n * 577, 4096) and bool(torch.isfinite(feats.float()).all())
    rows = eng.feature_rows([1] * n)
    r = np.random.default_rng(2)
    head = r.integers(1000, 30000, 20)
    ids = [np.concatenate([head, np.full(576, d.image_token_id), r.integers(1000, 30000, 8 + i)]) for i in range(n)]
    a = to_np(eng.generate_from_features(ids, feats, rows, 6))
    b = to_np(eng.generate_from_features(ids, feats, rows, 6))
    assert np.array_equal(a, b)
    for i in (0, n - 1):
        s = to_np(eng.generate_from_features([ids[i]], feats, [rows[i]], 6))
        assert np.array_equal(s[0], a[i]), i


def test_full_7b_properties(gpu):
    """BASELINE config #3 at its own dimensions - full Qwen2-VL-7B (3584 / 28 q / 4 kv heads / 18944, 28 layers, random weights),
    with the bench's own chunking: 250 prompts span two prefill launch groups (65536 packed rows hold 240 x 272 + the shared
    prefix) and 136 images span two vision launch groups (131072 tokens = 128 images).  Properties: finite outputs,
    determinism, and batch invariance - a sequence's tokens inside the chunked batch equal the same sequence run alone
    (first / last of each launch group, i.e. across the group boundaries and the batch-size dependent decode kernels)."""
    from lmms_owc_amd.engine.qwen2vl import DIMS, Qwen2VLEngine, Qwen2VLWeights

    d = DIMS["qwen2-vl-7b"]
    eng = Qwen2VLEngine(Qwen2VLWeights.random(d, gpu, seed=5))
    g = torch.Generator(device=gpu).manual_seed(1)
    n_img, n = 136, 250
    pix = torch.randn((n_img * 1024, 1176), generator=g, device=gpu, dtype=torch.bfloat16)
    emb_img = eng.encode_images(pix, [(1, 32, 32)] * n_img)          # two vision launch groups (128 + 8 images)
    assert emb_img.shape == (n_img * 256, d.d_model) and bool(torch.isfinite(emb_img.float()).all())
    # image 130 (second vision group) alone == inside the chunked launch
    solo = eng.encode_images(pix[130 * 1024:131 * 1024], [(1, 32, 32)])
    assert torch.equal(solo, emb_img[130 * 256:131 * 256])
    r = np.random.default_rng(2)
    head = r.integers(1000, 150000, 14)
    ids = [np.concatenate([head, np.full(256, d.image_token_id), r.integers(1000, 150000, 16)]) for _ in range(n)]
    pick = [i % n_img for i in range(n)]
    rows = [256 * p + np.arange(256) for p in pick]
    grids = [[(1, 32, 32)]] * n
    assert eng._common_prefix(ids, 0, n) == 14
    batch, logits = eng.generate(ids, emb_img, grids, 6, img_rows=rows, return_logits=True)
    assert bool(torch.isfinite(logits.float()).all())
    again = eng.generate(ids, emb_img, grids, 6, img_rows=rows)
    assert torch.equal(batch, again)                                   # deterministic
    for b in (0, 120, 239, 240, 249):                                  # 240.. : the second prefill launch group
        single = eng.generate([ids[b]], emb_img, [grids[b]], 6, img_rows=[rows[b]])
        assert torch.equal(single[0], batch[b]), (b, single[0].tolist(), batch[b].tolist())
    # EOS-aware row compaction at full width (reference: one `generate` per image, each stopping at its own EOS,
    # src/models/_qwen2_vl.py:319-337): 700 rows (the 250 image prompts + 450 text prompts) with seeded ragged answer lengths
    # shrink through the 256x256, 128x128, ring and skinny GEMM regimes; no token of any row may change
    from tests.test_qwen2vl_gpu import _ragged_forced

    eos, T = 151645, 28
    ids2 = ids + [r.integers(1000, 150000, 20 + i % 11) for i in range(450)]
    rows2 = rows + [np.zeros(0, np.int64)] * 450
    grids2 = grids + [[]] * 450
    forced, lens = _ragged_forced(r, len(ids2), T, eos, 150000, 7)
    st = {}
    plain = eng.generate(ids2, emb_img, grids2, T, img_rows=rows2, eos_token_id=eos, forced_tokens=forced, compact_rows=False)
    comp = eng.generate(ids2, emb_img, grids2, T, img_rows=rows2, eos_token_id=eos, forced_tokens=forced, compact_rows=True, stats=st)
    assert torch.equal(plain, comp)
    live = st["live_rows_per_step"]
    assert live[0] == 700 and min(live) <= 64 and all(x >= y for x, y in zip(live, live[1:])), live


def test_full_7b_properties_ragged(gpu):
    """BASELINE config #3 on the datasets' REAL image sizes: the reference resizes every image inside [min_pixels, max_pixels]
    (/root/reference/src/models/_qwen2_vl.py:64-65, 299-305), so a Food-101 / DTD / Flowers-102 batch holds 64...1024 image
    tokens per image - ragged `cu_seqlens` in the vision tower, prompts of unequal length in the prefill (KV slots sized by the
    longest, the 14 shared text tokens prefilled once) and in every decode step's KV attention.  Full Qwen2-VL-7B, random
    weights; 200 images of mixed grids (4x4 patches ... 64x64, incl. non-square and the 200:1-ish thin ones) spanning two vision
    launch groups and two prefill launch groups.  Properties: finite outputs, determinism, and batch invariance of both the
    image embeddings and the greedy tokens (an image / a prompt alone == inside the ragged, chunked batch)."""
    from lmms_owc_amd.engine.qwen2vl import DIMS, Qwen2VLEngine, Qwen2VLWeights
    from lmms_owc_amd.models import imageproc

    d = DIMS["qwen2-vl-7b"]
    eng = Qwen2VLEngine(Qwen2VLWeights.random(d, gpu, seed=5), vit_chunk_tokens=65536, prefill_chunk_tokens=32768)
    r = np.random.default_rng(11)
    # (height, width) in pixels -> the processor's own resize rule -> patch grids
    sizes = [(512, 512)] * 40 + [(384, 512)] * 20 + [(int(h), int(w)) for h, w in r.integers(300, 641, (60, 2))] \
        + [(500, int(w)) for w in r.integers(500, 1001, 40)] + [(int(h), 500) for h in r.integers(500, 1001, 30)] \
        + [(56, 56), (28, 140), (2000, 1500), (896, 896), (100, 3000), (640, 480), (90, 70), (1024, 768), (30, 30), (448, 448)]
    order = r.permutation(len(sizes))
    grids = []
    for i in order:
        h, w = sizes[i]
        h1, w1 = imageproc.smart_resize(h, w, 28, 4 * 28 * 28, 16384 * 28 * 28)
        h2, w2 = imageproc.smart_resize(h1, w1, 28, 4 * 784, 1024 * 784)
        grids.append((1, h2 // 14, w2 // 14))
    n = len(grids)
    lens = [g[1] * g[2] for g in grids]
    n_tok = [t // 4 for t in lens]
    assert min(n_tok) <= 8 and max(n_tok) >= 1000 and len(set(grids)) > 60        # really ragged, both extremes present
    assert sum(lens) > 2 * 65536 // 2 and sum(30 + t for t in n_tok) > 32768     # > 1 vision group, > 1 prefill group
    g = torch.Generator(device=gpu).manual_seed(1)
    pix = torch.randn((sum(lens), 1176), generator=g, device=gpu, dtype=torch.bfloat16)
    emb = eng.encode_images(pix, grids)
    assert emb.shape == (sum(n_tok), d.d_model) and bool(torch.isfinite(emb.float()).all())
    starts = np.concatenate([[0], np.cumsum(lens)])
    estarts = np.concatenate([[0], np.cumsum(n_tok)])
    probe = sorted({0, n - 1, int(np.argmin(n_tok)), int(np.argmax(n_tok)), n // 2, n // 3})
    for i in probe:   # an image alone == inside the ragged launch groups
        solo = eng.encode_images(pix[starts[i]:starts[i + 1]], [grids[i]])
        assert torch.equal(solo, emb[estarts[i]:estarts[i + 1]]), i
    head = r.integers(1000, 150000, 14)
    ids = [np.concatenate([head, np.full(n_tok[i], d.image_token_id), r.integers(1000, 150000, 10 + i % 9)]) for i in range(n)]
    gpp = [[gr] for gr in grids]
    assert eng._common_prefix(ids, 0, n) == 14
    batch, logits = eng.generate(ids, emb, gpp, 5, return_logits=True)
    assert bool(torch.isfinite(logits.float()).all())
    assert torch.equal(batch, eng.generate(ids, emb, gpp, 5))            # deterministic
    for i in probe:   # a prompt alone (no shared-prefix segment, its own KV slot size) == inside the batch
        single = eng.generate([ids[i]], emb[estarts[i]:estarts[i + 1]], [gpp[i]], 5)
        assert torch.equal(single[0], batch[i]), (i, grids[i], single[0].tolist(), batch[i].tolist())
