"""Op-level parity: each HIP kernel (through the C ABI) vs its numpy restatement in oracle/."""
import math

import numpy as np
import pytest
import torch

from oracle import np_ops
from oracle import qwen2vl_np as Q
from tests.util import assert_bf16_close, bf16_randn, to_np

pytestmark = pytest.mark.gpu
I32 = torch.int32


def i32(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)


@pytest.mark.parametrize("rows,d", [(5, 160), (300, 1280), (33, 3584), (4, 8192)])
def test_layernorm_rmsnorm(gpu, rows, d):
    from lmms_owc_amd import ops

    x = bf16_randn((rows, d), 1, 2.0, gpu)
    g = torch.Generator().manual_seed(rows * 7919 + d)
    w = (1 + 0.1 * torch.randn(d, generator=g)).to(torch.bfloat16).to(gpu)
    b = (0.1 * torch.randn(d, generator=g)).to(torch.bfloat16).to(gpu)
    y = ops.layernorm(x, w, b, 1e-6)
    assert_bf16_close(to_np(y), np_ops.layer_norm(to_np(x), to_np(w), to_np(b), 1e-6, bf16=True), atol=1e-3)
    # RMSNorm rounds twice (x*rstd -> bf16, then * weight -> bf16): a flipped first rounding is scaled by the
    # weight before the second one, so up to two bf16 spacings of slack
    y = ops.rmsnorm(x, w, 1e-6)
    assert_bf16_close(to_np(y), np_ops.rms_norm(to_np(x), to_np(w), 1e-6, bf16=True), ulps=4.5, atol=1e-3)
    idx = i32(np.arange(rows - 1, -1, -2), gpu)
    y = ops.rmsnorm(x, w, 1e-6, row_index=idx)
    assert_bf16_close(to_np(y), np_ops.rms_norm(to_np(x)[to_np(idx).astype(int)], to_np(w), 1e-6, bf16=True), ulps=4.5, atol=1e-3)


def test_vision_rope(gpu):
    from lmms_owc_amd import ops
    from lmms_owc_amd.engine import positions

    H, hd = 2, 80
    grid = [(1, 6, 4), (1, 4, 8)]
    T = sum(t * h * w for t, h, w in grid)
    qkv = bf16_randn((T, 3 * H * hd), 3, 1.0, gpu)
    ref = to_np(qkv).reshape(T, 3, H, hd)
    pos = positions.vision_pos_hw(grid)
    cos, sin = ops.rope_table(64, hd // 4, hd // 2, 10000.0, False, gpu)
    ops.vision_rope_(qkv, i32(pos, gpu), cos, sin, H, hd)
    inv = (1.0 / (10000.0 ** (np.arange(0, 40, 2, dtype=np.float32) / np.float32(40)))).astype(np.float32)
    fr = (pos[:, :, None].astype(np.float32) * inv).reshape(T, -1)
    emb = np.concatenate([fr, fr], -1)
    c, s = np.cos(emb)[:, None, :], np.sin(emb)[:, None, :]
    want = ref.copy()
    for j in (0, 1):
        want[:, j] = np_ops.bf16_round(ref[:, j] * c + Q._rotate_half(ref[:, j]) * s)
    assert_bf16_close(to_np(qkv).reshape(T, 3, H, hd), want, atol=1e-3, min_exact=0.99)


def test_mrope_kv_write(gpu):
    from lmms_owc_amd import ops

    Hq, Hkv, hd, T, s_max, slots = 4, 2, 128, 37, 50, 3
    qkv = bf16_randn((T, (Hq + 2 * Hkv) * hd), 5, 1.0, gpu)
    ref = to_np(qkv)
    r = np.random.default_rng(0)
    pos3 = r.integers(0, 300, (3, T))
    slot = r.integers(0, slots, T)
    # unique (slot, idx) pairs
    idx = np.array([np.sum(slot[:i] == slot[i]) for i in range(T)])
    kc = torch.zeros(slots * Hkv * s_max * hd, dtype=torch.bfloat16, device=gpu)
    vc = torch.zeros_like(kc)
    cos, sin = ops.rope_table(512, 64, 128, 1e6, True, gpu)
    ops.mrope_kv_write_(qkv, i32(pos3, gpu), cos, sin, kc, vc, i32(slot, gpu), i32(idx, gpu), Hq, Hkv, s_max, 16, 24)
    tc = Q.TextCfg()
    c, s = Q._mrope_cos_sin(pos3, tc, hd, True)
    x = ref.reshape(T, Hq + 2 * Hkv, hd)
    rot = lambda a: np_ops.bf16_round(np_ops.bf16_round(a * c[:, None]) + np_ops.bf16_round(Q._rotate_half(a) * s[:, None]))  # noqa: E731
    q_want, k_want, v_want = rot(x[:, :Hq]), rot(x[:, Hq:Hq + Hkv]), x[:, Hq + Hkv:]
    got = to_np(qkv).reshape(T, Hq + 2 * Hkv, hd)
    assert_bf16_close(got[:, :Hq], q_want, atol=1e-3, min_exact=0.98)
    kcn = to_np(kc).reshape(slots, Hkv, s_max, hd)
    vcn = to_np(vc).reshape(slots, Hkv, s_max, hd)
    assert_bf16_close(kcn[slot, :, idx], k_want, atol=1e-3, min_exact=0.98)
    assert np.array_equal(vcn[slot, :, idx], v_want)


def _attn_ref(q, k, v, causal):
    """[H, Sq, hd] x [H, Sk, hd] -> [H, Sq, hd], fp32 softmax, bf16-rounded output."""
    return Q._attn(q, k, v, q.shape[-1] ** -0.5, causal, False)


@pytest.mark.parametrize("lens", [[16], [1024], [24, 32, 700]])
def test_attention_vision_varlen(gpu, lens):
    from lmms_owc_amd import ops

    H, hd = 4, 80
    T = sum(lens)
    qkv = bf16_randn((T, 3 * H * hd), 9, 1.0, gpu)
    out = torch.zeros((T, H * hd), dtype=torch.bfloat16, device=gpu)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    E = H * hd
    ops.attention(qkv, 3 * E, hd, qkv[:, E:], 3 * E, hd, qkv[:, 2 * E:], 3 * E, hd, out, E, hd, i32(starts, gpu),
                  i32(starts, gpu), i32(lens, gpu), n_seq=len(lens), n_heads=H, kv_group=1, head_dim=hd,
                  max_q_len=max(lens), causal=False, scale=hd ** -0.5)
    x = to_np(qkv).reshape(T, 3, H, hd)
    want = np.empty((T, H, hd), np.float32)
    for s0, n in zip(starts, lens):
        sl = slice(s0, s0 + n)
        want[sl] = _attn_ref(x[sl, 0].transpose(1, 0, 2), x[sl, 1].transpose(1, 0, 2), x[sl, 2].transpose(1, 0, 2), False).transpose(1, 0, 2)
    got = to_np(out).reshape(T, H, hd)
    # P is rounded to bf16 before P.V (as HF eager/flash do): allow ~2 bf16 ulps of the value scale
    assert np.abs(got - want).max() <= 0.02 * np.abs(want).max() + 1e-3
    assert np.abs(got - want).mean() <= 2e-3 * np.abs(want).max()


@pytest.mark.parametrize("lens", [[7], [286], [130, 64, 257]])
def test_attention_causal_gqa_cache(gpu, lens):
    from lmms_owc_amd import ops

    Hq, Hkv, hd, s_max = 6, 2, 128, 300
    n = len(lens)
    T = sum(lens)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    q = bf16_randn((T, Hq * hd), 1, 1.0, gpu)
    kc = bf16_randn((n, Hkv, s_max, hd), 2, 1.0, gpu)
    vc = bf16_randn((n, Hkv, s_max, hd), 3, 1.0, gpu)
    out = torch.zeros((T, Hq * hd), dtype=torch.bfloat16, device=gpu)
    k_start = np.arange(n) * Hkv * s_max
    ops.attention(q, Hq * hd, hd, kc, hd, s_max * hd, vc, hd, s_max * hd, out, Hq * hd, hd, i32(starts, gpu),
                  i32(k_start, gpu), i32(lens, gpu), n_seq=n, n_heads=Hq, kv_group=Hq // Hkv, head_dim=hd,
                  max_q_len=max(lens), causal=True, scale=hd ** -0.5)
    qn, kn, vn = to_np(q).reshape(T, Hq, hd), to_np(kc), to_np(vc)
    got = to_np(out).reshape(T, Hq, hd)
    for b, (s0, L) in enumerate(zip(starts, lens)):
        kk = np.repeat(kn[b, :, :L], Hq // Hkv, 0)
        vv = np.repeat(vn[b, :, :L], Hq // Hkv, 0)
        want = _attn_ref(qn[s0:s0 + L].transpose(1, 0, 2), kk, vv, True).transpose(1, 0, 2)
        assert np.abs(got[s0:s0 + L] - want).max() <= 0.02 * np.abs(want).max() + 1e-3


@pytest.mark.parametrize("Hq,Hkv,lens,prefix", [(28, 4, [286, 1, 31, 300, 64, 129, 97], 0), (28, 4, [286] * 9, 0), (12, 2, [130, 64, 257, 5], 0),
                                                 (16, 8, [200, 333], 0), (28, 4, [271, 40, 150], 15), (6, 6, [286, 100], 0)])
def test_attention_causal_gqa_packing_is_bit_identical(gpu, Hq, Hkv, lens, prefix):
    """Round 5: causal launches with kv_group > 1 pack the (position, head) rows of a kv group into the blocks - position-major, so a
    128-row block holds ~18 positions of all 7 heads and walks the key tiles of 18 positions, not of 128.  Against the one-head-per-block
    mapping of rounds 1-4 (knob off): the SAME BITS (a row's key tiles, their order and its arithmetic do not change), for ragged
    lengths (1 ... 333: several blocks, partial waves), group sizes 7 / 6 / 2, a shared prefix in front of the rows (q_len < seq_len:
    the causal offset) and G = 1 (nothing to pack); and within 2 % of the fp32 reference."""
    from lmms_owc_amd import _lib, ops

    lib = _lib.load()
    hd, s_max = 128, 352
    n = len(lens)
    T = sum(lens)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    q = bf16_randn((T, Hq * hd), 11, 1.0, gpu)
    kc = bf16_randn((n, Hkv, s_max, hd), 12, 1.0, gpu)
    vc = bf16_randn((n, Hkv, s_max, hd), 13, 1.0, gpu)
    k_start = np.arange(n) * Hkv * s_max
    klen = np.asarray(lens) + prefix      # keys = the shared prefix's + the rows' own

    def run():
        out = torch.zeros((T, Hq * hd), dtype=torch.bfloat16, device=gpu)
        ops.attention(q, Hq * hd, hd, kc, hd, s_max * hd, vc, hd, s_max * hd, out, Hq * hd, hd, i32(starts, gpu), i32(k_start, gpu),
                      i32(klen, gpu), n_seq=n, n_heads=Hq, kv_group=Hq // Hkv, head_dim=hd, max_q_len=max(lens), causal=True,
                      scale=hd ** -0.5, q_len=i32(lens, gpu) if prefix else None)
        return out

    try:
        assert lib.owc_tuning_set(b"attn_gqa_pack", 0) == 0
        want = run()
        assert lib.owc_tuning_set(b"attn_gqa_pack", 1) == 0
        for _ in range(3):
            got = run()
            assert torch.equal(got, want), (got != want).sum().item()
    finally:
        lib.owc_tuning_set(b"attn_gqa_pack", 1)
    qn, kn, vn = to_np(q).reshape(T, Hq, hd), to_np(kc), to_np(vc)
    g = to_np(got).reshape(T, Hq, hd)
    G = Hq // Hkv
    for b in (0, n - 1):
        s0, L, Lk = starts[b], lens[b], klen[b]
        kk, vv = np.repeat(kn[b, :, :Lk], G, 0), np.repeat(vn[b, :, :Lk], G, 0)
        sc = np.einsum("qhd,hkd->hqk", qn[s0:s0 + L].astype(np.float64), kk.astype(np.float64)) * hd ** -0.5
        mask = np.arange(Lk)[None, :] > (np.arange(L)[:, None] + prefix)
        sc[:, mask] = -np.inf
        pr = np.exp(sc - sc.max(-1, keepdims=True))
        pr /= pr.sum(-1, keepdims=True)
        ref = np.einsum("hqk,hkd->qhd", pr, vv.astype(np.float64))
        assert np.abs(g[s0:s0 + L] - ref).max() <= 0.02 * np.abs(ref).max() + 1e-3


def test_attention_decode_mapping(gpu):
    """q_len = G query heads per kv group mapped onto kernel rows (what owc_llm_decode_step does)."""
    from lmms_owc_amd import ops

    Hq, Hkv, hd, s_max, B = 12, 2, 128, 96, 5
    G = Hq // Hkv
    klen = np.array([1, 17, 64, 65, 96])
    qkv = bf16_randn((B, (Hq + 2 * Hkv) * hd), 4, 1.0, gpu)
    kc = bf16_randn((B, Hkv, s_max, hd), 5, 1.0, gpu)
    vc = bf16_randn((B, Hkv, s_max, hd), 6, 1.0, gpu)
    out = torch.zeros((B, Hq * hd), dtype=torch.bfloat16, device=gpu)
    ar = np.arange(B)
    ops.attention(qkv, hd, G * hd, kc, hd, s_max * hd, vc, hd, s_max * hd, out, hd, G * hd, i32(ar * (Hq + 2 * Hkv), gpu),
                  i32(ar * Hkv * s_max, gpu), i32(klen, gpu), n_seq=B, n_heads=Hkv, kv_group=1, head_dim=hd, max_q_len=G,
                  causal=False, scale=hd ** -0.5, o_start=i32(ar * Hq, gpu), q_len=i32(np.full(B, G), gpu))
    qn = to_np(qkv)[:, :Hq * hd].reshape(B, Hq, hd)
    got = to_np(out).reshape(B, Hq, hd)
    for b in range(B):
        kk = np.repeat(to_np(kc)[b, :, :klen[b]], G, 0)
        vv = np.repeat(to_np(vc)[b, :, :klen[b]], G, 0)
        want = _attn_ref(qn[b][:, None, :], kk, vv, False)[:, 0]
        assert np.abs(got[b] - want).max() <= 0.02 * np.abs(want).max() + 1e-3


@pytest.mark.parametrize("Hq,Hkv", [(12, 2), (28, 4), (64, 8), (8, 8)])
def test_decode_attention_fused_equals_the_two_kernel_path(gpu, Hq, Hkv):
    """owc_decode_attention (rope + KV-cache write + attention of a decode step in one launch, the four waves of a block splitting
    the keys) against owc_mrope_kv_write + owc_attention_bf16 in the decode mapping: the cache rows it writes are the SAME BITS,
    nothing else in the caches is touched, the attention output agrees within the rounding of P and with the fp32 reference;
    key counts 1 ... 700 cross every tile / wave-ownership edge (1, 63, 64, 65, 128, 129, 255, 256, 257, 320)."""
    from lmms_owc_amd import ops

    hd, s_max = 128, 704
    G = Hq // Hkv
    klen = np.array([1, 2, 63, 64, 65, 128, 129, 255, 256, 257, 320, 500, 700])
    B = len(klen)
    r = np.random.default_rng(Hq)
    slot = r.permutation(B + 2)[:B]                  # slots are not the batch index
    pos = r.integers(0, 2000, B)
    qkv = bf16_randn((B, (Hq + 2 * Hkv) * hd), 4, 1.0, gpu)
    kc0 = bf16_randn((B + 2, Hkv, s_max, hd), 5, 1.0, gpu)
    vc0 = bf16_randn((B + 2, Hkv, s_max, hd), 6, 1.0, gpu)
    cos, sin = ops.rope_table(2048, 64, 128, 1e6, True, gpu)
    widx = klen - 1
    # reference path: mrope_kv (rotates q in place, writes the k / v rows) + the generic kernel in the decode mapping
    qkv_a, kc_a, vc_a = qkv.clone(), kc0.clone(), vc0.clone()
    ops.mrope_kv_write_(qkv_a, i32(pos, gpu), cos, sin, kc_a, vc_a, i32(slot, gpu), i32(widx, gpu), Hq, Hkv, s_max, 16, 24, pos_stride=0)
    out_a = torch.zeros((B, Hq * hd), dtype=torch.bfloat16, device=gpu)
    ar = np.arange(B)
    ops.attention(qkv_a, hd, G * hd, kc_a, hd, s_max * hd, vc_a, hd, s_max * hd, out_a, hd, G * hd, i32(ar * (Hq + 2 * Hkv), gpu),
                  i32(slot * Hkv * s_max, gpu), i32(klen, gpu), n_seq=B, n_heads=Hkv, kv_group=1, head_dim=hd, max_q_len=G,
                  causal=False, scale=hd ** -0.5, o_start=i32(ar * Hq, gpu), q_len=i32(np.full(B, G), gpu))
    # fused path
    kc_b, vc_b = kc0.clone(), vc0.clone()
    out_b = ops.decode_attention(qkv, i32(pos, gpu), cos, sin, kc_b, vc_b, i32(slot, gpu), i32(widx, gpu), i32(klen, gpu),
                                 Hq, Hkv, s_max, hd ** -0.5)
    assert torch.equal(kc_a, kc_b) and torch.equal(vc_a, vc_b)           # same rows, same bits, nothing else touched
    assert not torch.equal(kc_b, kc0)
    a, b = to_np(out_a), to_np(out_b)
    assert np.abs(a - b).max() <= 0.01 * np.abs(a).max() + 1e-3, np.abs(a - b).max()
    qn = to_np(qkv_a)[:, :Hq * hd].reshape(B, Hq, hd)                    # rotated q
    kn, vn = to_np(kc_b), to_np(vc_b)
    for i in range(B):
        kk = np.repeat(kn[slot[i], :, :klen[i]], G, 0)
        vv = np.repeat(vn[slot[i], :, :klen[i]], G, 0)
        want = _attn_ref(qn[i][:, None, :], kk, vv, False)[:, 0]
        got = b[i].reshape(Hq, hd)
        assert np.abs(got - want).max() <= 0.02 * np.abs(want).max() + 1e-3, (i, klen[i])
    # the large-batch form (one V buffer per wave, two blocks per CU) fetches the same tiles at other times: same bits
    from lmms_owc_amd import _lib

    lib = _lib.load()
    try:
        assert lib.owc_tuning_set(b"decode_attn_nbuf1", 0) == 0
        out_c = ops.decode_attention(qkv, i32(pos, gpu), cos, sin, kc0.clone(), vc0.clone(), i32(slot, gpu), i32(widx, gpu),
                                     i32(klen, gpu), Hq, Hkv, s_max, hd ** -0.5)
    finally:
        lib.owc_tuning_set(b"decode_attn_nbuf1", -1)
    assert torch.equal(out_c, out_b)
    # a sequence's result does not depend on its neighbours: the last sequence alone, bit for bit
    one = ops.decode_attention(qkv[B - 1:], i32(pos[B - 1:], gpu), cos, sin, kc0.clone(), vc0.clone(), i32(slot[B - 1:], gpu),
                               i32(widx[B - 1:], gpu), i32(klen[B - 1:], gpu), Hq, Hkv, s_max, hd ** -0.5)
    assert torch.equal(one[0], out_b[B - 1])


@pytest.mark.parametrize("qscale", [1.0, 0.05])
def test_decode_attention_forms_are_bit_identical_on_many_rows(gpu, qscale):
    """The small-batch form (two V buffers per wave) and the large-batch form (one) of attn_decode_fused_kernel on 1536 sequences of
    257 ... 1500 keys (every wave owns several tiles; near-flat scores at qscale 0.05, like a random-weight decoder): same bits.  A
    decode batch that shrinks (EOS-aware row compaction) crosses from one form to the other in the middle of a sequence; round 4
    found the two instantiations contracted one mul + add differently (a last-bit difference in ~1 % of the rows)."""
    from lmms_owc_amd import _lib, ops

    Hq, Hkv, hd, s_max = 28, 4, 128, 1504
    r = np.random.default_rng(17)
    B = 1536
    klen = r.integers(257, 1501, B)
    slot = r.permutation(B)
    pos = r.integers(0, 2000, B)
    g = torch.Generator(device=gpu).manual_seed(4)
    qkv = (torch.randn((B, (Hq + 2 * Hkv) * hd), generator=g, device=gpu) * qscale).to(torch.bfloat16)
    kc0 = torch.randn((B, Hkv, s_max, hd), generator=g, device=gpu).to(torch.bfloat16)
    vc0 = torch.randn((B, Hkv, s_max, hd), generator=g, device=gpu).to(torch.bfloat16)
    cos, sin = ops.rope_table(2048, 64, 128, 1e6, True, gpu)
    args = (i32(pos, gpu), cos, sin)
    tail = (i32(slot, gpu), i32(klen - 1, gpu), i32(klen, gpu), Hq, Hkv, s_max, hd ** -0.5)
    lib = _lib.load()
    outs = []
    try:
        for knob in (0, 1 << 30):            # 0: every launch takes the one-buffer form; huge: every launch the two-buffer form
            assert lib.owc_tuning_set(b"decode_attn_nbuf1", knob) == 0
            outs.append(ops.decode_attention(qkv, *args, kc0.clone(), vc0.clone(), *tail))
    finally:
        lib.owc_tuning_set(b"decode_attn_nbuf1", -1)
    diff = (outs[0] != outs[1]).any(dim=1).sum().item()
    assert diff == 0, f"{diff} of {B} rows differ between the two forms"


@pytest.mark.parametrize("T,k,p", [(0.8, 0, None), (1.0, 50, None), (1.0, 0, 0.9), (0.7, 40, 0.8), (1.5, 7, 0.6)])
def test_sampling_distribution_matches_the_hf_restatement(gpu, T, k, p):
    """owc_sample_bf16 (temperature -> top-k -> top-p -> one draw per row) against oracle/sampling_np.py (pinned on transformers'
    own warpers, tests/test_oracle_sampling.py): 60 000 rows of the SAME logits, one random stream each - the support is exactly the
    oracle's kept set and every token's count is within 5 sigma of its expectation (bit parity with torch's generator is not the
    claim, the distribution is)."""
    from lmms_owc_amd import ops
    from oracle import sampling_np as S

    V, R = 1003, 60000
    r = np.random.default_rng(int(T * 10) + k)
    row = torch.from_numpy((r.standard_normal(V) * 2.5).astype(np.float32)).to(torch.bfloat16)
    probs = S.sampling_probs(row.float().numpy(), T, k, p)
    logits = row.to(gpu)[None].expand(R, V).contiguous()
    got = to_np(ops.sample_bf16(logits, T, k, p, seed=1234)).astype(np.int64)
    counts = np.bincount(got, minlength=V)
    assert (counts[probs == 0] == 0).all(), "a token outside the kept set was drawn"
    exp = probs * R
    big = exp >= 20
    z = (counts[big] - exp[big]) / np.sqrt(exp[big] * (1 - probs[big]))
    assert np.abs(z).max() < 5.0, (np.abs(z).max(), int(big.sum()))
    assert abs(counts[~big].sum() - exp[~big].sum()) < 5.0 * np.sqrt(exp[~big].sum() + 1)
    # same seed -> same draws; another seed -> other draws; the stream follows the ORIGINAL row through a row permutation
    again = to_np(ops.sample_bf16(logits, T, k, p, seed=1234))
    other = to_np(ops.sample_bf16(logits, T, k, p, seed=99))
    assert np.array_equal(again, got) and (other != got).mean() > 0.3
    perm = r.permutation(R)[:5000].astype(np.int32)
    sub = to_np(ops.sample_bf16(logits[:5000], T, k, p, seed=1234, row_map=i32(perm, gpu)))
    assert np.array_equal(sub, got[perm])
    sid = r.integers(0, R, 5000).astype(np.int32)
    by_id = to_np(ops.sample_bf16(logits[:5000], T, k, p, seed=1234, stream_ids=i32(sid, gpu)))
    assert np.array_equal(by_id, got[sid])             # identical logits: the draw is a function of (seed, stream id, step) only
    step3 = to_np(ops.sample_bf16(logits[:5000], T, k, p, seed=1234, step=3))
    assert (step3 != got[:5000]).mean() > 0.3


def test_sampling_extremes_and_ties(gpu):
    """top_k = 1 (what Qwen2-VL's own generation_config.json sets) is the argmax whatever the temperature; a cold temperature is the
    argmax too; equal logits at the top-k cut all stay eligible (HF's `scores < kth value` rule); ragged vocabulary sizes."""
    from lmms_owc_amd import ops

    g = torch.Generator(device=gpu).manual_seed(8)
    for V in (152064, 2048, 517):
        logits = (torch.randn((64, V), generator=g, device=gpu) * 3).to(torch.bfloat16)
        vals = logits.float()
        top = vals.max(dim=1).values
        for kw in (dict(temperature=1.0, top_k=1), dict(temperature=1e-3), dict(temperature=0.7, top_k=1, top_p=0.5)):
            got = ops.sample_bf16(logits, seed=5, **kw)
            assert bool((vals.gather(1, got.long()[:, None])[:, 0] == top).all()), (V, kw)   # a maximal token (ties: any of them)
        assert torch.equal(ops.sample_bf16(logits, 1.0, top_k=1, seed=1), ops.sample_bf16(logits, 1.0, top_k=1, seed=1))
    tie = torch.tensor([[1.0, 3.0, 3.0, 2.0, 3.0, 0.0, -1.0, 0.5]], device=gpu).to(torch.bfloat16).expand(4000, 8).contiguous()
    drawn = set(to_np(ops.sample_bf16(tie, 1.0, top_k=2, seed=2)).tolist())
    assert drawn == {1, 2, 4}


def test_sampling_top_p_cut_inside_tied_logits(gpu):
    """Round 6: where the top-p cut falls inside a run of equal bf16 logits the kernel keeps HF's NUMBER of
    them (oracle/sampling_np.py, pinned on transformers' TopPLogitsWarper in tests/test_oracle_sampling.py) and takes the lowest ids;
    before, the whole run stayed eligible.  Support and distribution against the oracle; then the case the reference actually runs
    into: Qwen2-VL's generation_config.json (top_k 1, top_p 0.001) with a temperature > 0 must return `owc_argmax_bf16` bit for bit,
    tied maxima included."""
    from lmms_owc_amd import ops
    from oracle import sampling_np as S

    R = 40000
    r = np.random.default_rng(11)
    rows = [(np.array([2.0, 0.0, 2.0, 2.0, 2.0, 1.0, 2.0, -1.0], np.float32), 1.0, 0, 0.5),
            (np.array([3.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.0, 1.0, 1.0], np.float32), 1.0, 0, 0.8),
            (np.array([1.0, 3.0, 3.0, 2.0, 3.0, 0.0, -1.0, 0.5], np.float32), 1.0, 2, 0.6),
            (np.round(r.standard_normal(5003) * 2).astype(np.float32), 0.9, 40, 0.7),           # integer logits: the cut bin holds hundreds
            (np.round(r.standard_normal(152064) * 1.5).astype(np.float32) * 0.5, 1.1, 0, 0.3)]  # runs spread over all 1024 threads' chunks
    for row, T, k, p in rows:
        V = len(row)
        probs = S.sampling_probs(row, T, k, p)
        n = R if V < 10000 else 4000
        logits = torch.from_numpy(row).to(torch.bfloat16).to(gpu)[None].expand(n, V).contiguous()
        got = to_np(ops.sample_bf16(logits, T, k, p, seed=77)).astype(np.int64)
        counts = np.bincount(got, minlength=V)
        assert (counts[probs == 0] == 0).all(), f"a token outside the kept set was drawn (V={V}, k={k}, p={p}): {np.flatnonzero((counts > 0) & (probs == 0))[:8]}"
        exp = probs * n
        big = exp >= 20
        if big.any():
            z = (counts[big] - exp[big]) / np.sqrt(exp[big] * (1 - probs[big]) + 1e-12)
            assert np.abs(z).max() < 5.0, (V, k, p, np.abs(z).max())
        # tokens too rare to test one by one: by VALUE (equal logits = equal probability: the kept part of a run shares its mass evenly)
        for v in np.unique(row[probs > 0]):
            idx = np.flatnonzero((row == v) & (probs > 0))
            e = exp[idx].sum()
            if e >= 20:
                assert abs(counts[idx].sum() - e) < 5.0 * np.sqrt(e), (V, k, p, float(v), int(counts[idx].sum()), float(e))
    # top_k 1 + top_p 0.001 at any temperature == the greedy argmax, ties included (lowest index)
    g = torch.Generator(device=gpu).manual_seed(3)
    for V in (152064, 520):   # (owc_argmax_bf16 reads 16-byte vectors: row stride a multiple of 8)
        logits = torch.round(torch.randn((256, V), generator=g, device=gpu) * 2).to(torch.bfloat16)   # integer values: many tied maxima
        assert (logits.float() == logits.float().max(dim=1, keepdim=True).values).sum(dim=1).max().item() > 1
        want = ops.argmax_bf16(logits)
        for T in (0.01, 1.0, 3.0):
            got = ops.sample_bf16(logits, T, 1, 0.001, seed=5)
            assert torch.equal(got.to(want.dtype), want), (V, T)


def test_embed_argmax_patchify(gpu):
    from lmms_owc_amd import ops

    table = bf16_randn((50, 64), 1, 1.0, gpu)
    img = bf16_randn((6, 64), 2, 1.0, gpu)
    ids = np.array([3, 7, 49, 49, 49, 0, 49], dtype=np.int32)
    iidx = np.array([-1, -1, 0, 1, 2, -1, 5], dtype=np.int32)
    out = ops.embed_tokens(i32(ids, gpu), i32(iidx, gpu), table, img)
    want = np.where(iidx[:, None] >= 0, to_np(img)[np.maximum(iidx, 0)], to_np(table)[ids])
    assert np.array_equal(to_np(out), want)

    logits = bf16_randn((9, 1000), 3, 1.0, gpu)
    logits[2, 10] = 50.0
    logits[2, 700] = 50.0  # tie -> lowest index
    got = ops.argmax_bf16(logits)
    assert np.array_equal(to_np(got).astype(int), np.argmax(to_np(logits), -1))
    # the real vocabulary (152 064 = 19 008 chunks of 8: the 4-way unrolled part, the remainder loop) and a ragged one with a
    # scalar tail; ties planted across thread / unroll / wave boundaries must resolve to the LOWEST index
    for V in (152064, 152064 - 3, 8 * 1024 * 4 + 8):
        logits = bf16_randn((5, 152064), 4, 1.0, gpu)[:, :V]   # row stride stays a multiple of 8 elements (16-byte loads)
        logits[1, [7, 8 * 1024 + 7, V - 1]] = 60.0
        logits[2, [V - 1, V - 2]] = 60.0
        logits[3, [8 * 4096 - 1, 8 * 4096]] = 60.0
        logits[4, 0] = 60.0
        got = to_np(ops.argmax_bf16(logits)).astype(int)
        assert np.array_equal(got, np.argmax(to_np(logits), -1)), (V, got)

    g = torch.Generator().manual_seed(0)
    im = torch.randint(0, 256, (2, 3, 56, 84), generator=g, dtype=torch.uint8)
    mean, std = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)
    pv = to_np(ops.patchify_u8(im.to(gpu), mean, std))
    x = (im.numpy().astype(np.float32) / 255.0 - np.array(mean, np.float32)[None, :, None, None]) / np.array(std, np.float32)[None, :, None, None]
    gh, gw = 4, 6
    p = x.reshape(2, 3, gh // 2, 2, 14, gw // 2, 2, 14).transpose(0, 2, 5, 3, 6, 1, 4, 7)  # n, bh, bw, ih, iw, c, py, px
    p = np.repeat(p[:, :, :, :, :, :, None], 2, axis=6).reshape(2 * gh * gw, 1176)
    assert_bf16_close(pv, np_ops.bf16_round(p), ulps=1.0, atol=1e-6, min_exact=0.99)


@pytest.mark.parametrize("m,n,k,epi", [(100, 384, 384, "none"), (257, 1536, 384, "gelu"), (64, 384, 1536, "res"), (5, 64, 132, "none")])
def test_gemm_f32(gpu, m, n, k, epi):
    from lmms_owc_amd import _lib, ops

    g = torch.Generator().manual_seed(1)
    a = torch.randn(m, k, generator=g).to(gpu)
    w = (torch.randn(n, k, generator=g) / math.sqrt(k)).to(gpu)
    b = torch.randn(n, generator=g).to(gpu)
    y = to_np(a).astype(np.float64) @ to_np(w).astype(np.float64).T + to_np(b)
    if epi == "gelu":
        out = ops.gemm_f32(a, w, b, epilogue=_lib.EPI_GELU_ERF)
        want = np_ops.gelu_erf(y.astype(np.float32))
    elif epi == "res":
        r = torch.randn(m, n, generator=g).to(gpu)
        out = ops.gemm_f32(a, w, b, epilogue=_lib.EPI_RESIDUAL, residual=r)
        want = y + to_np(r)
    else:
        out = ops.gemm_f32(a, w, b)
        want = y
    np.testing.assert_allclose(to_np(out), want, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("V,k", [(512, 4), (1000, 7), (151936, 8), (32064, 64)])
def test_beam_candidates_vs_numpy(gpu, V, k):
    """`owc_beam_candidates` (the device half of beam search): per row the log-sum-exp of the bf16 logits and the k largest
    (value, id), ties by the lowest id (bf16 logits over a 150 k vocabulary tie often) - against numpy on the same bf16 values."""
    from lmms_owc_amd import ops

    rng = np.random.default_rng(V + k)
    rows = 5
    x = torch.from_numpy((rng.normal(size=(rows, V)) * 3).astype(np.float32)).to(torch.bfloat16)
    x[1, : V // 2] = x[1, V // 2: V // 2 * 2]          # planted ties, also at the top
    x[2, 7] = x[2, V - 3] = 40.0
    x[3] = 1.5                                          # a constant row: ids 0 .. k-1
    logz, tv, ti = (to_np(t) for t in ops.beam_candidates(x.to(gpu), k))
    f = x.float().numpy()
    order = np.lexsort((np.broadcast_to(np.arange(V), f.shape), -f.astype(np.float64)), axis=-1)[:, :k]
    assert np.array_equal(ti, order)
    assert np.array_equal(tv, np.take_along_axis(f, order, -1))
    m = f.max(-1).astype(np.float64)
    want = m + np.log(np.exp(f.astype(np.float64) - m[:, None]).sum(-1))
    np.testing.assert_allclose(logz, want, rtol=0, atol=2e-5 * max(1.0, np.abs(want).max()))
    with pytest.raises(Exception):
        ops.beam_candidates(x.to(gpu), 65)


@pytest.mark.parametrize("V", [152064, 1000, 520])
def test_penalised_argmax_and_seen_bitmap(gpu, V):
    """`owc_seen_mark` + `owc_argmax_penalized_bf16` against oracle `repetition_penalty_scores` (pinned on HF's processor): rows that
    share bitmap rows through `row_slot`, duplicate ids, negative and positive maxima, ties (lowest index), ids at the word edges."""
    from lmms_owc_amd import ops

    r = np.random.default_rng(V)
    rows, slots = 48, 7
    logits = (torch.from_numpy(np.round(r.standard_normal((rows, V)) * 3, 1).astype(np.float32)) - (1.0 if V == 1000 else 0.0)).to(torch.bfloat16)
    if V == 1000:
        logits[:8] -= 20.0        # all-negative rows: the penalty multiplies
    row_slot = r.integers(0, slots, rows)
    top1 = np.argmax(logits.float().numpy(), 1)
    # a slot's history: random ids, ids at the bitmap's word edges, and the plain argmax of HALF of the rows that use the slot
    hist = [np.unique(np.concatenate([r.integers(0, V, 40), [0, 31, 32, V - 1], top1[(row_slot == s) & (np.arange(rows) % 2 == 0)]])) for s in range(slots)]
    seen = torch.zeros((slots, (V + 31) // 32), dtype=torch.int32, device=gpu)
    ids = np.concatenate([np.concatenate([h, h[:5]]) for h in hist])          # duplicates are marked once
    slot_of = np.concatenate([np.full(len(h) + 5, s) for s, h in enumerate(hist)])
    ops.seen_mark_(seen, i32(ids, gpu), i32(slot_of, gpu))
    bits = seen.cpu().numpy().view(np.uint32)
    for s in range(slots):
        marked = np.flatnonzero(np.unpackbits(np.ascontiguousarray(bits[s]).view(np.uint8), bitorder="little")[:V])
        assert np.array_equal(marked, hist[s])
    for penalty in (1.05, 1.5, 0.7):
        got = to_np(ops.argmax_penalized_bf16(logits.to(gpu), seen, penalty, i32(row_slot, gpu))).astype(int)
        x = logits.float().numpy()
        want = np.array([int(np.argmax(Q.repetition_penalty_scores(x[i], hist[row_slot[i]], penalty))) for i in range(rows)])
        assert np.array_equal(got, want), np.flatnonzero(got != want)
        if penalty > 1:
            assert (want != np.argmax(x, 1)).any()     # the penalty moved some argmax (the plain maximum of half the rows is in their history)
