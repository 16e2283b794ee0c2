"""Helpers shared by the parity tests."""
import numpy as np
import torch


def to_np(t: torch.Tensor) -> np.ndarray:
    return t.detach().float().cpu().numpy()


def bf16_randn(shape, seed, scale=1.0, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(torch.bfloat16).to(device)


def assert_bf16_close(got: np.ndarray, want: np.ndarray, ulps=2.0, min_exact=0.90, atol=1e-30):
    """Both sides are bf16-rounded values; accumulation order may move a result by a bf16 ulp."""
    got = got.astype(np.float32)
    want = want.astype(np.float32)
    assert got.shape == want.shape
    tol = ulps * np.maximum(np.abs(want), np.abs(got)) * 2.0**-8 + atol
    bad = np.abs(got - want) > tol
    assert not bad.any(), f"{bad.sum()} / {bad.size} beyond {ulps} bf16 ulp; max abs diff {np.abs(got - want).max()}"
    exact = (got == want).mean()
    assert exact >= min_exact, f"only {exact:.4f} bit-exact"


def check_forced_steps(got_logits, got_toks, ref_logits, ref_toks, frac, tag, mean_frac=None, margin_frac=None) -> int:
    """Teacher-forced decode parity: EVERY step j of `got_logits` [T, V] is within `frac` * max |ref_logits[j]| of the
    reference's step-j logits (both conditional on the same forced continuation).  Token rule per step: where the
    reference's top-2 margin exceeds `margin_frac` (default 2 * frac) * max|ref| the engine's argmax must equal the reference's (`ref_toks[j]`,
    or argmax(ref_logits[j]) when None); on a near-tie the engine's token must still score within that margin of the
    reference's maximum.  Never stops at a near-tie.  Returns the number of steps whose token was asserted equal."""
    got_logits, ref_logits = np.asarray(got_logits, np.float32), np.asarray(ref_logits, np.float32)
    assert got_logits.shape == ref_logits.shape, (got_logits.shape, ref_logits.shape)
    decisive, worst = 0, 0.0
    margin_frac = 2 * frac if margin_frac is None else margin_frac
    for j in range(ref_logits.shape[0]):
        ref, got = ref_logits[j], got_logits[j]
        scale = np.abs(ref).max()
        err = np.abs(got - ref).max()
        worst = max(worst, float(err / scale))
        assert err <= frac * scale, f"{tag}: step {j} logits off by {err / scale:.4f} of max|logit| (bound {frac})"
        if mean_frac is not None:
            assert np.abs(got - ref).mean() <= mean_frac * scale, f"{tag}: step {j} mean logit error"
        want = int(ref_toks[j]) if ref_toks is not None else int(np.argmax(ref))
        top2 = np.sort(ref)[-2:]
        if top2[1] - top2[0] > margin_frac * scale:
            assert int(got_toks[j]) == want, f"{tag}: step {j} token {int(got_toks[j])} != {want}"
            decisive += 1
        else:
            assert ref[int(got_toks[j])] >= top2[1] - margin_frac * scale, f"{tag}: step {j} picked a non-candidate token"
    print(f"[forced-steps] {tag}: {ref_logits.shape[0]} steps, worst logit error {worst:.4f} of max|logit| (bound {frac}), "
          f"{decisive} decisive token steps")
    return decisive
