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
