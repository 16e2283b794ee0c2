"""Accuracy of owc_attention_bf16 against a float64 softmax attention of the SAME bf16 inputs (no rounding anywhere in the reference):
what the kernel's own rounding points cost - P rounded to bf16 before P.V, and whatever a build does to Q on the way in.
Shapes: vision (head_dim 80, non-causal, 1024 and 4096 keys) and decoder prefill (head_dim 128, causal GQA); score statistics
from flat (logit std 1) to peaked (std 12).  Prints max |err| / max |O| and rms(err) / rms(O) per case.
  python tools/attn_accuracy.py          (A/B of two builds: tools/ab_libs.sh 1 python tools/attn_accuracy.py)"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lmms_owc_amd import ops  # noqa: E402


def i32(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)


def ref_attn(q, k, v, scale, causal):
    """q [H, L, d], k / v [Hk, L, d] float64 -> [H, L, d]"""
    H, L, _ = q.shape
    G = H // k.shape[0]
    out = np.empty_like(q)
    for h in range(H):
        s = (q[h] @ k[h // G].T) * scale
        if causal:
            s = np.where(np.tril(np.ones((L, L), bool)), s, -np.inf)
        s -= s.max(1, keepdims=True)
        p = np.exp(s)
        out[h] = (p @ v[h // G]) / p.sum(1, keepdims=True)
    return out


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    for name, H, Hk, L, hd, causal in (("vision hd80 L=1024", 4, 4, 1024, 80, False), ("vision hd80 L=4096", 2, 2, 4096, 80, False),
                                       ("vision hd80 L=3996", 2, 2, 3996, 80, False), ("prefill hd128 causal L=286", 4, 2, 286, 128, True),
                                       ("prefill hd128 causal L=2388", 4, 2, 2388, 128, True)):
        for logit_std in (1.0, 3.0, 12.0):
            # q.k * scale has std logit_std: q, k ~ N(0, a), sum of hd products has std a^2 sqrt(hd), scale = hd^-0.5 -> std a^2
            a = logit_std ** 0.5
            q = (torch.randn(L, H, hd, generator=g) * a).to(torch.bfloat16)
            k = (torch.randn(L, Hk, hd, generator=g) * a).to(torch.bfloat16)
            v = torch.randn(L, Hk, hd, generator=g).to(torch.bfloat16)
            out = torch.empty(L, H * hd, device=dev, dtype=torch.bfloat16)
            qd, kd, vd = q.reshape(L, H * hd).to(dev), k.reshape(L, Hk * hd).to(dev), v.reshape(L, Hk * hd).to(dev)
            st, ln = i32([0], dev), i32([L], dev)
            ops.attention(qd, H * hd, hd, kd, Hk * hd, hd, vd, Hk * hd, hd, out, H * hd, hd, st, st, ln, n_seq=1, n_heads=H,
                          kv_group=H // Hk, head_dim=hd, max_q_len=L, causal=causal, scale=hd ** -0.5)
            got = out.float().cpu().numpy().reshape(L, H, hd).transpose(1, 0, 2).astype(np.float64)
            ref = ref_attn(q.float().numpy().transpose(1, 0, 2).astype(np.float64), k.float().numpy().transpose(1, 0, 2).astype(np.float64),
                           v.float().numpy().transpose(1, 0, 2).astype(np.float64), hd ** -0.5, causal)
            ref_b = torch.from_numpy(ref).to(torch.bfloat16).double().numpy()          # the best any bf16 output can do
            err, floor = got - ref, ref_b - ref
            print(f"{name:30s} logit std {logit_std:4.1f}: max err {np.abs(err).max() / np.abs(ref).max():.5f} of max|O|, "
                  f"rms err / rms O {np.sqrt((err ** 2).mean() / (ref ** 2).mean()):.5f}   (output rounding alone: "
                  f"{np.abs(floor).max() / np.abs(ref).max():.5f}, {np.sqrt((floor ** 2).mean() / (ref ** 2).mean()):.5f})", flush=True)


if __name__ == "__main__":
    main()
