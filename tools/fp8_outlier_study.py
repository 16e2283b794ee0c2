"""fp8 (e4m3fn, per-token x per-channel scales) under REALISTIC activation statistics - CPU, oracles only.

Random N(0, sigma) weights give every hidden dimension the same scale; real decoders carry a few "massive" channels 100-1000 x the
median.  For the 2-layer decoder slice at a model's widths (tests/test_decode_parity_gpu.py), with CHANNEL_OUTLIER_DIMS planted
channels scaled x 1 / 128 / 1024, this prints per case:
  * what the quantiser sees: max / median |x| of the projection inputs, the share of ordinary-channel values whose e4m3 code is
    0 or subnormal (the "crushed" codes), and the relative rounding error of the ordinary channels;
  * fp8 oracle vs bf16 oracle on the same weights, teacher-forced on the bf16 tokens: logit distance (max and mean, of max |logit|)
    and top-1 agreement over all (sequence, step) pairs.
usage: python tools/fp8_outlier_study.py [7b|72b] [n_sequences] [steps]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import fp8_np as F  # noqa: E402
from oracle import qwen2vl_np as Q  # noqa: E402
from tests import recipes  # noqa: E402
from tests import test_decode_parity_gpu as T  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "7b"
n_seq = int(sys.argv[2]) if len(sys.argv) > 2 else 6
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
grid = [(1, 4, 4)]
pix = recipes.pixel_values(grid, 50)
print(f"# {name} widths {T.WIDTHS[name]}, {n_seq} sequences x {steps} steps, {T.CHANNEL_OUTLIER_DIMS} planted channels")
for scale in (0.0, 128.0, 1024.0, 16384.0):
    t0 = time.time()
    cfg, w = T._slice_weights(name, channel_scale=scale)
    fp8 = F.quantize_decoder(w, Q.T, cfg.text.num_hidden_layers)
    prompts, _, _ = T._slice_case(cfg, n_seq, n_seq, seed=5)
    idx = T.channel_outlier_dims(cfg.text.hidden_size)
    ordinary = np.ones(cfg.text.hidden_size, bool)
    ordinary[idx] = False
    agree = total = 0
    dist_max, dist_mean, ratio, crushed, relerr = [], [], [], [], []
    for b in range(n_seq):
        taps = {}
        tb, lb = Q.generate(w, cfg, prompts[b], pix, grid, steps, bf16=True, return_logits=True)
        t8, l8 = Q.generate(w, cfg, prompts[b], pix, grid, steps, bf16=True, return_logits=True, fp8=fp8, forced_tokens=tb)
        lb, l8 = np.asarray(lb, np.float32), np.asarray(l8, np.float32)
        for j in range(steps):
            sc = np.abs(lb[j]).max()
            dist_max.append(np.abs(l8[j] - lb[j]).max() / sc)
            dist_mean.append(np.abs(l8[j] - lb[j]).mean() / sc)
            agree += int(np.argmax(l8[j]) == np.argmax(lb[j]))
            total += 1
        # what the quantiser sees at the qkv input of layer 0: rmsnorm(embedding rows) of this prompt's text tokens
        emb = w[Q.T + "embed_tokens.weight"][prompts[b][:10]].astype(np.float32)
        x = emb / np.sqrt((emb ** 2).mean(-1, keepdims=True) + 1e-6) * w[Q.T + "layers.0.input_layernorm.weight"].astype(np.float32)
        q, s = F.quantize_rows(x)
        ratio.append(float(np.median(np.abs(x).max(-1) / np.median(np.abs(x), -1))))
        mag = q[:, ordinary] & 0x7F
        crushed.append(float((mag < 8).mean()))            # codes 0..7: zero and the subnormals (< 2^-6 in code units)
        back = F.e4m3_decode(q).astype(np.float32) * s[:, None]
        xo = x[:, ordinary]
        relerr.append(float(np.abs(back[:, ordinary] - xo).mean() / np.abs(xo).mean()))
    print(f"channels x {scale:6.0f}: quantiser input max/median {np.mean(ratio):8.1f}, ordinary codes zero-or-subnormal {np.mean(crushed):6.1%}, "
          f"ordinary-channel rounding error {np.mean(relerr):6.2%} | fp8 vs bf16 logits: max {np.max(dist_max):.3f} mean-of-max {np.mean(dist_max):.3f} "
          f"mean {np.mean(dist_mean):.4f} of max|logit|, top-1 agreement {agree}/{total}   ({time.time() - t0:.0f} s)", flush=True)
