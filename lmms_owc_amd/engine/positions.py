"""Integer position bookkeeping done on the host (numpy), mirroring what HF computes on the CPU side of
`generate` before the first forward:

* vision (h, w) patch coordinates in merge-block order — transformers/vision_utils.py
  get_vision_position_ids (called from HF modeling_qwen2_vl.py:710);
* 3-D M-RoPE position ids + rope delta — Qwen2VLModel.get_rope_index (HF modeling_qwen2_vl.py:914-1019).

Pure index arithmetic (bit-exact by construction), cached per grid shape because a dataset has few.
"""

from __future__ import annotations

from functools import lru_cache

import numpy as np


@lru_cache(maxsize=256)
def _vision_hw(t: int, h: int, w: int, merge: int) -> np.ndarray:
    hh = np.arange(h, dtype=np.int32)[:, None].repeat(w, 1)
    ww = np.arange(w, dtype=np.int32)[None, :].repeat(h, 0)

    def blockify(a):
        return a.reshape(h // merge, merge, w // merge, merge).swapaxes(1, 2).reshape(-1)

    hw = np.stack([blockify(hh), blockify(ww)], axis=1)
    return np.tile(hw, (t, 1))


def vision_pos_hw(grid_thw, merge: int = 2) -> np.ndarray:
    """[(sum t*h*w), 2] int32 (h, w) per patch row of pixel_values."""
    return np.concatenate([_vision_hw(int(t), int(h), int(w), merge) for t, h, w in grid_thw], axis=0)


@lru_cache(maxsize=256)
def _image_block(t: int, gh: int, gw: int) -> np.ndarray:
    tt = np.arange(t, dtype=np.int32).repeat(gh * gw)
    hh = np.tile(np.arange(gh, dtype=np.int32).repeat(gw), t)
    ww = np.tile(np.arange(gw, dtype=np.int32), t * gh)
    return np.stack([tt, hh, ww], axis=0)


def mrope_positions(ids: np.ndarray, grids, image_token_id: int, merge: int = 2):
    """Position ids [3, S] (int32) and rope delta for ONE un-padded prompt.

    Text runs advance all three streams together; an image run of t*gh*gw tokens gets
    (t, h, w) grid coordinates offset by the running position, which then advances by max(gh, gw).
    """
    ids = np.asarray(ids)
    is_img = ids == image_token_id
    pos = np.empty((3, ids.shape[0]), dtype=np.int32)
    edges = np.flatnonzero(np.diff(is_img.astype(np.int8))) + 1
    starts = np.concatenate([[0], edges])
    ends = np.concatenate([edges, [ids.shape[0]]])
    cur, gi = 0, 0
    for s0, s1 in zip(starts.tolist(), ends.tolist()):
        if not is_img[s0]:
            pos[:, s0:s1] = np.arange(cur, cur + (s1 - s0), dtype=np.int32)[None, :]
            cur += s1 - s0
            continue
        # one run of image tokens may hold several back-to-back images
        run = s0
        while run < s1:
            t, h, w = (int(x) for x in grids[gi])
            gi += 1
            gh, gw = h // merge, w // merge
            n = t * gh * gw
            pos[:, run:run + n] = _image_block(t, gh, gw) + cur
            cur += max(gh, gw)
            run += n
        if run != s1:
            raise ValueError("image token count does not match image_grid_thw")
    return pos, int(pos.max()) + 1 - ids.shape[0]
