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
def _window_perm(t: int, h: int, w: int, merge: int, ws: int):
    """One image of Qwen2.5-VL window attention (transformers/vision_utils.get_vision_window_index): the order in which the
    merged 2x2 token groups are visited so that the groups of one window (ws x ws groups) are contiguous, and the number of
    groups per window (border windows are ragged; the rule pads a full extra - empty - window when a side is a multiple of
    `ws`, those are dropped).  Returns (group order int64 [t*gh*gw], groups per non-empty window)."""
    gh, gw = h // merge, w // merge
    pad_h, pad_w = ws - gh % ws, ws - gw % ws
    nh, nw = (gh + pad_h) // ws, (gw + pad_w) // ws
    padded = np.full((t, gh + pad_h, gw + pad_w), -1, np.int64)
    padded[:, :gh, :gw] = np.arange(t * gh * gw).reshape(t, gh, gw)
    padded = padded.reshape(t, nh, ws, nw, ws).transpose(0, 1, 3, 2, 4).reshape(t * nh * nw, ws * ws)
    counts = (padded >= 0).sum(1)
    return padded[padded >= 0], counts[counts > 0]


def vision_windows(grid_thw, merge: int = 2, window_size: int = 112, patch_size: int = 14):
    """Window attention bookkeeping of the Qwen2.5-VL vision tower (HF modeling_qwen2_5_vl.py:423-445, :463-465), in PATCH rows:
    tok_index [T] (source row of every window-ordered row), out_index [T/merge^2] (window-ordered merged row of every output
    row = argsort(window_index)), win_start / win_len (patch rows per window, in window order)."""
    ws, unit = window_size // merge // patch_size, merge * merge
    order, lens, base = [], [], 0
    for t, h, w in grid_thw:
        o, c = _window_perm(int(t), int(h), int(w), merge, ws)
        order.append(o + base)
        lens.append(c * unit)
        base += int(t) * (int(h) // merge) * (int(w) // merge)
    window_index = np.concatenate(order)
    win_len = np.concatenate(lens).astype(np.int32)
    win_start = (np.cumsum(win_len) - win_len).astype(np.int32)
    tok_index = (window_index[:, None] * unit + np.arange(unit)[None, :]).reshape(-1).astype(np.int32)
    return tok_index, np.argsort(window_index).astype(np.int32), win_start, win_len


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
