"""LLaVA-NeXT "anyres" integer bookkeeping (host side): which tiling an image gets and which projected CLIP
rows its <image> placeholders read.  Restates HF transformers
  image_processing_utils.select_best_resolution, modeling_llava_next.py get_anyres_image_grid_shape (:41-69),
  unpad_image (:109-145), pack_image_features (:265-330), processing_llava_next.py _get_unpadded_features
which the reference reaches through AutoProcessor / LlavaNextForConditionalGeneration
(/root/reference/src/models/_llava_hf.py:85-103, :340-376)."""

from __future__ import annotations

import numpy as np


def select_best_resolution(original_size: tuple, possible_resolutions) -> tuple:
    """(height, width) of the pinpoint that keeps the most image pixels, ties -> least padding."""
    oh, ow = original_size
    best, best_eff, best_waste = None, 0, float("inf")
    for h, w in possible_resolutions:
        scale = min(w / ow, h / oh)
        dw, dh = int(ow * scale), int(oh * scale)
        eff = min(dw * dh, ow * oh)
        waste = w * h - eff
        if eff > best_eff or (eff == best_eff and waste < best_waste):
            best, best_eff, best_waste = (h, w), eff, waste
    return best


def tile_grid(image_size: tuple, pinpoints, tile: int) -> tuple:
    """(tiles down, tiles across) of the anyres canvas for an image of (height, width)."""
    h, w = select_best_resolution(tuple(int(x) for x in image_size), pinpoints)
    return h // tile, w // tile


def unpad_bounds(original_size: tuple, cur_h: int, cur_w: int) -> tuple:
    """Row/column range [y0, y1) x [x0, x1) of the feature canvas that covers the un-padded image (unpad_image)."""
    oh, ow = original_size
    if ow / oh > cur_w / cur_h:
        new_h = int(round(oh * (cur_w / ow), 7))
        pad = (cur_h - new_h) // 2
        return pad, cur_h - pad, 0, cur_w
    new_w = int(round(ow * (cur_h / oh), 7))
    pad = (cur_w - new_w) // 2
    return 0, cur_h, pad, cur_w - pad


def packed_rows(image_size: tuple, pinpoints, tile: int, g: int, base_row: int, tokens: int, newline_row: int) -> np.ndarray:
    """Rows of the projected-feature buffer (view-major, `tokens` rows per view, CLS first) in the order
    pack_image_features concatenates them: base view, then the tile canvas row by row, un-padded, each
    canvas row followed by `newline_row`."""
    nh, nw = tile_grid(image_size, pinpoints, tile)
    base = base_row + 1 + np.arange(g * g, dtype=np.int64)
    y0, y1, x0, x1 = unpad_bounds(tuple(int(x) for x in image_size), nh * g, nw * g)
    ys, xs = np.arange(y0, y1)[:, None], np.arange(x0, x1)[None, :]
    view = 1 + (ys // g) * nw + (xs // g)                      # tile index (view 0 is the base image)
    rows = base_row + view * tokens + 1 + (ys % g) * g + (xs % g)
    rows = np.concatenate([rows, np.full((rows.shape[0], 1), newline_row, dtype=np.int64)], axis=1)
    return np.concatenate([base, rows.reshape(-1)])


def num_views(image_size: tuple, pinpoints, tile: int) -> int:
    nh, nw = tile_grid(image_size, pinpoints, tile)
    return 1 + nh * nw
