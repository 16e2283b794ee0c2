"""Host-side image preparation in front of the GPU patchify kernel.

Restates the integer/resampling decisions of the third-party code the reference goes through
(`src/models/_qwen2_vl.py:237-250,292,299-305`): the JPEG round trip, `qwen_vl_utils.smart_resize`
(0.0.8) and HF `Qwen2VLImageProcessor.smart_resize` (image_processing_qwen2_vl.py:62-88).  The float
work (rescale, normalise, patchify) is `owc_patchify_u8` on the GPU."""

from __future__ import annotations

import math
from io import BytesIO

import numpy as np

IMAGE_FACTOR = 28
OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def smart_resize(height: int, width: int, factor: int = IMAGE_FACTOR, min_pixels: int = 56 * 56,
                 max_pixels: int = 14 * 14 * 4 * 1280) -> tuple[int, int]:
    """Round both sides to multiples of `factor`, keep the area within [min_pixels, max_pixels], keep the aspect."""
    if max(height, width) / min(height, width) > 200:
        raise ValueError("absolute aspect ratio must be smaller than 200")
    h_bar = max(factor, round(height / factor) * factor)
    w_bar = max(factor, round(width / factor) * factor)
    if h_bar * w_bar > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        h_bar = max(factor, math.floor(height / beta / factor) * factor)
        w_bar = max(factor, math.floor(width / beta / factor) * factor)
    elif h_bar * w_bar < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h_bar = math.ceil(height * beta / factor) * factor
        w_bar = math.ceil(width * beta / factor) * factor
    return h_bar, w_bar


def jpeg_round_trip(img):
    """PIL -> JPEG bytes -> PIL, as the reference's base64 data-URI detour does (lossy on purpose: parity).

    The encoder writes to an anonymous in-memory FILE (`memfd_create`), not to a `BytesIO`: with a real file descriptor Pillow
    runs the whole encode loop outside the GIL (`encode_to_file`, ImageFile._save), with a `BytesIO` it calls the encoder chunk by
    chunk HOLDING it - one rank's PIL workers then encode one image at a time however many there are (measured on 8 cores, Food-101
    sizes: 610 encodes/s at 1, 2, 4 and 8 threads into a BytesIO; 540 / 1690 / 2020 at 1 / 4 / 8 threads into a memfd; this round trip
    was the ~400 images/s per-rank ceiling of tools/soak_host_ranks.py).  Same encoder, same parameters: the JPEG bytes are
    identical (tests/test_host_logic.py::test_jpeg_round_trip_bytes_do_not_depend_on_where_the_encoder_writes)."""
    import os

    from PIL import Image

    rgb = img.convert("RGB")
    if hasattr(os, "memfd_create") and os.environ.get("OWC_JPEG_BYTESIO", "0") in ("", "0"):   # (1: the A side of tools/run_host_soak_r6.sh)
        with os.fdopen(os.memfd_create("owc-jpeg"), "w+b") as f:
            rgb.save(f, format="JPEG")
            f.seek(0)
            return Image.open(f).convert("RGB")     # (convert() loads the pixels: the file can go)
    buf = BytesIO()
    rgb.save(buf, format="JPEG")
    buf.seek(0)
    return Image.open(buf).convert("RGB")


def prepare_image(img, min_pixels: int, max_pixels: int, jpeg: bool = True) -> np.ndarray:
    """PIL image -> uint8 CHW array whose sides are multiples of 28 (bicubic, two stages like the reference)."""
    from PIL import Image

    if jpeg:
        img = jpeg_round_trip(img)
    else:
        img = img.convert("RGB")
    w, h = img.size
    # stage 1: qwen_vl_utils.fetch_image (defaults: min 4*28*28, max 16384*28*28)
    h1, w1 = smart_resize(h, w, IMAGE_FACTOR, 4 * 28 * 28, 16384 * 28 * 28)
    if (h1, w1) != (h, w):
        img = img.resize((w1, h1), Image.BICUBIC)
    # stage 2: the HF image processor with the wrapper's min/max pixels
    h2, w2 = smart_resize(h1, w1, IMAGE_FACTOR, min_pixels, max_pixels)
    if (h2, w2) != (h1, w1):
        img = img.resize((w2, h2), Image.BICUBIC)
    return np.ascontiguousarray(np.asarray(img, dtype=np.uint8).transpose(2, 0, 1))


# --------------------------------------------------------------------------------------------------
# LLaVA (CLIP image processor).  Restates the resampling / cropping / tiling decisions of HF
# CLIPImageProcessor (image_processing_clip.py: resize shortest edge -> center crop) and
# LlavaNextImageProcessor (image_processing_llava_next.py: select_best_resolution, _resize_for_patching,
# _pad_for_patching, divide_to_patches, base view) which the reference reaches through
# `self.processor(images=visuals, text=text)` (/root/reference/src/models/_llava_hf.py:347).  Everything stays
# uint8; rescale + normalise + patch layout are `owc_clip_patchify_u8` on the GPU.
def clip_view(img, size: int = 336) -> np.ndarray:
    """PIL image -> uint8 [3, size, size]: shortest edge to `size` (bicubic), then center crop."""
    from PIL import Image

    img = img.convert("RGB")
    w, h = img.size
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
    if (nw, nh) != (w, h):
        img = img.resize((nw, nh), Image.BICUBIC)
    a = np.asarray(img, dtype=np.uint8)
    top, left = (nh - size) // 2, (nw - size) // 2
    return np.ascontiguousarray(a[top:top + size, left:left + size].transpose(2, 0, 1))


def anyres_views(img, pinpoints, tile: int = 336) -> tuple[np.ndarray, tuple[int, int]]:
    """PIL image -> (uint8 [1 + nh*nw, 3, tile, tile], (height, width)): view 0 is the whole image squashed to
    tile x tile, the rest are the row-major tiles of the aspect-preserving resize, zero-padded to the pinpoint."""
    from PIL import Image

    from ..engine.anyres import select_best_resolution

    img = img.convert("RGB")
    w, h = img.size
    th, tw = select_best_resolution((h, w), pinpoints)
    sw, sh = tw / w, th / h
    if sw < sh:
        nw, nh = tw, min(math.ceil(h * sw), th)
    else:
        nh, nw = th, min(math.ceil(w * sh), tw)
    resized = np.asarray(img.resize((nw, nh), Image.BICUBIC), dtype=np.uint8)
    canvas = np.zeros((th, tw, 3), dtype=np.uint8)
    py, px = (th - nh) // 2, (tw - nw) // 2
    canvas[py:py + nh, px:px + nw] = resized
    base = np.asarray(img.resize((tile, tile), Image.BICUBIC), dtype=np.uint8)
    tiles = canvas.reshape(th // tile, tile, tw // tile, tile, 3).transpose(0, 2, 1, 3, 4).reshape(-1, tile, tile, 3)
    views = np.concatenate([base[None], tiles], 0).transpose(0, 3, 1, 2)
    return np.ascontiguousarray(views), (h, w)
