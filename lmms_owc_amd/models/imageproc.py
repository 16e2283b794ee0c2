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
    """PIL -> JPEG bytes -> PIL, as the reference's base64 data-URI detour does (lossy on purpose: parity)."""
    from PIL import Image

    buf = BytesIO()
    img.convert("RGB").save(buf, format="JPEG")
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
