"""Texture decoding for the scene loader (host side, not on the hot path).

The reference decodes textures with the `image` crate at load time (image::open,
/root/reference/src/scene.rs:1053,1065) and samples them through DynamicImage::get_pixel, which
yields RGBA with alpha 255 for RGB and Luma images; row 0 is the top row.  Here Pillow does the
decoding and the C ABI receives RGBA8 texels.
"""
import os

import numpy as np


def procedural_uv_grid(width=1024, height=512):
    """Deterministic stand-in for textures the reference repository does not ship
    (resources/universe_dim.jpg is listed in its .MISSING_LARGE_BLOBS).  Matches
    euclider::procedural_uv_grid in csrc/scene_host.cpp texel for texel."""
    y, x = np.mgrid[0:height, 0:width]
    r = (x * 255 // max(1, width - 1)).astype(np.uint8)
    g = (y * 255 // max(1, height - 1)).astype(np.uint8)
    b = (64 + 128 * (((x // 32) + (y // 32)) % 2)).astype(np.uint8)
    img = np.stack([r, g, b, np.full_like(r, 255)], axis=-1)
    img[(x % 64 == 0) | (y % 64 == 0)] = (255, 255, 255, 255)
    return np.ascontiguousarray(img)


def load_rgba(path, search_dirs):
    """Returns an (h, w, 4) uint8 array, or None when the file cannot be found."""
    from PIL import Image
    for d in search_dirs:
        p = os.path.normpath(os.path.join(d, path))
        if os.path.exists(p):
            return np.ascontiguousarray(np.asarray(Image.open(p).convert("RGBA"), dtype=np.uint8))
    return None
