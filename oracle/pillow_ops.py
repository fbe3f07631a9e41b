"""TEST INFRASTRUCTURE -- not part of the product path.

numpy restatement of the Pillow (12.x, libImaging) pixel arithmetic the reference's strong
augmentation runs through torchvision on PIL images (/root/reference/datasets/DAcoco.py:348-360:
ColorJitter(0.4, 0.4, 0.4, 0.1), RandomGrayscale, GaussianBlur; torchvision's PIL branch is
ImageEnhance.{Brightness,Contrast,Color}, an HSV round trip for hue, Image.convert("L") and
ImageFilter.GaussianBlur).  Pillow is a third-party dependency of the reference (not vendored in
/root/reference); it IS installed here, so every function below is pinned against Pillow itself in
tests/test_strong_aug_cpu.py (the HSV pair exhaustively over all 2^24 colours).  csrc/strong_aug.hip
follows this arithmetic; only tests may import this file.
"""
import numpy as np


def luma(rgb):
    """Image.convert("L") for RGB (libImaging Convert.c rgb2l): ITU-R 601-2, 16-bit fixed point."""
    r, g, b = (rgb[..., i].astype(np.uint32) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend(degenerate, image, alpha):
    """Image.blend (libImaging Blend.c): float32 arithmetic, truncation inside [0, 1], clipped
    truncation when extrapolating."""
    a = np.float32(alpha)
    d = degenerate.astype(np.float32)
    t = (d + a * (image.astype(np.float32) - d).astype(np.float32)).astype(np.float32)
    if 0.0 <= alpha <= 1.0:
        return t.astype(np.int32).astype(np.uint8)
    return np.where(t <= 0, 0, np.where(t >= 255, 255, t.astype(np.int32))).astype(np.uint8)


def brightness(rgb, factor):
    return blend(np.zeros_like(rgb), rgb, factor)


def contrast_mean(rgb):
    """ImageEnhance.Contrast: int(mean of the L image + 0.5)."""
    L = luma(rgb)
    return int(int(L.sum(dtype=np.int64)) / L.size + 0.5)


def contrast(rgb, factor):
    return blend(np.full_like(rgb, contrast_mean(rgb)), rgb, factor)


def saturation(rgb, factor):
    return blend(np.repeat(luma(rgb)[..., None], 3, axis=-1), rgb, factor)


def grayscale3(rgb):
    return np.repeat(luma(rgb)[..., None], 3, axis=-1)


def rgb_to_hsv(rgb):
    """libImaging Convert.c rgb2hsv_row: float32 ratios, double for the hue fold, truncation."""
    r, g, b = (rgb[..., i] for i in range(3))
    maxc = np.maximum(r, np.maximum(g, b))
    minc = np.minimum(r, np.minimum(g, b))
    grey = maxc == minc
    cr = np.where(grey, 1, maxc.astype(np.int32) - minc).astype(np.float32)
    mx = np.where(maxc == 0, 1, maxc).astype(np.float32)
    s = (cr / mx).astype(np.float32)
    rc = ((maxc.astype(np.int32) - r).astype(np.float32) / cr).astype(np.float32)
    gc = ((maxc.astype(np.int32) - g).astype(np.float32) / cr).astype(np.float32)
    bc = ((maxc.astype(np.int32) - b).astype(np.float32) / cr).astype(np.float32)
    h = np.where(r == maxc, (bc - gc).astype(np.float32).astype(np.float64),
                 np.where(g == maxc, 2.0 + rc.astype(np.float64) - bc.astype(np.float64),
                          4.0 + gc.astype(np.float64) - rc.astype(np.float64))).astype(np.float32)
    h = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(np.float32)
    uh = np.clip((h.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    us = np.clip((s.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    return np.stack([np.where(grey, 0, uh), np.where(grey, 0, us), maxc], -1).astype(np.uint8)


def _round_half_away(x):
    return np.where(x >= 0, np.floor(x + 0.5), np.ceil(x - 0.5)).astype(np.int32)


def hsv_to_rgb(hsv):
    """libImaging Convert.c hsv2rgb."""
    h, s, v = (hsv[..., i] for i in range(3))
    hf = h.astype(np.float32).astype(np.float64) * 6.0 / 255.0
    i = np.floor(hf).astype(np.int32)
    f = (hf - i.astype(np.float32).astype(np.float64)).astype(np.float32).astype(np.float64)
    fs = (s.astype(np.float32).astype(np.float64) / 255.0).astype(np.float32).astype(np.float64)
    vd = v.astype(np.float64)
    p = np.clip(_round_half_away(vd * (1.0 - fs)), 0, 255)
    q = np.clip(_round_half_away(vd * (1.0 - fs * f)), 0, 255)
    t = np.clip(_round_half_away(vd * (1.0 - fs * (1.0 - f))), 0, 255)
    vi = v.astype(np.int32)
    k = i % 6
    table = [(vi, t, p), (q, vi, p), (p, vi, t), (p, q, vi), (t, p, vi), (vi, p, q)]
    out = np.empty(hsv.shape, dtype=np.uint8)
    for c in range(3):
        ch = np.select([k == n for n in range(6)], [table[n][c] for n in range(6)])
        out[..., c] = np.where(s == 0, vi, ch)
    return out


def hue_shift_byte(hue_factor):
    """torchvision's `np_h += np.uint8(hue_factor * 255)`: truncate toward zero, wrap to a byte."""
    return int(hue_factor * 255) & 0xFF


def hue(rgb, hue_factor):
    hsv = rgb_to_hsv(rgb)
    hsv[..., 0] = (hsv[..., 0].astype(np.int32) + hue_shift_byte(hue_factor)) & 0xFF
    return hsv_to_rgb(hsv)


def gaussian_box_radius(sigma, passes=3):
    """libImaging BoxBlur.c _gaussian_blur_radius (float32 variables, double constants)."""
    f = np.float32
    sigma2 = f(f(sigma) * f(sigma) / passes)
    L = f(np.sqrt(12.0 * float(sigma2) + 1.0))
    l = f(np.floor((float(L) - 1.0) / 2.0))
    a = f(f(2 * l + 1) * f(f(l * f(l + 1)) - f(3 * sigma2)))
    a = f(a / f(6 * f(sigma2 - f(f(l + 1) * f(l + 1)))))
    return float(f(l + a))


def box_weights(float_radius):
    """(radius, ww, fw) of libImaging BoxBlur.c ImagingLineBoxBlur32: 8.24 fixed point."""
    radius = int(float_radius)
    ww = int(np.uint32(np.float32(1 << 24) / np.float32(np.float32(float_radius) * 2 + 1)))
    fw = (((1 << 24) - (radius * 2 + 1) * ww) // 2) & 0xFFFFFFFF
    return radius, ww, fw


def box_blur_axis(img, float_radius, axis):
    """One pass of the extended box blur along `axis` with edge clamping."""
    radius, ww, fw = box_weights(float_radius)
    n = img.shape[axis]
    idx = np.arange(n)
    src = img.astype(np.uint64)
    acc = np.zeros(img.shape, dtype=np.uint64)
    for d in range(-radius, radius + 1):
        acc += np.take(src, np.clip(idx + d, 0, n - 1), axis=axis)
    far = np.take(src, np.clip(idx - radius - 1, 0, n - 1), axis=axis) + np.take(src, np.clip(idx + radius + 1, 0, n - 1), axis=axis)
    bulk = (acc * ww + far * fw) & 0xFFFFFFFF
    return (((bulk + (1 << 23)) & 0xFFFFFFFF) >> 24).astype(np.uint8)


def gaussian_blur(rgb, sigma, passes=3):
    """ImageFilter.GaussianBlur(radius=sigma): `passes` horizontal box passes then `passes` vertical."""
    fr = gaussian_box_radius(sigma, passes)
    out = rgb
    if fr != 0:
        for _ in range(passes):
            out = box_blur_axis(out, fr, 1)
        for _ in range(passes):
            out = box_blur_axis(out, fr, 0)
    return out
