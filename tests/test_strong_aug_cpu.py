"""CPU: oracle/pillow_ops.py (the arithmetic csrc/strong_aug.hip follows) against Pillow itself --
the library the reference's strong augmentation runs on (/root/reference/datasets/DAcoco.py:330-360
through torchvision's PIL branch) -- and the host half of datr_amd.strong_aug (box radius / weights,
the order of the random draws)."""
import random

import numpy as np
import pytest
import torch

PIL = pytest.importorskip("PIL")
from PIL import Image, ImageEnhance, ImageFilter  # noqa: E402

from oracle import pillow_ops as P  # noqa: E402


def _image(seed, h=97, w=131):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    a[: h // 3] = (a[: h // 3] // 32) * 32
    a[h // 3: h // 2, : w // 2] = rng.integers(0, 256, 3, dtype=np.uint8)       # a flat patch (grey / hue edge cases)
    return a


@pytest.mark.parametrize("factor", [0.6, 0.81, 1.0, 1.17, 1.4, 0.0, 2.5])
def test_enhancers_and_luma(factor):
    a = _image(int(factor * 100))
    im = Image.fromarray(a)
    assert np.array_equal(P.luma(a), np.asarray(im.convert("L")))
    assert np.array_equal(P.brightness(a, factor), np.asarray(ImageEnhance.Brightness(im).enhance(factor)))
    assert np.array_equal(P.contrast(a, factor), np.asarray(ImageEnhance.Contrast(im).enhance(factor)))
    assert np.array_equal(P.saturation(a, factor), np.asarray(ImageEnhance.Color(im).enhance(factor)))


def test_hsv_round_trip_is_exhaustively_exact():
    g = np.arange(1 << 24, dtype=np.uint32)
    cube = np.stack([(g >> 16) & 255, (g >> 8) & 255, g & 255], -1).astype(np.uint8).reshape(4096, 4096, 3)
    assert np.array_equal(P.rgb_to_hsv(cube), np.asarray(Image.fromarray(cube).convert("HSV")))
    assert np.array_equal(P.hsv_to_rgb(cube), np.asarray(Image.fromarray(cube, "HSV").convert("RGB")))


@pytest.mark.parametrize("hue", [-0.1, -0.0371, 0.0, 0.002, 0.1, 0.5, -0.5])
def test_hue_matches_torchvision_pil_recipe(hue):
    a = _image(7)
    h, s, v = Image.fromarray(a).convert("HSV").split()
    np_h = np.array(h, dtype=np.uint8)
    with np.errstate(over="ignore"):
        np_h += np.array(int(hue * 255)).astype(np.uint8)       # np.uint8(hue_factor * 255) on x86
    want = np.asarray(Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert("RGB"))
    assert np.array_equal(P.hue(a, hue), want)


@pytest.mark.parametrize("sigma", [0.1, 0.3, 0.57, 0.9, 1.3, 1.77, 2.0, 3.7, 6.0])
@pytest.mark.parametrize("hw", [(97, 131), (5, 3), (1, 40)])
def test_gaussian_blur(sigma, hw):
    a = _image(int(sigma * 10), *hw)
    want = np.asarray(Image.fromarray(a).filter(ImageFilter.GaussianBlur(radius=sigma)))
    assert np.array_equal(P.gaussian_blur(a, sigma), want)


def test_host_blur_parameters_follow_the_oracle():
    from datr_amd.strong_aug import box_weights, gaussian_box_radius
    for sigma in np.linspace(0.1, 8.0, 80):
        fr = gaussian_box_radius(float(sigma))
        assert fr == P.gaussian_box_radius(float(sigma))
        assert box_weights(fr) == P.box_weights(fr)


def test_random_draw_order():
    """RandomApply -> randperm(4) -> four uniform_ -> RandomGrayscale -> RandomApply -> random.uniform."""
    from datr_amd import strong_aug as S
    tf = S.make_strong_transforms("train")
    jit_apply, gray, blur_apply = tf.transforms
    for seed in range(20):
        torch.manual_seed(seed)
        random.seed(seed)
        ops = jit_apply.draw() + gray.draw()
        fired = not (0.5 < torch.rand(1))
        torch.manual_seed(seed)
        want = []
        if not (0.8 < torch.rand(1)):
            order = torch.randperm(4).tolist()
            fac = [float(torch.empty(1).uniform_(lo, hi)) for lo, hi in ((0.6, 1.4), (0.6, 1.4), (0.6, 1.4), (-0.1, 0.1))]
            want = [(k, fac[k]) for k in order]
        if torch.rand(1) < 0.2:
            want.append((S.GRAYSCALE, 0.0))
        assert ops == want
        assert fired == (not (0.5 < torch.rand(1)))
    assert S.make_strong_transforms("val") is None
    with pytest.raises(ValueError):
        S.make_strong_transforms("test")
