"""GPU: csrc/strong_aug.hip against Pillow itself (installed on the GPU box as in the build
container), i.e. against what the reference's strong augmentation computes on PIL images
(/root/reference/datasets/DAcoco.py:330-360 via torchvision's PIL branch: ImageEnhance.*, the HSV
round trip, convert("L"), ImageFilter.GaussianBlur) -- bit for bit -- and against the numpy
restatement in oracle/pillow_ops.py where Pillow has no single call (the fused chain)."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
PIL = pytest.importorskip("PIL")
from PIL import Image, ImageEnhance, ImageFilter  # noqa: E402


def _image(seed, h=203, w=331):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    a[: h // 3] = (a[: h // 3] // 32) * 32
    a[h // 3: h // 2, : w // 2] = rng.integers(0, 256, 3, dtype=np.uint8)
    return a


def _dev(a):
    return torch.from_numpy(a).to("cuda:0")


def _pil_hue(im, hue):
    h, s, v = im.convert("HSV").split()
    np_h = np.array(h, dtype=np.uint8)
    with np.errstate(over="ignore"):
        np_h += np.array(int(hue * 255)).astype(np.uint8)
    return Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert("RGB")


def _pil_op(im, code, factor):
    from datr_amd import strong_aug as S
    if code == S.BRIGHTNESS:
        return ImageEnhance.Brightness(im).enhance(factor)
    if code == S.CONTRAST:
        return ImageEnhance.Contrast(im).enhance(factor)
    if code == S.SATURATION:
        return ImageEnhance.Color(im).enhance(factor)
    if code == S.HUE:
        return _pil_hue(im, factor)
    L = np.asarray(im.convert("L"))
    return Image.fromarray(np.dstack([L, L, L]), "RGB")


@pytest.mark.parametrize("factor", [0.6, 0.8137, 1.0, 1.17, 1.4, 0.0, 2.5])
@pytest.mark.parametrize("hw", [(203, 331), (7, 5), (1, 1)])
def test_single_ops_match_pillow(factor, hw):
    from datr_amd import strong_aug as S
    a = _image(int(factor * 100), *hw)
    im = Image.fromarray(a)
    for code in (S.BRIGHTNESS, S.CONTRAST, S.SATURATION, S.GRAYSCALE):
        got = S.pixel_ops_on_device(_dev(a), [(code, factor)]).cpu().numpy()
        assert np.array_equal(got, np.asarray(_pil_op(im, code, factor))), code


def test_hue_is_exact_for_every_colour():
    """All 2^24 colours through RGB -> HSV -> (+shift) -> RGB, five shifts."""
    from datr_amd import strong_aug as S
    g = np.arange(1 << 24, dtype=np.uint32)
    cube = np.stack([(g >> 16) & 255, (g >> 8) & 255, g & 255], -1).astype(np.uint8).reshape(4096, 4096, 3)
    im = Image.fromarray(cube)
    d = _dev(cube)
    for hue in (0.0, 0.1, -0.1, 0.0371, -0.5):
        got = S.pixel_ops_on_device(d, [(S.HUE, hue)]).cpu().numpy()
        want = np.asarray(_pil_hue(im, hue))
        assert np.array_equal(got, want), f"hue {hue}: {(got != want).any(-1).sum()} colours differ"


def test_fused_chains_match_sequential_pillow():
    """The chains the reference's pipeline draws (ColorJitter in a random order, then grayscale)
    in ONE launch against the same operations applied one after another by Pillow; the contrast
    mean is taken at its place in the chain."""
    from datr_amd import strong_aug as S
    jitter, gray = S.ColorJitter(0.4, 0.4, 0.4, 0.1), S.RandomGrayscale(p=0.5)
    a = _image(3, 301, 407)
    orders = set()
    for seed in range(12):
        torch.manual_seed(seed)
        ops = jitter.draw() + gray.draw()
        orders.add(tuple(c for c, _ in ops))
        im = Image.fromarray(a)
        for code, factor in ops:
            im = _pil_op(im, code, factor)
        got = S.pixel_ops_on_device(_dev(a), ops).cpu().numpy()
        assert np.array_equal(got, np.asarray(im)), (seed, ops)
    assert len(orders) >= 6
    # two contrast steps in one chain, and more operations than one launch holds
    ops = [(S.CONTRAST, 1.3), (S.HUE, 0.05), (S.CONTRAST, 0.7), (S.SATURATION, 1.2)]
    im = Image.fromarray(a)
    for code, factor in ops:
        im = _pil_op(im, code, factor)
    assert np.array_equal(S.pixel_ops_on_device(_dev(a), ops).cpu().numpy(), np.asarray(im))
    with pytest.raises(ValueError):
        S.pixel_ops_on_device(_dev(a), [(S.BRIGHTNESS, 1.1)] * 9)


@pytest.mark.parametrize("sigma", [0.1, 0.3, 0.57, 0.9, 1.3, 1.77, 2.0, 3.7, 6.0])
@pytest.mark.parametrize("hw", [(203, 331), (64, 128), (5, 3), (1, 40), (33, 1)])
def test_gaussian_blur_matches_pillow(sigma, hw):
    from datr_amd.strong_aug import gaussian_blur_on_device
    a = _image(int(sigma * 10), *hw)
    want = np.asarray(Image.fromarray(a).filter(ImageFilter.GaussianBlur(radius=sigma)))
    got = gaussian_blur_on_device(_dev(a), sigma).cpu().numpy()
    assert np.array_equal(got, want), f"{(got != want).sum()} bytes differ, max {np.abs(got.astype(int) - want).max()}"


def test_strong_pipeline_and_pair_at_full_size():
    """make_strong_transforms on a Cityscapes-sized frame, seed by seed against the same draws
    replayed through Pillow; then the (image, strong image) pair through the geometric pipeline."""
    from datr_amd import strong_aug as S
    from datr_amd.transforms import make_train_transforms
    a = _image(11, 1024, 2048)
    tf = S.make_strong_transforms("train")
    jit_apply, gray, _ = tf.transforms
    changed = 0
    for seed in range(8):
        torch.manual_seed(seed)
        random.seed(seed)
        got = tf(_dev(a))
        torch.manual_seed(seed)
        random.seed(seed)
        im = Image.fromarray(a)
        for code, factor in jit_apply.draw() + gray.draw():
            im = _pil_op(im, code, factor)
        if not (0.5 < torch.rand(1)):
            im = im.filter(ImageFilter.GaussianBlur(radius=random.uniform(0.1, 2.0)))
        want = np.asarray(im)
        assert np.array_equal(got.cpu().numpy(), want), seed
        changed += not np.array_equal(want, a)
    assert changed >= 6
    random.seed(1)
    torch.manual_seed(1)
    img, strong, tgt = make_train_transforms()(_dev(a), got, {"boxes": torch.tensor([[10., 20., 500., 700.]]),
                                                              "labels": torch.tensor([1])})
    assert strong.shape == img.shape and strong.dtype == torch.uint8 and tgt["size"].tolist() == list(img.shape[:2])
