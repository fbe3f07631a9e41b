"""CPU: the host half of the device resize -- Pillow's bilinear weight tables restated
(datr_amd.input_pipeline.pillow_coeffs) -- applied with numpy exactly as csrc/resize.hip applies
them, against Pillow itself (`F.resize` on PIL images in the reference's RandomResize,
/root/reference/datasets/da_transforms.py:108)."""
import numpy as np
import pytest

PIL = pytest.importorskip("PIL")
from PIL import Image  # noqa: E402


def _resample(img, oh, ow):
    from datr_amd.input_pipeline import pillow_coeffs
    H, W, _ = img.shape
    cur = img
    if ow != W:
        b, k, _ = pillow_coeffs(W, ow)
        out = np.empty((H, ow, 3), dtype=np.uint8)
        for xx in range(ow):
            x0, n = b[xx]
            acc = (1 << 21) + (cur[:, x0:x0 + n, :].astype(np.int64) * k[xx, :n, None].astype(np.int64)).sum(1)
            out[:, xx, :] = np.clip(acc >> 22, 0, 255)
        cur = out
    if oh != H:
        b, k, _ = pillow_coeffs(H, oh)
        out = np.empty((oh, cur.shape[1], 3), dtype=np.uint8)
        for yy in range(oh):
            y0, n = b[yy]
            acc = (1 << 21) + (cur[y0:y0 + n].astype(np.int64) * k[yy, :n, None, None].astype(np.int64)).sum(0)
            out[yy] = np.clip(acc >> 22, 0, 255)
        cur = out
    return cur


@pytest.mark.parametrize("hw,target", [((75, 100), (96, 128)), ((128, 256), (100, 200)), ((60, 91), (40, 91)),
                                       ((33, 51), (80, 124)), ((97, 131), (31, 45)), ((40, 40), (40, 125))])
def test_restated_pillow_bilinear_is_bit_exact(hw, target):
    rng = np.random.default_rng(hw[0] * 7 + hw[1])
    img = rng.integers(0, 256, size=(hw[0], hw[1], 3), dtype=np.uint8)
    want = np.asarray(Image.fromarray(img).resize((target[1], target[0]), Image.BILINEAR))
    assert np.array_equal(_resample(img, *target), want)


def test_random_resize_size_rule_and_box_updates():
    import torch
    from datr_amd.input_pipeline import get_size_with_aspect_ratio, hflip_boxes, resize_boxes
    assert get_size_with_aspect_ratio((2048, 1024), 800, 1333) == (666, 1332)      # max_size caps the long side
    assert get_size_with_aspect_ratio((640, 480), 480, 1333) == (480, 640)
    assert get_size_with_aspect_ratio((500, 375), 600, 1333) == (600, 800)
    assert get_size_with_aspect_ratio((375, 500), 600, 1333) == (800, 600)
    b = torch.tensor([[10., 20., 110., 220.]])
    assert hflip_boxes(b, 640).tolist() == [[530., 20., 630., 220.]]
    assert resize_boxes(b, (640, 480), (1280, 720)).tolist() == [[20., 30., 220., 330.]]
