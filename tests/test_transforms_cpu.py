"""CPU: the random decisions and target arithmetic of datr_amd.transforms against the reference
fixture (tests/golden/transforms.npz, see test_transforms_gpu.py), with the two device image ops
replaced by Pillow calls -- what is checked here is the host logic: RNG order, size rule, box /
area / label updates, the crop's box filter."""
import os
import random

import numpy as np
import pytest
import torch

PIL = pytest.importorskip("PIL")
from PIL import Image  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "transforms.npz")


def _pillow_resize(image, size_hw, flip=False):
    pil = Image.fromarray(image.numpy())
    if flip:
        pil = pil.transpose(Image.FLIP_LEFT_RIGHT)
    return torch.from_numpy(np.asarray(pil.resize((int(size_hw[1]), int(size_hw[0])), Image.BILINEAR)).copy())


def test_host_logic_matches_reference_fixture(monkeypatch):
    import datr_amd.transforms as T
    monkeypatch.setattr(T, "resize_uint8_on_device", _pillow_resize)
    z = np.load(GOLDEN)
    tf = T.make_train_transforms(z["cfg/data_aug_scales"].tolist(), int(z["cfg/data_aug_max_size"]),
                                 z["cfg/data_aug_scales2_resize"].tolist(), z["cfg/data_aug_scales2_crop"].tolist())
    mean = torch.tensor(T.IMAGENET_MEAN).view(3, 1, 1)
    std = torch.tensor(T.IMAGENET_STD).view(3, 1, 1)
    for seed in z["seeds"].tolist():
        p = f"s{seed}/"
        boxes = torch.from_numpy(z[p + "boxes_in"])
        target = {"boxes": boxes, "labels": torch.from_numpy(z[p + "labels_in"]),
                  "area": (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]),
                  "iscrowd": torch.zeros(len(boxes), dtype=torch.int64)}
        random.seed(seed)
        torch.manual_seed(seed)
        img, _, tgt = tf(torch.from_numpy(z[p + "image_in"]), None, target)
        got = (img.permute(2, 0, 1).float().div(255) - mean) / std
        assert torch.equal(got, torch.from_numpy(z[p + "image_out"])), seed
        for k in ("boxes", "labels", "area", "size", "iscrowd"):
            assert torch.equal(tgt[k], torch.from_numpy(z[p + k])), (seed, k)


def test_crop_clips_and_drops_empty_boxes():
    from datr_amd.transforms import crop
    img = torch.zeros(50, 60, 3, dtype=torch.uint8)
    t = {"boxes": torch.tensor([[0., 0., 10., 10.], [20., 20., 40., 45.], [55., 5., 59., 9.]]),
         "labels": torch.tensor([1, 2, 3]), "area": torch.tensor([100., 500., 16.]), "iscrowd": torch.zeros(3, dtype=torch.int64)}
    out, _, tt = crop(img, None, t, (15, 12, 20, 30))                    # top 15, left 12, 20 x 30
    assert out.shape == (20, 30, 3) and tt["size"].tolist() == [20, 30]
    assert tt["boxes"].tolist() == [[8., 5., 28., 20.]] and tt["labels"].tolist() == [2] and tt["area"].tolist() == [300.]
    assert t["boxes"].shape == (3, 4)                           # the caller's target is untouched
