"""Generate tests/golden/transforms.npz by running the REFERENCE's training transforms (build
container only): `make_coco_transforms('train')` (/root/reference/datasets/DAcoco.py:483-563, the
pipeline without strong augmentation) with its scale lists shrunk through the `args` hooks it
reads (`data_aug_scales`, ...), on synthetic uint8 images with boxes, for several seeds of Python's
`random` and torch's generator.  Records per sample: the source image, the boxes / labels / area in,
the normalised output image and the output target (boxes, labels, area, size).  torchvision is not
installed: oracle/ref_shims.py stands in its PIL-image branch of transforms.functional (Pillow calls).

    python tests/golden/make_golden_transforms.py
"""
import argparse
import os
import random
import sys

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_shims  # noqa: E402

ref_shims.install()
import datasets.da_transforms  # noqa: E402,F401  (the reference's; DAcoco imports it as T)
from datasets.DAcoco import make_coco_transforms  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
ARGS = argparse.Namespace(data_aug_scales=[48, 56, 64, 72, 80], data_aug_max_size=133,
                          data_aug_scales2_resize=[40, 50, 60], data_aug_scales2_crop=[38, 60])


def sample(seed):
    rng = np.random.default_rng(seed)
    h, w = int(rng.integers(70, 110)), int(rng.integers(90, 150))
    img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    n = int(rng.integers(1, 6))
    x0, y0 = rng.uniform(0, w * 0.7, n), rng.uniform(0, h * 0.7, n)
    bw, bh = rng.uniform(4, w * 0.4, n), rng.uniform(4, h * 0.4, n)
    boxes = np.stack([x0, y0, np.minimum(x0 + bw, w), np.minimum(y0 + bh, h)], 1).astype(np.float32)
    labels = rng.integers(1, 9, n).astype(np.int64)
    return img, boxes, labels


def main():
    tf = make_coco_transforms("train", args=ARGS)
    out = {"seeds": np.arange(12, dtype=np.int64)}
    for k, v in vars(ARGS).items():
        out["cfg/" + k] = np.asarray(v, dtype=np.int64)
    for seed in out["seeds"]:
        img, boxes, labels = sample(int(seed))
        b = torch.from_numpy(boxes)
        target = {"boxes": b, "labels": torch.from_numpy(labels),
                  "area": (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]), "iscrowd": torch.zeros(len(labels), dtype=torch.int64),
                  "orig_size": torch.tensor(img.shape[:2]), "size": torch.tensor(img.shape[:2])}
        random.seed(int(seed))
        torch.manual_seed(int(seed))
        image, _, tgt = tf(Image.fromarray(img), None, target)
        p = f"s{seed}/"
        out[p + "image_in"], out[p + "boxes_in"], out[p + "labels_in"] = img, boxes, labels
        out[p + "image_out"] = image.numpy()
        for key in ("boxes", "labels", "area", "size", "iscrowd"):
            out[p + key] = tgt[key].numpy()
    np.savez_compressed(os.path.join(OUT, "transforms.npz"), **out)
    print("wrote transforms.npz", os.path.getsize(os.path.join(OUT, "transforms.npz")), "bytes")


if __name__ == "__main__":
    main()
