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


# ---- a whole domain-adaptation batch: CocoDetection.__getitem__ (augmentation half) + DADataset + collate_fn_da ----
def make_da_batch():
    """tests/golden/da_batch.npz: two (source, target) pairs through the reference's train transforms
    (the target with a strongly augmented PIL copy -- fixed Pillow calls here, torchvision's random
    ColorJitter is absent) and the reference's `collate_fn_da` (/root/reference/util/misc.py:291-300)."""
    from PIL import ImageEnhance, ImageFilter
    import util.misc as ref_misc
    args = argparse.Namespace(data_aug_scales=[24, 28, 32], data_aug_max_size=50,
                              data_aug_scales2_resize=[20, 25, 30], data_aug_scales2_crop=[18, 30])
    tf = make_coco_transforms("train", args=args)
    out = {}
    for k, v in vars(args).items():
        out["cfg/" + k] = np.asarray(v, dtype=np.int64)
    random.seed(5)
    torch.manual_seed(5)
    items = []
    for n in range(2):
        pair = []
        for dom in ("source", "target"):
            rng = np.random.default_rng(100 + 2 * n + (dom == "target"))
            h, w = int(rng.integers(36, 48)), int(rng.integers(48, 64))
            img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
            boxes = np.array([[3., 4., w * 0.6, h * 0.7], [w * 0.3, h * 0.2, w - 2., h - 3.]], dtype=np.float32)
            out[f"{dom}{n}/image"], out[f"{dom}{n}/boxes"] = img, boxes
            b = torch.from_numpy(boxes)
            tgt = {"boxes": b, "labels": torch.tensor([1 + n, 3]), "area": (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]),
                   "iscrowd": torch.zeros(2, dtype=torch.int64), "size": torch.tensor([h, w])}
            pil = Image.fromarray(img)
            strong = None
            if dom == "target":
                strong = ImageEnhance.Contrast(ImageEnhance.Color(pil).enhance(1.3)).enhance(0.8)
                strong = strong.filter(ImageFilter.GaussianBlur(radius=1.1))
            pair.append(tf(pil, strong, tgt))
        (s_img, _, s_lab), (t_img, t_strong, t_lab) = pair
        items.append((s_img, s_lab, t_img, t_lab, t_strong))
    samples, source_labels, target_labels, strong = ref_misc.collate_fn_da(items)
    out["samples/tensors"], out["samples/mask"] = samples.tensors.numpy(), samples.mask.numpy()
    out["strong/tensors"], out["strong/mask"] = strong.tensors.numpy(), strong.mask.numpy()
    for n in range(2):
        for dom, labs in (("source", source_labels), ("target", target_labels)):
            for key in ("boxes", "labels", "size"):
                out[f"{dom}{n}/out_{key}"] = labs[n][key].numpy()
    np.savez_compressed(os.path.join(OUT, "da_batch.npz"), **out)
    print("wrote da_batch.npz", os.path.getsize(os.path.join(OUT, "da_batch.npz")), "bytes", samples.tensors.shape)


if __name__ == "__main__":
    make_da_batch()
