"""The geometric part of the reference's training-time data pipeline on the device.

Mirror of /root/reference/datasets/da_transforms.py (`hflip` :62-81, `crop` :16-59, `resize`
:85-146, `RandomSizeCrop` :177-186, `RandomHorizontalFlip` :200-207, `RandomResize` :210-218,
`RandomSelect` :232-244, `ToTensor` + `Normalize` :247-287, `Compose` :290-298) and of
`make_coco_transforms('train')` (/root/reference/datasets/DAcoco.py:483-563) for the pipeline
without strong augmentation.  Images are uint8 [H, W, 3] DEVICE tensors instead of PIL images:
flip + resize run in csrc/resize.hip (bit-exact with Pillow's bilinear resampler, which is what
torchvision's `F.resize` calls for PIL images), the crop is a slice, ToTensor + Normalize (+ the
batch padding) run in csrc/preprocess.hip (`input_pipeline.collate_uint8_on_device`).  The random
decisions consume Python's `random` and torch's generator in the reference's order, so the same
seeds give the same augmentation; the target updates are the reference's arithmetic.  As in the
reference every transform takes and returns (image, image_strong_aug, target): the strongly augmented
copy of a target-domain image (datr_amd/strong_aug.py; None for the source domain) goes through the
same flips, resizes and crops as the image itself (DAcoco.py:391-398).
"""
from __future__ import annotations

import random

import torch

from .boxes import box_xyxy_to_cxcywh
from .input_pipeline import IMAGENET_MEAN, IMAGENET_STD, get_size_with_aspect_ratio, resize_uint8_on_device


def hflip(image, image_strong_aug, target):
    h, w = image.shape[:2]
    image = resize_uint8_on_device(image, (h, w), flip=True)
    if image_strong_aug is not None:
        image_strong_aug = resize_uint8_on_device(image_strong_aug, (h, w), flip=True)
    target = target.copy()
    if "boxes" in target:
        boxes = target["boxes"]
        target["boxes"] = boxes[:, [2, 1, 0, 3]] * torch.as_tensor([-1, 1, -1, 1]) + torch.as_tensor([w, 0, w, 0])
    return image, image_strong_aug, target


def crop(image, image_strong_aug, target, region):
    i, j, h, w = region
    image = image[i:i + h, j:j + w].contiguous()
    if image_strong_aug is not None:
        image_strong_aug = image_strong_aug[i:i + h, j:j + w].contiguous()
    target = target.copy()
    target["size"] = torch.tensor([h, w])
    fields = ["labels", "area", "iscrowd"]
    if "boxes" in target:
        max_size = torch.as_tensor([w, h], dtype=torch.float32)
        cropped = target["boxes"] - torch.as_tensor([j, i, j, i])
        cropped = torch.min(cropped.reshape(-1, 2, 2), max_size).clamp(min=0)
        target["area"] = (cropped[:, 1, :] - cropped[:, 0, :]).prod(dim=1)
        target["boxes"] = cropped.reshape(-1, 4)
        fields.append("boxes")
        keep = torch.all(cropped[:, 1, :] > cropped[:, 0, :], dim=1)
        for f in fields:
            if f in target:
                target[f] = target[f][keep]
    return image, image_strong_aug, target


def resize(image, image_strong_aug, target, size, max_size=None):
    h0, w0 = image.shape[:2]
    if isinstance(size, (list, tuple)):
        oh, ow = size[::-1]
    else:
        oh, ow = get_size_with_aspect_ratio((w0, h0), size, max_size)
    image = resize_uint8_on_device(image, (oh, ow))
    if image_strong_aug is not None:
        image_strong_aug = resize_uint8_on_device(image_strong_aug, (oh, ow))
    if target is None:
        return image, image_strong_aug, None
    rw, rh = float(ow) / float(w0), float(oh) / float(h0)
    target = target.copy()
    if "boxes" in target:
        target["boxes"] = target["boxes"] * torch.as_tensor([rw, rh, rw, rh])
    if "area" in target:
        target["area"] = target["area"] * (rw * rh)
    target["size"] = torch.tensor([oh, ow])
    return image, image_strong_aug, target


class RandomHorizontalFlip:
    def __init__(self, p=0.5):
        self.p = p

    def __call__(self, img, image_strong_aug, target):
        return hflip(img, image_strong_aug, target) if random.random() < self.p else (img, image_strong_aug, target)


class RandomResize:
    def __init__(self, sizes, max_size=None):
        assert isinstance(sizes, (list, tuple))
        self.sizes, self.max_size = sizes, max_size

    def __call__(self, img, image_strong_aug, target=None):
        return resize(img, image_strong_aug, target, random.choice(self.sizes), self.max_size)


class RandomSizeCrop:
    def __init__(self, min_size: int, max_size: int):
        self.min_size, self.max_size = min_size, max_size

    def __call__(self, img, image_strong_aug, target):
        H, W = img.shape[:2]
        w = random.randint(self.min_size, min(W, self.max_size))
        h = random.randint(self.min_size, min(H, self.max_size))
        if W == w and H == h:                      # torchvision.transforms.RandomCrop.get_params
            region = (0, 0, H, W)
        else:
            i = torch.randint(0, H - h + 1, size=(1,)).item()
            j = torch.randint(0, W - w + 1, size=(1,)).item()
            region = (i, j, h, w)
        return crop(img, image_strong_aug, target, region)


class RandomSelect:
    def __init__(self, transforms1, transforms2, p=0.5):
        self.transforms1, self.transforms2, self.p = transforms1, transforms2, p

    def __call__(self, img, image_strong_aug, target):
        if random.random() < self.p:
            return self.transforms1(img, image_strong_aug, target)
        return self.transforms2(img, image_strong_aug, target)


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, image, image_strong_aug, target):
        for t in self.transforms:
            image, image_strong_aug, target = t(image, image_strong_aug, target)
        return image, image_strong_aug, target


class NormalizeBoxes:
    """The target half of the reference's `Normalize`: xyxy pixels -> normalised cxcywh.  The image
    half (ToTensor + Normalize) happens when the batch is collated on the device."""

    def __call__(self, image, image_strong_aug, target=None):
        if target is None:
            return image, image_strong_aug, None
        target = target.copy()
        h, w = image.shape[:2]
        if "boxes" in target:
            target["boxes"] = box_xyxy_to_cxcywh(target["boxes"]) / torch.tensor([w, h, w, h], dtype=torch.float32)
        return image, image_strong_aug, target


def make_train_transforms(scales=(480, 512, 544, 576, 608, 640, 672, 704, 736, 768, 800), max_size=1333,
                          scales2_resize=(400, 500, 600), scales2_crop=(384, 600)):
    """`make_coco_transforms('train')` without strong augmentation (DAcoco.py:549-562), up to the
    image normalisation (done at collate time: input_pipeline.collate_uint8_on_device)."""
    return Compose([
        RandomHorizontalFlip(),
        RandomSelect(
            RandomResize(list(scales), max_size=max_size),
            Compose([RandomResize(list(scales2_resize)), RandomSizeCrop(*scales2_crop),
                     RandomResize(list(scales), max_size=max_size)])),
        NormalizeBoxes(),
    ])


def dataset_item(image, target, transforms, strong_transforms=None):
    """The augmentation half of the reference's `CocoDetection.__getitem__` (DAcoco.py:391-398):
    the strong transforms see the image BEFORE the geometric ones, then the pair shares them."""
    image_strong_aug = strong_transforms(image) if strong_transforms is not None else None
    if transforms is not None:
        image, image_strong_aug, target = transforms(image, image_strong_aug, target)
    return image, image_strong_aug, target


def da_item(source, target, transforms, strong_transforms=None):
    """`DADataset.__getitem__` (DAcoco.py:655-670) for decoded (image, target) pairs: the source
    item never gets a strong copy; the target's labels travel along only for their sizes."""
    source_img, _, source_label = dataset_item(source[0], source[1], transforms)
    target_img, target_img_strong_aug, target_label = dataset_item(target[0], target[1], transforms, strong_transforms)
    return source_img, source_label, target_img, target_label, target_img_strong_aug


__all__ = ["dataset_item", "da_item", "Compose", "RandomHorizontalFlip", "RandomResize", "RandomSizeCrop", "RandomSelect", "NormalizeBoxes",
           "make_train_transforms", "hflip", "crop", "resize", "IMAGENET_MEAN", "IMAGENET_STD"]
