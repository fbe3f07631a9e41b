"""Domain-adaptation pieces of DATR: gradient reversal, the image-level discriminator and
class-wise query prototypes.

Mirror of /root/reference/models/dino/DA_utils.py (`decompose_features` :5-31, `GradReverse`
:33-43, `FCDiscriminator_img` :61-79, `get_prototype_class_wise` :82-120).  The prototype
extraction computes the same class-wise means with one [C, B*N] x [B*N, 256] product instead
of materialising the reference's [B*N, C, 256] masked copy (:96-108); sums are reassociated,
so values agree to fp32 rounding, the argmax class map and counts are exact.
"""
from __future__ import annotations

import torch
from torch import nn


def decompose_features(srcs, masks, poss):
    """Split every level's batch into (source half, all, target half)."""
    half = srcs[0].shape[0] // 2
    src = ([s[:half] for s in srcs], [m[:half] for m in masks], [p[:half] for p in poss])
    tgt = ([s[half:] for s in srcs], [m[half:] for m in masks], [p[half:] for p in poss])
    return (*src, srcs, masks, poss, *tgt)


class GradReverse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output.neg()


def grad_reverse(x):
    return GradReverse.apply(x)


class FCDiscriminator_img(nn.Module):
    """3x3 convs 256->256->128->128->1 with LeakyReLU(0.2) between (per-pixel domain logit)."""

    def __init__(self, num_classes, ndf1=256, ndf2=128):
        super().__init__()
        self.conv1 = nn.Conv2d(num_classes, ndf1, kernel_size=3, padding=1)
        self.conv2 = nn.Conv2d(ndf1, ndf2, kernel_size=3, padding=1)
        self.conv3 = nn.Conv2d(ndf2, ndf2, kernel_size=3, padding=1)
        self.classifier = nn.Conv2d(ndf2, 1, kernel_size=3, padding=1)
        self.leaky_relu = nn.LeakyReLU(negative_slope=0.2, inplace=True)

    def forward(self, x):
        x = self.leaky_relu(self.conv1(x))
        x = self.leaky_relu(self.conv2(x))
        x = self.leaky_relu(self.conv3(x))
        return self.classifier(x)


def get_prototype_class_wise(object_query_last_layer, outputs_class, num_classes,
                             global_proto=None, global_amount=None):
    """Class-wise mean of the last-layer queries, with classes assigned by argmax of the
    predicted scores; also advances the running (count-weighted) global prototypes.

    Returns (prototypes [C,256], present [C] in {0,1}, new_global [C,256] detached,
             new_amount [C], onehot [B*N, C])."""
    B, N, C = object_query_last_layer.shape
    labels = torch.argmax(outputs_class.sigmoid(), dim=2).reshape(B * N, 1)
    feats = object_query_last_layer.reshape(B * N, C)
    onehot = torch.zeros(B * N, num_classes, device=feats.device)
    onehot.scatter_(dim=1, index=labels, value=1)
    count = onehot.sum(0)                                          # [num_classes]
    present = torch.where(count != 0, torch.ones_like(count), count)
    denom = torch.where(count == 0, torch.ones_like(count), count)
    prototypes = (onehot.t() @ feats) / denom[:, None]
    weight = count / (count + global_amount)
    weight = torch.where(count == 0, torch.zeros_like(weight), weight)[:, None]
    new_global = (global_proto * (1 - weight) + prototypes * weight).detach()
    return prototypes, present, new_global, global_amount + count, onehot
