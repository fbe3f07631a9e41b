"""Deterministic synthetic weights and inputs shared by the golden generator (which loads them
into the REFERENCE model) and the tests (which load them into datr_amd's model).  Nothing is
stored: every tensor is regenerated from a CRC of its canonical parameter name, so the
fixtures stay small and the state_dict key sets of both models must agree exactly."""
import zlib
from collections import defaultdict

import torch


def _gen(name: str, shape, kind: str) -> torch.Tensor:
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    x = torch.randn(tuple(shape), generator=g)
    if kind == "bn_var":
        return torch.rand(tuple(shape), generator=g) * 0.5 + 0.75
    if kind == "bn_weight":
        return 1.0 + 0.1 * x
    if kind == "norm_weight":
        return 1.0 + 0.05 * x
    if kind == "small":
        return 0.05 * x
    if kind == "embed":
        return x
    # dense / conv weight: variance-preserving for ReLU nets
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    return x * (1.6 / max(fan_in, 1)) ** 0.5


def _kind(name: str, t: torch.Tensor) -> str:
    leaf = name.split(".")[-1]
    if leaf == "running_var":
        return "bn_var"
    if leaf == "running_mean":
        return "small"
    if ".bn" in name or "downsample.1" in name:
        return "bn_weight" if leaf == "weight" else "small"
    if "norm" in name or (name.startswith("input_proj") and ".1." in name):
        return "norm_weight" if leaf == "weight" else "small"
    if leaf == "bias" or leaf == "in_proj_bias":
        return "small"
    if "tgt_embed" in name or "label_enc" in name or "level_embed" in name:
        return "embed"
    return "dense"


@torch.no_grad()
def synth_init_(model: torch.nn.Module) -> None:
    """Overwrite every parameter and buffer of `model` in place.  Tensors that appear under
    several state_dict names (the shared detection heads) are keyed by their smallest name."""
    sd = model.state_dict(keep_vars=True)
    names_of = defaultdict(list)
    for k, v in sd.items():
        names_of[v.data_ptr()].append(k)
    done = set()
    for k, v in sd.items():
        if v.data_ptr() in done:
            continue
        done.add(v.data_ptr())
        canon = min(names_of[v.data_ptr()])
        if canon.endswith("sampling_offsets.bias"):
            # keep the module's analytic ring initialisation, add a little noise
            v.add_(_gen(canon, v.shape, "small"))
        elif "class_embed" in canon and canon.endswith("bias"):
            v.copy_(-4.0 + _gen(canon, v.shape, "small") * 10)
        else:
            v.copy_(_gen(canon, v.shape, _kind(canon, v)).to(v.dtype))


def synth_batch(seed=1, sizes=((256, 320), (240, 300)), num_gt=3, num_classes=9):
    """One source + one target image (different sizes -> non-trivial padding masks) and the
    source image's ground truth (labels 1 .. num_classes - 1: the configs count the background slot)."""
    g = torch.Generator().manual_seed(seed)
    imgs = [torch.randn(3, h, w, generator=g) for h, w in sizes]
    cxcy = torch.rand(num_gt, 2, generator=g) * 0.5 + 0.25
    wh = torch.rand(num_gt, 2, generator=g) * 0.2 + 0.05
    labels = torch.randint(1, num_classes, (num_gt,), generator=g)
    targets = [{"boxes": torch.cat([cxcy, wh], 1), "labels": labels}]
    return imgs, targets
