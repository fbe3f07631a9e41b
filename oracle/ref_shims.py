"""oracle/ref_shims.py -- TEST INFRASTRUCTURE, build-container only.

Makes the *reference's own Python* (read-only at /root/reference) importable on a
CPU-only box so that golden vectors can be generated from it
(tests/golden/make_golden*.py).  Nothing here is reference code: it only
installs stand-ins for third-party modules that are absent from this image and
neutralises the reference's hard-coded `.cuda()` calls.  /root/reference does
not exist on the GPU box, so nothing under tests/ -m gpu, smoke() or bench.py
may call `install()`.

Stand-ins:
  MultiScaleDeformableAttention  -> forward = the reference's own
        ms_deform_attn_core_pytorch, backward = autograd through it
  torchvision                    -> __version__, _is_tracing, ops.boxes.{box_area,
        nms, batched_nms}, ops.misc.interpolate, models.resnet50 (oracle/resnet_ref.py:
        plain nn.Conv2d / norm / ReLU ResNet-50 v1.5 -- NOT the product's class),
        models._utils.IntermediateLayerGetter, transforms stubs
  timm / cv2 / pycocotools / panopticapi / addict / yapf / termcolor -> empty shells
"""
from __future__ import annotations

import sys
import types
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

REFERENCE_ROOT = "/root/reference"
_installed = False


def _module(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _box_area(boxes):
    return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])


def _nms(boxes, scores, iou_threshold):
    """Greedy NMS on CPU tensors (stand-in for torchvision.ops.nms)."""
    if boxes.numel() == 0:
        return torch.empty(0, dtype=torch.int64)
    order = scores.argsort(descending=True)
    area = _box_area(boxes)
    keep = []
    suppressed = torch.zeros(len(boxes), dtype=torch.bool)
    for idx in order.tolist():
        if suppressed[idx]:
            continue
        keep.append(idx)
        lt = torch.maximum(boxes[idx, :2], boxes[:, :2])
        rb = torch.minimum(boxes[idx, 2:], boxes[:, 2:])
        wh = (rb - lt).clamp(min=0)
        inter = wh[:, 0] * wh[:, 1]
        iou = inter / (area[idx] + area - inter)
        suppressed |= iou > iou_threshold
    return torch.as_tensor(keep, dtype=torch.int64)


def _batched_nms(boxes, scores, idxs, iou_threshold):
    if boxes.numel() == 0:
        return torch.empty(0, dtype=torch.int64)
    offsets = idxs.to(boxes) * (boxes.max() + 1)
    return _nms(boxes + offsets[:, None], scores, iou_threshold)


class _IntermediateLayerGetter(nn.ModuleDict):
    """Runs the children of `model` in order and collects the named outputs."""

    def __init__(self, model, return_layers):
        layers = OrderedDict()
        remaining = dict(return_layers)
        for name, child in model.named_children():
            layers[name] = child
            remaining.pop(name, None)
            if not remaining:
                break
        super().__init__(layers)
        self.return_layers = dict(return_layers)

    def forward(self, x):
        out = OrderedDict()
        for name, child in self.items():
            x = child(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out


def _neutralise_cuda():
    """The reference hard-codes .cuda()/.to('cuda') (dino.py:106-107,790-818;
    dn_components.py:36-113).  On this CPU-only box make them no-ops."""
    def _is_cuda_spec(a):
        return (isinstance(a, str) and a.startswith("cuda")) or \
               (isinstance(a, torch.device) and a.type == "cuda")

    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    _orig_to = torch.Tensor.to

    def _to(self, *args, **kwargs):
        args = tuple("cpu" if _is_cuda_spec(a) else a for a in args)
        if "device" in kwargs and _is_cuda_spec(kwargs["device"]):
            kwargs["device"] = "cpu"
        return _orig_to(self, *args, **kwargs)

    torch.Tensor.to = _to
    _orig_mod_to = nn.Module.to

    def _mod_to(self, *args, **kwargs):
        args = tuple("cpu" if _is_cuda_spec(a) else a for a in args)
        if "device" in kwargs and _is_cuda_spec(kwargs["device"]):
            kwargs["device"] = "cpu"
        return _orig_mod_to(self, *args, **kwargs)

    nn.Module.to = _mod_to


def install(neutralise_cuda: bool = True):
    """Install the stand-ins and put /root/reference on sys.path (idempotent)."""
    global _installed
    if _installed:
        return
    import os
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError("reference tree not present; golden generation only runs in the "
                           "build container")
    sys.dont_write_bytecode = True          # the reference tree is read-only
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    # --- the native op: the reference's own pure-PyTorch core ------------------------
    msda = _module("MultiScaleDeformableAttention")

    def _core():
        from models.dino.ops.functions.ms_deform_attn_func import ms_deform_attn_core_pytorch
        return ms_deform_attn_core_pytorch

    def ms_deform_attn_forward(value, shapes, lsi, loc, attn, im2col_step):
        with torch.no_grad():
            return _core()(value, shapes, loc, attn)

    def ms_deform_attn_backward(value, shapes, lsi, loc, attn, grad_output, im2col_step):
        with torch.enable_grad():
            v = value.detach().requires_grad_(True)
            s = loc.detach().requires_grad_(True)
            a = attn.detach().requires_grad_(True)
            out = _core()(v, shapes, s, a)
            gv, gs, ga = torch.autograd.grad(out, (v, s, a), grad_output)
        return gv, gs, ga

    msda.ms_deform_attn_forward = ms_deform_attn_forward
    msda.ms_deform_attn_backward = ms_deform_attn_backward

    # --- torchvision ------------------------------------------------------------------
    tv = _module("torchvision", __version__="0.15.2", _is_tracing=lambda: False)
    ops = _module("torchvision.ops")
    boxes = _module("torchvision.ops.boxes", box_area=_box_area, nms=_nms,
                    batched_nms=_batched_nms)
    misc = _module("torchvision.ops.misc", interpolate=F.interpolate)
    ops.boxes, ops.misc, ops.nms, ops.batched_nms = boxes, misc, _nms, _batched_nms
    tv.ops = ops

    # torchvision's ResNet-50: oracle/resnet_ref.py, a plain-nn restatement of the public architecture
    # that imports nothing from the product (the product is what the fixtures are compared WITH)
    from . import resnet_ref

    models = _module("torchvision.models", resnet50=resnet_ref.resnet50)
    models_utils = _module("torchvision.models._utils",
                           IntermediateLayerGetter=_IntermediateLayerGetter)
    models._utils = models_utils
    tv.models = models
    tfm = _module("torchvision.transforms")
    tfm_f = _module("torchvision.transforms.functional")
    tfm.functional = tfm_f
    tv.transforms = tfm
    for n in ("Compose", "ToTensor", "Normalize", "RandomCrop", "RandomErasing", "ColorJitter",
              "RandomGrayscale", "RandomApply", "ToPILImage", "GaussianBlur", "RandomResizedCrop"):
        setattr(tfm, n, type(n, (), {"__init__": lambda self, *a, **k: None}))
    # What torchvision.transforms.functional does for PIL images (the only kind the reference's data
    # pipeline, datasets/da_transforms.py, hands it): thin wrappers over Pillow and two tensor ops.
    # Used by tests/golden/make_golden_transforms.py to run the reference's own transform classes.
    def _tv_resize(img, size, interpolation=None):
        from PIL import Image
        return img.resize((int(size[1]), int(size[0])), Image.BILINEAR)

    def _tv_to_tensor(img):
        import numpy as np
        a = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy())
        return a.permute(2, 0, 1).contiguous().to(torch.float32).div(255)

    def _tv_normalize(t, mean, std):
        m = torch.as_tensor(mean, dtype=t.dtype).view(-1, 1, 1)
        s_ = torch.as_tensor(std, dtype=t.dtype).view(-1, 1, 1)
        return t.clone().sub_(m).div_(s_)

    def _tv_hflip(img):
        from PIL import Image
        return img.transpose(Image.FLIP_LEFT_RIGHT)

    tfm_f.resize, tfm_f.to_tensor, tfm_f.normalize, tfm_f.hflip = _tv_resize, _tv_to_tensor, _tv_normalize, _tv_hflip
    tfm_f.crop = lambda img, top, left, height, width: img.crop((left, top, left + width, top + height))

    def _random_crop_params(img, output_size):                 # torchvision.transforms.RandomCrop.get_params
        w, h = img.size
        th, tw = output_size
        if w == tw and h == th:
            return 0, 0, h, w
        i = torch.randint(0, h - th + 1, size=(1,)).item()
        j = torch.randint(0, w - tw + 1, size=(1,)).item()
        return i, j, th, tw
    tfm.RandomCrop.get_params = staticmethod(_random_crop_params)
    dsets = _module("torchvision.datasets")
    dsets.CocoDetection = type("CocoDetection", (torch.utils.data.Dataset,), {})
    dsets.VisionDataset = type("VisionDataset", (torch.utils.data.Dataset,), {})
    tv.datasets = dsets
    _module("torchvision.datasets.vision", VisionDataset=dsets.VisionDataset)

    # --- empty shells for optional dependencies ---------------------------------------
    tl = _module("timm.models.layers", DropPath=nn.Identity,
                 to_2tuple=lambda x: (x, x) if not isinstance(x, tuple) else x,
                 trunc_normal_=lambda t, std=0.02, **k: nn.init.trunc_normal_(t, std=std))
    tm = _module("timm.models", layers=tl)
    _module("timm", models=tm)
    _module("cv2")
    pm = _module("pycocotools.mask")
    pc = _module("pycocotools.coco", COCO=type("COCO", (), {}))
    pe = _module("pycocotools.cocoeval", COCOeval=type("COCOeval", (), {}))
    _module("pycocotools", mask=pm, coco=pc, cocoeval=pe)
    pu = _module("panopticapi.utils", id2rgb=None, rgb2id=None)
    pv = _module("panopticapi.evaluation", pq_compute=None)
    _module("panopticapi", utils=pu, evaluation=pv)
    _module("addict", Dict=dict)
    ya = _module("yapf.yapflib.yapf_api", FormatCode=lambda s, **k: (s, False))
    yl = _module("yapf.yapflib", yapf_api=ya)
    _module("yapf", yapflib=yl)
    _module("termcolor", colored=lambda s, *a, **k: s)
    import PIL.Image  # noqa: F401  (datasets/ imports PIL lazily in a way that needs this)

    if neutralise_cuda:
        _neutralise_cuda()
    _installed = True


def load_config(relpath: str = "config/DA/Cityscapes2FoggyCityscapes/DINO_4scale_C2F.py",
                **overrides):
    """exec() a reference config file (they are flat python assignments) and return
    an argparse.Namespace with main.py's CLI defaults that build_dino reads."""
    import argparse
    import os
    ns: dict = {}
    with open(os.path.join(REFERENCE_ROOT, relpath)) as f:
        exec(compile(f.read(), relpath, "exec"), ns)
    cfg = {k: v for k, v in ns.items() if not k.startswith("__") and k != "_base_"}
    # scripts/DINO_train.sh overrides
    cfg.update(embed_init_tgt=True, dn_box_noise_scale=1.0, use_ema=False)
    cfg.update(device="cpu", frozen_weights=None, dataset_file="city2foggy", amp=False,
               debug=False, onecyclelr=False)
    cfg.update(overrides)
    return argparse.Namespace(**cfg)
