"""Generate tests/golden/ref_checkpoint.pth: a checkpoint written by the REFERENCE's own save path
(build container only) -- SURVEY.md section 8 row f2.

Builds the reference's DINO (`build_dino`, /root/reference/models/dino/dino.py:999) with the
Cityscapes->Foggy config, its parameter groups / AdamW / StepLR as /root/reference/main.py:161-212
does, takes one optimizer step (so the optimizer state exists), and saves
`{'model', 'optimizer', 'lr_scheduler', 'epoch', 'args', 'ema_model'}` with
`utils.save_on_master` exactly as main.py:401-412 -- once plain, once with the `module.` prefix a
DistributedDataParallel wrapper puts on every key (what `clean_state_dict`, util/misc.py:593-599,
strips at load time).

The file must stay small, so every tensor is replaced -- right before saving -- by a ONE-element
tensor expanded to the original shape (torch.save stores the 4-byte storage, shape and zero
strides): the value encodes the key (CRC32 of the name mapped into [1, 2)), so a test can check that
every key lands in the tensor of the same name.  Key set, key order, shapes, dtypes, optimizer
param-group structure, scheduler state and the pickled `args` namespace are the reference's own.

    python tests/golden/make_golden_checkpoint.py
"""
import copy
import os
import sys
import tempfile
import zlib

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import ref_shims  # noqa: E402

ref_shims.install()
from models.dino.dino import build_dino  # noqa: E402  (the reference's)
from util.get_param_dicts import get_param_dict  # noqa: E402
import util.misc as utils  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def key_value(name: str) -> float:
    return 1.0 + (zlib.crc32(name.encode()) % 8192) / 8192.0


def constant_like(name: str, t: torch.Tensor) -> torch.Tensor:
    if not torch.is_tensor(t) or t.dim() == 0:
        return t
    v = key_value(name) if t.is_floating_point() else zlib.crc32(name.encode()) % 97
    return torch.full((1,) * t.dim(), v, dtype=t.dtype).expand(t.shape)


def main():
    tmp = tempfile.mkdtemp()
    args = ref_shims.load_config(output_dir=tmp, param_dict_type="default")
    torch.manual_seed(0)
    model, _, _ = build_dino(args)
    optimizer = torch.optim.AdamW(get_param_dict(args, model), lr=args.lr, weight_decay=args.weight_decay)
    lr_scheduler = torch.optim.lr_scheduler.StepLR(optimizer, args.lr_drop)          # main.py:212
    for p in model.parameters():
        if p.requires_grad:
            p.grad = torch.zeros_like(p)
    optimizer.step()
    lr_scheduler.step()
    ema = copy.deepcopy(model)                                                       # ModelEma.module

    def model_sd(m, prefix=""):
        return type(m.state_dict())((prefix + k, constant_like(k, v)) for k, v in m.state_dict().items())

    opt_sd = optimizer.state_dict()
    for idx, st in opt_sd["state"].items():
        for name in list(st):
            st[name] = constant_like(f"opt.{idx}.{name}", st[name])
    epoch = 7
    for fname, prefix in (("ref_checkpoint.pth", ""), ("ref_checkpoint_ddp.pth", "module.")):
        weights = {                                                                  # main.py:401-412
            "model": model_sd(model, prefix),
            "optimizer": opt_sd,
            "lr_scheduler": lr_scheduler.state_dict(),
            "epoch": epoch,
            "args": args,
        }
        weights.update({"ema_model": model_sd(ema, prefix)})
        utils.save_on_master(weights, os.path.join(OUT, fname))
        print(fname, os.path.getsize(os.path.join(OUT, fname)), "bytes,", len(weights["model"]), "keys")
    utils.save_on_master({"ema_model": model_sd(ema), "epoch": epoch},               # main.py:487-507
                         os.path.join(OUT, "ref_best_ema_teacher.pth"))


if __name__ == "__main__":
    main()
