"""CPU: checkpoint round trip in the reference's dict layout, DDP-prefix stripping,
torchvision-style BN entries, --finetune_ignore filtering."""
import torch

from helpers import build_model


def test_roundtrip_and_prefix_and_ignore(tmp_path):
    from datr_amd.checkpoint import load_model_state, save_checkpoint, save_ema_checkpoint
    args, model, _, _ = build_model()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
    p = tmp_path / "checkpoint.pth"
    save_checkpoint(p, model, opt, None, epoch=3, args=args)
    ck = torch.load(p, map_location="cpu", weights_only=False)
    assert set(ck) >= {"model", "optimizer", "epoch", "args"} and len(ck["model"]) == 640

    _, fresh, _, _ = build_model()
    with torch.no_grad():
        for q in fresh.parameters():
            q.add_(1.0)
    # a DDP-saved checkpoint: every key prefixed with "module.", BN carries num_batches_tracked
    ddp = {"module." + k: v for k, v in ck["model"].items()}
    ddp["module.backbone.0.body.bn1.num_batches_tracked"] = torch.tensor(5)
    res = load_model_state(fresh, {"model": ddp})
    assert not res.missing_keys and not res.unexpected_keys
    for (k, a), (_, b) in zip(model.state_dict().items(), fresh.state_dict().items()):
        assert torch.equal(a, b), k

    # --finetune_ignore label_enc class_embed  (different number of classes)
    res = load_model_state(fresh, ck, ignore_keywords=["label_enc", "class_embed"])
    assert all(("label_enc" in k or "class_embed" in k) for k in res.missing_keys) and res.missing_keys

    e = tmp_path / "best_ema_teacher.pth"
    save_ema_checkpoint(e, model, epoch=7)
    assert load_model_state(fresh, str(e)).missing_keys == []


def _key_value(name: str) -> float:
    import zlib
    return 1.0 + (zlib.crc32(name.encode()) % 8192) / 8192.0


def _assert_values_follow_keys(module):
    """Every floating-point entry holds the constant of its key; the shared box / class heads appear
    under 12 keys per tensor (dino.py:155-166) and load_state_dict copies key by key, so such a
    tensor ends up with the value of its LAST alias."""
    sd = module.state_dict()
    last_alias = {}
    for k, v in sd.items():
        last_alias[v.data_ptr()] = k
    for k, v in sd.items():
        if v.is_floating_point():
            want = _key_value(last_alias[v.data_ptr()])
            assert torch.all(v == torch.tensor(want, dtype=v.dtype)), k


def test_reference_written_checkpoint_loads_and_resumes(tmp_path):
    """tests/golden/ref_checkpoint*.pth were written by the REFERENCE's own save path
    (make_golden_checkpoint.py: build_dino + get_param_dict + AdamW + StepLR +
    utils.save_on_master as main.py:401-412; every tensor a constant that encodes its key).
    They load strictly into datr_amd's model -- same key set, same key ORDER, same shapes, every
    value in the tensor of the same name -- with and without the DDP prefix; the optimizer state
    (parameter order inside the two groups!) and the scheduler resume; a checkpoint datr_amd
    saves has the reference's layout."""
    import os
    from datr_amd.checkpoint import load_model_state, resume, save_checkpoint
    from datr_amd.config import get_param_dict
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    ref = torch.load(os.path.join(gold, "ref_checkpoint.pth"), map_location="cpu", weights_only=False)
    args, model, _, _ = build_model()
    assert list(ref["model"].keys()) == list(model.state_dict().keys())
    for k, v in model.state_dict().items():
        assert ref["model"][k].shape == v.shape and ref["model"][k].dtype == v.dtype, k

    for fname in ("ref_checkpoint.pth", "ref_checkpoint_ddp.pth"):
        _, fresh, _, _ = build_model()
        res = load_model_state(fresh, os.path.join(gold, fname))
        assert not res.missing_keys and not res.unexpected_keys
        _assert_values_follow_keys(fresh)

    # --resume: model + EMA copy + optimizer + scheduler + epoch (main.py:226-245)
    _, fresh, _, _ = build_model()
    _, ema, _, _ = build_model()
    opt = torch.optim.AdamW(get_param_dict(args, fresh), lr=args.lr, weight_decay=args.weight_decay)
    sched = torch.optim.lr_scheduler.StepLR(opt, args.lr_drop)
    assert [len(g["params"]) for g in opt.state_dict()["param_groups"]] == \
           [len(g["params"]) for g in ref["optimizer"]["param_groups"]]
    start = resume(os.path.join(gold, "ref_checkpoint_ddp.pth"), fresh, opt, sched, ema_model=ema)
    assert start == ref["epoch"] + 1 == 8
    assert sched.last_epoch == ref["lr_scheduler"]["last_epoch"] and sched.step_size == args.lr_drop
    assert [g["lr"] for g in opt.param_groups] == [args.lr, args.lr_backbone]
    idx = 0
    for g in opt.param_groups:
        for p in g["params"]:                         # same parameter at the same index as the reference
            st = opt.state[p]
            assert st["exp_avg"].shape == p.shape and st["exp_avg_sq"].shape == p.shape, idx
            assert torch.all(st["exp_avg"] == torch.tensor(_key_value(f"opt.{idx}.exp_avg"))), idx
            idx += 1
    assert idx == len(ref["optimizer"]["state"])
    _assert_values_follow_keys(ema)
    assert resume(ref, fresh, opt, sched, eval_only=True) == 0

    # best_ema_teacher.pth layout (main.py:487-507)
    _, teacher, _, _ = build_model()
    assert load_model_state(teacher, os.path.join(gold, "ref_best_ema_teacher.pth")).missing_keys == []

    # what datr_amd saves has the reference's layout
    p = tmp_path / "checkpoint.pth"
    save_checkpoint(p, fresh, opt, sched, epoch=7, args=args, ema_model=ema)
    mine = torch.load(p, map_location="cpu", weights_only=False)
    assert list(mine.keys()) == ["model", "epoch", "args", "optimizer", "lr_scheduler", "ema_model"] or \
        set(mine.keys()) == set(ref.keys())
    assert set(mine.keys()) == set(ref.keys())
    assert list(mine["model"].keys()) == list(ref["model"].keys())
    assert set(mine["lr_scheduler"]) == set(ref["lr_scheduler"])
    assert [sorted(g) for g in mine["optimizer"]["param_groups"]] == \
           [sorted(g) for g in ref["optimizer"]["param_groups"]]
