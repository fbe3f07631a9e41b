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
