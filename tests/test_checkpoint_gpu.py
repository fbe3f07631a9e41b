"""GPU: checkpoint I/O against the DEVICE model (/root/reference/main.py:226-245,
:401-412, util/misc.py:593-599).

  * the checkpoints the REFERENCE's own save path wrote (tests/golden/ref_checkpoint*.pth: every
    tensor a constant that encodes its key) load strictly into the model on cuda:0, in NCHW and in
    the NHWC (channels_last) backbone layout bench.py trains in, with and without the DDP prefix;
  * a device model with real weights: eval forward -> save in the reference's layout -> load into
    a second device model THAT HAS ALREADY RUN (its shape- / version-keyed caches -- folded frozen-BN
    weights, BN affines, Winograd filter transforms -- are populated with other weights) -> the same
    eval forward (to the run-to-run noise of the vendor kernels); also through the `module.`-prefixed
    form and the EMA layout.
"""
import os

import pytest
import torch

from helpers import build_model
from test_checkpoint_cpu import _assert_values_follow_keys

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("nhwc", [False, True])
def test_reference_written_checkpoints_load_into_the_device_model(nhwc):
    from datr_amd.checkpoint import load_model_state, resume
    from datr_amd.config import get_param_dict
    for fname in ("ref_checkpoint.pth", "ref_checkpoint_ddp.pth", "ref_best_ema_teacher.pth"):
        args, model, _, _ = build_model("cuda:0")
        if nhwc:
            model.backbone.to(memory_format=torch.channels_last)
        res = load_model_state(model, os.path.join(GOLD, fname))
        assert not res.missing_keys and not res.unexpected_keys
        assert all(v.is_cuda for v in model.state_dict().values())
        _assert_values_follow_keys(model)
        if nhwc:        # loading must not silently change the layout the kernels are routed by
            w = model.backbone[0].body.layer2[0].conv2.weight
            assert w.is_contiguous(memory_format=torch.channels_last)
    # --resume on the device: optimizer state lands on the parameters' device
    args, model, _, _ = build_model("cuda:0")
    opt = torch.optim.AdamW(get_param_dict(args, model), lr=args.lr, weight_decay=args.weight_decay,
                            fused=True)
    sched = torch.optim.lr_scheduler.StepLR(opt, args.lr_drop)
    assert resume(os.path.join(GOLD, "ref_checkpoint_ddp.pth"), model, opt, sched) == 8
    st = opt.state[opt.param_groups[0]["params"][0]]
    assert st["exp_avg"].is_cuda and st["exp_avg_sq"].is_cuda


def _eval_forward(model, imgs, nhwc, selection=None, record=None):
    from datr_amd.nested import nested_tensor_from_tensor_list
    model.eval()
    own = type(model.transformer).select_queries.__get__(model.transformer)
    if selection is not None:
        it = iter(selection)
        model.transformer.select_queries = lambda scores: next(it)
    elif record is not None:
        model.transformer.select_queries = lambda scores: (record.append(own(scores)), record[-1])[1]
    with torch.no_grad():
        samples = nested_tensor_from_tensor_list(imgs)
        if nhwc:
            samples.tensors = samples.tensors.contiguous(memory_format=torch.channels_last)
        out = model(samples)
    model.transformer.__dict__.pop("select_queries", None)
    return out


@pytest.mark.parametrize("nhwc", [False, True])
def test_save_load_roundtrip_reproduces_the_eval_forward(tmp_path, nhwc):
    import synth
    from datr_amd.checkpoint import load_model_state, save_checkpoint, save_ema_checkpoint
    dev = torch.device("cuda:0")
    imgs = [i.to(dev) for i in synth.synth_batch()[0]]
    args, model, _, _ = build_model("cuda:0")
    if nhwc:
        model.backbone.to(memory_format=torch.channels_last)
    picked = []
    want = _eval_forward(model, imgs, nhwc, record=picked)
    assert bool(torch.isfinite(want["pred_logits"]).all())

    p = tmp_path / "checkpoint.pth"
    save_checkpoint(p, model, None, None, epoch=1, args=args, ema_model=model)
    ck = torch.load(p, map_location="cpu", weights_only=False)
    assert len(ck["model"]) == 640 and set(ck) >= {"model", "ema_model", "epoch", "args"}
    ddp = tmp_path / "checkpoint_ddp.pth"
    torch.save({"model": {"module." + k: v for k, v in ck["model"].items()}, "epoch": 1}, ddp)
    ema = tmp_path / "best_ema_teacher.pth"
    save_ema_checkpoint(ema, model, epoch=1)

    for path, key in ((p, "model"), (ddp, None), (ema, None), (p, "ema_model")):
        # a model with OTHER weights that has already run: every cache is populated and stale
        _, other, _, _ = build_model("cuda:0")
        if nhwc:
            other.backbone.to(memory_format=torch.channels_last)
        with torch.no_grad():
            for q in other.parameters():
                q.mul_(1.01)
            for b in other.buffers():
                if b.is_floating_point():
                    b.mul_(1.02)
        stale = _eval_forward(other, imgs, nhwc, selection=[s.clone() for s in picked])
        assert float((stale["pred_logits"] - want["pred_logits"]).abs().max()) > 1e-2
        res = load_model_state(other, str(path), key=key)
        assert not res.missing_keys and not res.unexpected_keys
        got = _eval_forward(other, imgs, nhwc, selection=[s.clone() for s in picked])
        for k in ("pred_logits", "pred_boxes"):
            # run-to-run the vendor GEMM / convolution kernels agree to ~6e-5 (test_model_gpu.py)
            torch.testing.assert_close(got[k], want[k], rtol=5e-4, atol=5e-4)
