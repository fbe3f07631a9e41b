"""Two size points of BASELINE.json on the device:
  * the real Cityscapes -> Foggy Cityscapes training size: 1 source + 1 target image of 1024 x 2048
    (config/DA/Cityscapes2FoggyCityscapes/coco_transformer_C2F.py:1-7, S = 43 520 encoder tokens) through
    `engine.train_one_epoch` -- the pyramid-region MSDA kernels cover the geometry, the losses are finite,
    memory high-water mark and ms/step are recorded;
  * BASELINE configs[0]'s shape: ONE source image 640 x 640, DA branch off (S = 8 500) -- forward +
    SetCriterion on the device against the same weights on the host cores with oracle/msda_ref.c as the
    MSDA op, as tests/test_model_gpu.py does at 800 x 1333."""
import json
import os
import sys
import time

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
from helpers import build_model  # noqa: E402

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _record(name, payload):
    """Measurements of these tests go to gpurun_out/ (scratch, merged back by gpurun); the builder copies
    the ones to be judged into profiles/."""
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, name), "w") as f:
            json.dump(payload, f, indent=1)
    except OSError:
        pass


def test_training_step_at_the_real_c2f_size():
    from datr_amd import msda
    from datr_amd.training import build_training, run_steps, synthetic_batch
    dev = torch.device("cuda:0")
    H, W = 1024, 2048
    shapes = torch.tensor([[(H + s - 1) // s, (W + s - 1) // s] for s in (8, 16, 32, 64)], dtype=torch.int64)
    S = int(shapes.prod(1).sum())
    assert S == 43520
    lsi = torch.cat([shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]])
    plan = msda.pyramid_plan(shapes, lsi, 2, 8, 32, 4)
    assert plan["forward"] and plan["backward"], plan       # no DATR_EUNSUPPORTED at this geometry

    torch.zeros(1, device=dev)                  # (initialises the device context)
    torch.cuda.reset_peak_memory_stats(dev)
    state = build_training(device=dev)
    pool = [synthetic_batch(1, H, W, 10, dev, seed=1 + 1000 * i) for i in range(2)]
    calls = {"fwd": 0, "bwd": 0}
    orig_f, orig_b = msda.ms_deform_attn_forward, msda.ms_deform_attn_backward

    def fwd(value, *a, **kw):
        calls["fwd"] += value.shape[1] == S and a[2].shape[1] == S
        return orig_f(value, *a, **kw)

    def bwd(value, *a, **kw):
        calls["bwd"] += value.shape[1] == S and a[2].shape[1] == S
        return orig_b(value, *a, **kw)
    orig_q = msda.ms_deform_attn_backward_query_grad

    def bwd_q(value, *a, **kw):                     # the module's entry to the same two kernels
        done = orig_q(value, *a, **kw)
        calls["bwd"] += done is not None and value.shape[1] == S
        return done
    msda.ms_deform_attn_forward, msda.ms_deform_attn_backward = fwd, bwd
    msda.ms_deform_attn_backward_query_grad = bwd_q
    try:
        stats = run_steps(state, [pool[i % 2] for i in range(3)])       # warm-up; raises on any native error code
        torch.cuda.synchronize()
        n = 4
        t0 = time.perf_counter()
        stats = run_steps(state, [pool[i % 2] for i in range(n)])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
    finally:
        msda.ms_deform_attn_forward, msda.ms_deform_attn_backward = orig_f, orig_b
        msda.ms_deform_attn_backward_query_grad = orig_q
    assert calls["fwd"] == 6 * 7 and calls["bwd"] == 6 * 7, calls    # six encoder layers, merged N = 2 pass
    assert np.isfinite(stats["loss"]) and stats["loss"] > 0
    for k, v in stats.items():
        if isinstance(v, float):
            assert np.isfinite(v), k
    peak = torch.cuda.max_memory_allocated(dev)
    assert peak < 200 * 2 ** 30
    _record("c2f_size_step.json", {
        "workload": "1 source + 1 target image 1024x2048 (C2F training size), S = 43520, engine.train_one_epoch",
        "ms_per_step": round(ms, 2), "images_per_s": round(2e3 / ms, 2), "steps_timed": n,
        "peak_allocated_GiB": round(peak / 2 ** 30, 2), "loss": stats["loss"],
        "pyramid_plan": {k: (list(v) if isinstance(v, tuple) else v) for k, v in plan.items()}})
    print(f"C2F-size step: {ms:.1f} ms, peak {peak / 2 ** 30:.1f} GiB, loss {stats['loss']:.3f}")


def test_config1_shape_source_only_matches_host_run_with_oracle_msda(monkeypatch):
    import copy
    from datr_amd.nested import nested_tensor_from_tensor_list
    from helpers import patch_msda_with_oracle
    dev = torch.device("cuda:0")
    args, model, criterion, _ = build_model("cuda:0")
    model.domain_adaptation = False
    g = torch.Generator().manual_seed(6)
    imgs = [torch.randn(3, 640, 640, generator=g)]
    n_gt = 5
    cxcy = torch.rand(n_gt, 2, generator=g) * 0.6 + 0.2
    wh = torch.rand(n_gt, 2, generator=g) * 0.2 + 0.05
    targets = [{"boxes": torch.cat([cxcy, wh], 1), "labels": torch.randint(1, 9, (n_gt,), generator=g)}]
    known = 2 * (200 // (2 * n_gt)) * n_gt
    p = torch.rand(known, generator=g)
    noise = {"label_p": p, "new_label": torch.randint(0, 9, (int((p < 0.25).sum()),), generator=g),
             "rand_sign": torch.randint(0, 2, (known, 4), generator=g).float() * 2 - 1,
             "rand_part": torch.rand(known, 4, generator=g)}
    host_model = copy.deepcopy(model).cpu()
    host_model.domain_adaptation = False

    def run(m, device, selection=None):
        m.train()
        criterion.train()
        m.dn_noise_override = {k: v.clone() for k, v in noise.items()}
        picked = []
        own = m.transformer.select_queries

        def select(scores):
            idx = own(scores) if selection is None else selection.pop(0).to(scores.device)
            picked.append(idx.cpu())
            return idx
        m.transformer.select_queries = select
        samples = nested_tensor_from_tensor_list([i.to(device) for i in imgs])
        assert samples.tensors.shape[-2:] == (640, 640)
        tg = [{k: v.to(device) for k, v in t_.items()} for t_ in targets]
        out = m(samples, tg)
        losses = criterion(out, tg)
        out = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}
        return out, {k: float(v.detach()) for k, v in losses.items()}, picked

    out_d, loss_d, picked = run(model, dev)
    assert all(np.isfinite(v) for v in loss_d.values())
    assert out_d["pred_logits"].shape[:2] == (1, 900)
    patch_msda_with_oracle(monkeypatch, kind="c")
    out_h, loss_h, _ = run(host_model, torch.device("cpu"), selection=[p_.clone() for p_ in picked])
    assert list(loss_d) == list(loss_h)
    for k in loss_d:
        if "class_error" in k or "cardinality" in k:
            continue
        assert abs(loss_d[k] - loss_h[k]) <= 1e-3 * abs(loss_h[k]) + 1e-4, (k, loss_d[k], loss_h[k])
    for key in ("pred_logits", "pred_boxes"):
        torch.testing.assert_close(out_d[key].float().cpu(), out_h[key], rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("sizes,nhwc", [(((352, 597), (333, 640)), True), (((415, 333), (415, 333)), True),
                                        (((301, 517), (352, 480)), False)],
                         ids=["padded-odd-nhwc", "portrait-unpadded-nhwc", "padded-odd-nchw"])
def test_odd_sized_da_step_matches_host_run_with_oracle_msda(sizes, nhwc, monkeypatch):
    """A multi-scale training run hands the model image sizes that are multiples of nothing (datasets/coco transforms:
    random shorter side, aspect kept) and pairs of different sizes (padded batch, masks, valid ratios).  The whole DA
    step's forward + criterion on the device -- own convolution kernels with their edge tiles, NHWC GroupNorm, pyramid
    MSDA plans for that geometry, masks in the encoder -- against the same weights on the host cores with the C oracle
    as the MSDA op, the device's top-900 selection substituted: losses to 1e-3 relative, logits / boxes to 1e-3."""
    import copy
    from datr_amd.nested import nested_tensor_from_tensor_list
    from helpers import patch_msda_with_oracle
    dev = torch.device("cuda:0")
    args, model, criterion, _ = build_model("cuda:0")
    host_model = copy.deepcopy(model).cpu()
    if nhwc:
        model.backbone.to(memory_format=torch.channels_last)
    g = torch.Generator().manual_seed(sizes[0][0] + sizes[1][1])
    imgs = [torch.randn(3, h, w, generator=g) for h, w in sizes]
    n_gt = 4
    cxcy = torch.rand(n_gt, 2, generator=g) * 0.6 + 0.2
    wh = torch.rand(n_gt, 2, generator=g) * 0.2 + 0.05
    targets = [{"boxes": torch.cat([cxcy, wh], 1), "labels": torch.randint(1, 9, (n_gt,), generator=g)}]
    known = 2 * (200 // (2 * n_gt)) * n_gt
    p = torch.rand(known, generator=g)
    noise = {"label_p": p, "new_label": torch.randint(0, 9, (int((p < 0.25).sum()),), generator=g),
             "rand_sign": torch.randint(0, 2, (known, 4), generator=g).float() * 2 - 1,
             "rand_part": torch.rand(known, 4, generator=g)}

    def run(m, device, selection=None):
        m.train()
        criterion.train()
        m.dn_noise_override = {k: v.clone() for k, v in noise.items()}
        m.global_proto.zero_()
        m.Amount.zero_()
        picked = []
        own = m.transformer.select_queries

        def select(scores):
            idx = own(scores) if selection is None else selection.pop(0).to(scores.device)
            picked.append(idx.cpu())
            return idx
        m.transformer.select_queries = select
        samples = nested_tensor_from_tensor_list([i.to(device) for i in imgs])
        if nhwc and device.type == "cuda":
            samples.tensors = samples.tensors.contiguous(memory_format=torch.channels_last)
        tg = [{k: v.to(device) for k, v in t_.items()} for t_ in targets]
        out = m(samples, tg)
        losses = criterion(out, tg)
        return out, {k: float(v.detach()) for k, v in losses.items()}, picked

    out_d, loss_d, picked = run(model, dev)
    assert all(np.isfinite(v) for v in loss_d.values()) and len(loss_d) == 82
    patch_msda_with_oracle(monkeypatch, kind="c")
    out_h, loss_h, _ = run(host_model, torch.device("cpu"), selection=[p_.clone() for p_ in picked])
    assert list(loss_d) == list(loss_h)
    for k in loss_d:
        if "class_error" in k or "cardinality" in k:
            continue
        assert abs(loss_d[k] - loss_h[k]) <= 2e-3 * abs(loss_h[k]) + 2e-4, (k, loss_d[k], loss_h[k])
    for key in ("pred_logits", "pred_boxes"):
        torch.testing.assert_close(out_d[key].detach().float().cpu(), out_h[key].detach(), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(out_d["da_output"]["backbone_DA"].detach().float().cpu(),
                               out_h["da_output"]["backbone_DA"].detach(), rtol=1e-3, atol=1e-3)
