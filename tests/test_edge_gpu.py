"""GPU: training steps whose source images have NO ground-truth boxes (one image / every image)
-- Cityscapes frames without instances exist.  The HIP path (device matcher, fused cost / box
losses, cached CDN parts, merged passes) must agree with the same model run on the CPU through
the oracle MSDA (which is pinned to the reference for the ordinary case, test_model_cpu.py)."""
import pytest
import torch

from helpers import build_model, patch_msda_with_oracle

pytestmark = pytest.mark.gpu


def batch(case):
    g = torch.Generator().manual_seed(5)
    imgs = [torch.randn(3, 256, 320, generator=g) for _ in range(4)]        # 2 source + 2 target
    full = {"boxes": torch.tensor([[0.5, 0.5, 0.2, 0.3], [0.3, 0.6, 0.1, 0.1]]),
            "labels": torch.tensor([1, 4])}
    none = {"boxes": torch.zeros(0, 4), "labels": torch.zeros(0, dtype=torch.long)}
    return imgs, {"one_empty": [full, none], "all_empty": [none, none]}[case]


def run(model, criterion, device, case, selection=None, record=None):
    from datr_amd.nested import nested_tensor_from_tensor_list
    imgs, targets = batch(case)
    samples = nested_tensor_from_tensor_list([i.to(device) for i in imgs])
    targets = [{k: v.to(device) for k, v in t.items()} for t in targets]
    model.train()
    criterion.train()
    model.dn_label_noise_ratio = 0.0          # no random draws: CPU and GPU see the same CDN queries
    model.dn_box_noise_scale = 0.0
    own = model.transformer.select_queries
    if selection is not None:
        it = iter(selection)

        def forced(scores):
            idx = next(it).to(scores.device)
            if idx.shape[0] != scores.shape[0]:                # merged source + target decoder pass
                idx = torch.cat([idx, next(it).to(scores.device)], 0)
            return idx
        model.transformer.select_queries = forced
    elif record is not None:
        def rec(scores):
            idx = own(scores)
            record.append(idx.cpu())
            return idx
        model.transformer.select_queries = rec
    out = model(samples, targets)
    loss_dict = criterion(out, targets)
    wd = criterion.weight_dict
    total = sum(loss_dict[k] * wd[k] for k in loss_dict if k in wd)
    model.zero_grad()
    total.backward()
    gsq = sum(float(p.grad.double().pow(2).sum()) for p in model.parameters() if p.grad is not None)
    return float(total), {k: float(v) for k, v in loss_dict.items()}, gsq ** 0.5


@pytest.mark.parametrize("case", ["one_empty", "all_empty"])
def test_step_with_images_without_boxes(case, monkeypatch):
    _, cpu_model, cpu_crit, _ = build_model()
    cpu_model.merge_encoder_passes = False
    selections = []
    with monkeypatch.context() as m:
        patch_msda_with_oracle(m, kind="grid_sample")
        total_c, losses_c, gnorm_c = run(cpu_model, cpu_crit, torch.device("cpu"), case, record=selections)
    _, model, crit, _ = build_model()
    dev = torch.device("cuda:0")
    model.to(dev)
    crit.to(dev)
    total_g, losses_g, gnorm_g = run(model, crit, dev, case, selection=selections)
    assert set(losses_c) == set(losses_g)
    assert all(v == v and abs(v) != float("inf") for v in losses_g.values())
    assert abs(total_g - total_c) <= 2e-3 * abs(total_c) + 1e-4, (total_g, total_c)
    for k, v in losses_c.items():
        if "class_error" in k or "cardinality" in k:
            continue
        assert abs(losses_g[k] - v) <= 5e-3 * abs(v) + 1e-4, (k, losses_g[k], v)
    assert abs(gnorm_g - gnorm_c) <= 2e-2 * gnorm_c + 1e-6, (gnorm_g, gnorm_c)
