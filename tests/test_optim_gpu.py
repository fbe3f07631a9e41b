"""FusedClipAdamW (datr_amd/optim.py, csrc/adamw.hip) against the reference's optimizer tail
    torch.nn.utils.clip_grad_norm_(params, max_norm);  torch.optim.AdamW.step()
(/root/reference/engine.py:99-104, /root/reference/main.py:165): same parameters after several steps,
the `used` flags reproduce AdamW's skipping of parameters without a gradient, state_dicts load both ways."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(dev, seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(256, 256), (9,), (4,), (64, 128, 3, 3), (2048, 512, 1, 1), (1, 1), (33000,), (300, 7)]
    ps = [torch.nn.Parameter(torch.randn(*s, generator=g).to(dev)) for s in shapes]
    ps[3].data = ps[3].data.contiguous(memory_format=torch.channels_last)
    return ps


def _groups(ps):
    return [{"params": ps[:3] + ps[5:]}, {"params": ps[3:5], "lr": 1e-5}]


def _set_grads(ps, step, scale, skip=()):
    g = torch.Generator().manual_seed(100 + step)
    for i, p in enumerate(ps):
        gr = (torch.randn(*p.shape, generator=g) * scale).to(p.device)       # drawn for skipped ones too
        if i in skip:
            p.grad = None
        else:
            p.grad = gr.contiguous(memory_format=torch.channels_last) if p.dim() == 4 and i == 3 else gr


@pytest.mark.parametrize("max_norm,scale", [(0.1, 1.0), (0.1, 1e-6), (0.0, 0.3)])
def test_matches_clip_grad_norm_and_adamw(max_norm, scale):
    from datr_amd.optim import FusedClipAdamW
    dev = torch.device("cuda:0")
    mine, ref = _params(dev, 1), _params(dev, 1)
    om = FusedClipAdamW(_groups(mine), lr=1e-4, weight_decay=1e-4)
    orf = torch.optim.AdamW(_groups(ref), lr=1e-4, weight_decay=1e-4, foreach=False, fused=False)
    for step in range(4):
        _set_grads(mine, step, scale)
        _set_grads(ref, step, scale)
        out = om.clip_and_step(max_norm)
        if max_norm > 0:
            norm = torch.nn.utils.clip_grad_norm_(ref, max_norm)
            torch.testing.assert_close(out[0], norm, rtol=2e-6, atol=0)
        orf.step()
        if step == 1:                                   # a scheduler changes the learning rate
            for o in (om, orf):
                for grp in o.param_groups:
                    grp["lr"] *= 0.1
    for a, b in zip(mine, ref):
        torch.testing.assert_close(a, b, rtol=2e-6, atol=5e-7)          # a few ulp of the largest values
        torch.testing.assert_close(om.state[a]["exp_avg_sq"], orf.state[b]["exp_avg_sq"], rtol=2e-6, atol=1e-20)
        assert float(om.state[a]["step"]) == float(orf.state[b]["step"]) == 4


def test_used_flags_skip_like_missing_gradients():
    """A parameter whose flag is 0 keeps its value, its moments and its step count although it has a (zero)
    gradient tensor -- what AdamW does for .grad = None (the reference's DDP leaves a globally unused
    parameter's .grad None, main.py:156)."""
    from datr_amd.optim import FusedClipAdamW
    dev = torch.device("cuda:0")
    mine, ref = _params(dev, 2), _params(dev, 2)
    om = FusedClipAdamW(_groups(mine), lr=1e-3, weight_decay=1e-2)
    orf = torch.optim.AdamW(_groups(ref), lr=1e-3, weight_decay=1e-2, foreach=False, fused=False)
    om.set_used_order(mine)
    skips = [(), (1, 4), (1,), ()]
    for step, skip in enumerate(skips):
        _set_grads(mine, step, 1.0)
        for i in skip:
            mine[i].grad.zero_()                        # the reducer's zero-filled slice
        _set_grads(ref, step, 1.0, skip=skip)
        used = torch.ones(len(mine), dtype=torch.int32, device=dev)
        for i in skip:
            used[i] = 0
        om.clip_and_step(0.1, used)
        torch.nn.utils.clip_grad_norm_(ref, 0.1)
        orf.step()
    for i, (a, b) in enumerate(zip(mine, ref)):
        # lr 1e-3: a step moves a weight by ~1e-3, each of the four steps may round the last bit differently
        torch.testing.assert_close(a, b, rtol=2e-6, atol=2e-6, msg=lambda m: f"parameter {i}: {m}")
        assert float(om.state[a]["step"]) == float(orf.state[b]["step"]), i


def test_state_dict_round_trip_with_torch_adamw():
    from datr_amd.optim import FusedClipAdamW
    dev = torch.device("cuda:0")
    a, b = _params(dev, 3), _params(dev, 3)
    oa = torch.optim.AdamW(_groups(a), lr=1e-4, weight_decay=1e-4)
    for step in range(2):
        _set_grads(a, step, 1.0)
        oa.step()
    ob = FusedClipAdamW(_groups(b), lr=1e-4, weight_decay=1e-4)
    ob.load_state_dict(copy.deepcopy(oa.state_dict()))
    for pa, pb in zip(a, b):
        pb.data.copy_(pa.data)
    _set_grads(a, 5, 1.0)
    _set_grads(b, 5, 1.0)
    oa.step()
    ob.step()
    for pa, pb in zip(a, b):
        torch.testing.assert_close(pb, pa, rtol=2e-6, atol=5e-7)
    oc = torch.optim.AdamW(_groups(_params(dev, 3)), lr=1e-4, weight_decay=1e-4)
    oc.load_state_dict(ob.state_dict())               # and back
    assert float(list(oc.state.values())[0]["step"]) == 3
