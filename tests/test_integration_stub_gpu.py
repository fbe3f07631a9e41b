"""GPU: the binding INTEGRATION.md tells a DATR maintainer to add
(`models/dino/ops/MultiScaleDeformableAttention.py`, the stand-in for the pybind11 module of
/root/reference/models/dino/ops/src/vision.cpp:13-16) is EXECUTED as written -- the code block is
taken out of the document, pointed at the built library, imported as a module -- and its two
functions are compared with the oracle, and driven through an autograd Function shaped like the
reference's MSDeformAttnFunction (ops/functions/ms_deform_attn_func.py:21-38)."""
import os
import re
import types

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_stub():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    code = next(b for b in blocks if "MultiScaleDeformableAttention.py" in b.splitlines()[0])
    assert "/path/to/datr_amd/lib/libdatr_hip.so" in code
    code = code.replace("/path/to/datr_amd/lib/libdatr_hip.so", os.path.join(ROOT, "datr_amd", "lib", "libdatr_hip.so"))
    mod = types.ModuleType("MultiScaleDeformableAttention")
    exec(compile(code, "INTEGRATION.md", "exec"), mod.__dict__)
    return mod


def test_documented_binding_runs_and_matches_the_oracle():
    from oracle import msda_oracle as O
    MSDA = _load_stub()
    dev = torch.device("cuda:0")
    shapes_l = [(20, 27), (10, 14), (5, 7), (3, 4)]
    S = sum(h * w for h, w in shapes_l)
    value, shapes, lsi, loc, attn = O.random_inputs(2, 50, 8, 32, shapes_l, 4, seed=3, loc_range=(-0.1, 1.1))
    assert value.shape[1] == S
    go = torch.randn(2, 50, 256, generator=torch.Generator().manual_seed(1))
    want = O.msda_forward(value, shapes, lsi, loc, attn)
    want_g = O.msda_backward(value, shapes, lsi, loc, attn, go)
    d = [t.to(dev) for t in (value, shapes, lsi, loc, attn)]
    out = MSDA.ms_deform_attn_forward(*d, 64)
    torch.testing.assert_close(out.cpu(), want, rtol=1e-5, atol=1e-6)
    grads = MSDA.ms_deform_attn_backward(*d, go.to(dev), 64)
    for g, w in zip(grads, want_g):
        torch.testing.assert_close(g.cpu(), w, rtol=1e-4, atol=1e-5)

    # the reference's autograd Function over the binding (ms_deform_attn_func.py:21-38)
    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, value, shapes, lsi, loc, attn, step):
            ctx.step = step
            ctx.save_for_backward(value, shapes, lsi, loc, attn)
            return MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, attn, step)

        @staticmethod
        def backward(ctx, grad_output):
            gv, gl, ga = MSDA.ms_deform_attn_backward(*ctx.saved_tensors, grad_output, ctx.step)
            return gv, None, None, gl, ga, None
    v, l_, a = (d[i].clone().requires_grad_(True) for i in (0, 3, 4))
    Fn.apply(v, d[1], d[2], l_, a, 64).backward(go.to(dev))
    for g, w in zip((v.grad, l_.grad, a.grad), want_g):
        torch.testing.assert_close(g.cpu(), w, rtol=1e-4, atol=1e-5)
    with pytest.raises(AssertionError):                      # CPU tensors are refused, as in the reference
        MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, attn, 64)
