"""Generate tests/golden/msda_*.npz by running the REFERENCE's own code.

Runs only in the build container (needs /root/reference).  The op outputs come
from the reference's `ms_deform_attn_core_pytorch`
(/root/reference/models/dino/ops/functions/ms_deform_attn_func.py:41-61) and
autograd through it -- the same comparison target the reference's op test uses
(/root/reference/models/dino/ops/test.py:31-86).  Inputs follow that test's
recipe (seed 3, value*0.01, normalised attention weights); extra cases widen the
sampling locations past the borders and place some exactly on pixel centres and
image edges.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ref_shims  # noqa: E402
from oracle.msda_oracle import level_start_index  # noqa: E402

ref_shims.install(neutralise_cuda=False)
from models.dino.ops.functions.ms_deform_attn_func import ms_deform_attn_core_pytorch  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def make_case(name, N, M, D, Lq, P, shapes, dtype, seed=3, loc_lo=0.0, loc_hi=1.0,
              special_locs=False):
    torch.manual_seed(seed)
    shapes_t = torch.as_tensor(shapes, dtype=torch.long)
    L = len(shapes)
    S = int(sum(h * w for h, w in shapes))
    # same draw order as the reference's test
    value = torch.rand(N, S, M, D) * 0.01
    loc = torch.rand(N, Lq, M, L, P, 2) * (loc_hi - loc_lo) + loc_lo
    attn = torch.rand(N, Lq, M, L, P) + 1e-5
    attn /= attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
    if special_locs:
        # exact borders / pixel centres / just outside, on the first few queries
        specials = [0.0, 1.0, -0.05, 1.05, 0.5]
        for lvl, (h, w) in enumerate(shapes):
            specials_l = specials + [0.5 / w, 1.0 - 0.5 / w, 1.5 / w, -0.5 / w, 1.0 + 0.5 / w,
                                     -1.0 / w, 1.0 + 1.0 / w]
            for i, sx in enumerate(specials_l):
                for j, sy in enumerate(specials_l[:4]):
                    q = (i * 4 + j) % Lq
                    loc[:, q, :, lvl, (i + j) % P, 0] = sx
                    loc[:, q, :, lvl, (i + j) % P, 1] = sy * (w / h) if j == 3 else sy
    grad_out = torch.randn(N, Lq, M * D)
    value, loc, attn, grad_out = (t.to(dtype) for t in (value, loc, attn, grad_out))

    v = value.clone().requires_grad_(True)
    s = loc.clone().requires_grad_(True)
    a = attn.clone().requires_grad_(True)
    out = ms_deform_attn_core_pytorch(v, shapes_t, s, a)
    gv, gs, ga = torch.autograd.grad(out, (v, s, a), grad_out)
    np.savez_compressed(
        os.path.join(OUT, f"msda_{name}.npz"),
        value=value.numpy(), shapes=shapes_t.numpy(),
        lsi=level_start_index(shapes_t).numpy(), loc=loc.numpy(), attn=attn.numpy(),
        grad_out=grad_out.numpy(), out=out.detach().numpy(), grad_value=gv.numpy(),
        grad_loc=gs.numpy(), grad_attn=ga.numpy())
    print(f"wrote msda_{name}.npz  out {tuple(out.shape)}")


if __name__ == "__main__":
    small = [(6, 4), (3, 2)]
    # the reference test's own shape (ops/test.py:21-28), double and float
    make_case("ref_test_f64", 1, 2, 2, 2, 2, small, torch.float64)
    make_case("ref_test_f32", 1, 2, 2, 2, 2, small, torch.float32)
    # one case per backward-kernel family the reference dispatches on D (ops/test.py:85),
    # kept tiny; 1025/2048/3096 are exercised property-wise in the tests instead
    make_case("d30_f64", 1, 2, 30, 2, 2, small, torch.float64)
    make_case("d71_f64", 1, 2, 71, 2, 2, small, torch.float64)
    make_case("d64_f32", 1, 2, 64, 3, 2, small, torch.float32)
    # DINO head geometry (M=8, D=32, L=P=4) on a small pyramid, with out-of-range and
    # exactly-on-border sampling locations
    pyr = [(12, 16), (6, 8), (3, 4), (2, 2)]
    make_case("dino_small_f32", 2, 8, 32, 40, 4, pyr, torch.float32, loc_lo=-0.1, loc_hi=1.1,
              special_locs=True)
    make_case("dino_small_f64", 2, 8, 32, 40, 4, pyr, torch.float64, loc_lo=-0.1, loc_hi=1.1,
              special_locs=True)
