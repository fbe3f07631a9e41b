"""GPU parity tests: the HIP kernels, called through the C ABI (datr_amd.msda ->
libdatr_hip.so), against (a) the golden vectors captured from the reference and (b) the C
oracle on seeded inputs, plus size-independent properties at BASELINE's full sizes.

Tolerances: BASELINE.json north_star asks for deformable-attn outputs within 1e-3 fp32; the
reference's own float check is rtol 1e-2 / atol 1e-3 (ops/test.py:56).  We hold fp32 to
rtol 1e-4 / atol 1e-6 (inputs are O(1e-2)) and fp64 to default allclose."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN_CASES = ["ref_test_f64", "ref_test_f32", "d30_f64", "d71_f64", "d64_f32",
                "dino_small_f32", "dino_small_f64"]

# (H, W) pyramid of BASELINE config 2/3
FULL_SHAPES = [(100, 167), (50, 84), (25, 42), (13, 21)]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def M():
    from datr_amd import msda
    return msda


@pytest.fixture(scope="module")
def O():
    from oracle import msda_oracle
    return msda_oracle


def tol(dtype, loose=1.0):
    if dtype == torch.float64:
        return dict(rtol=1e-5, atol=1e-8)
    return dict(rtol=1e-4 * loose, atol=1e-6 * loose)


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, f"msda_{name}.npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def off_grid(loc, shapes, eps=1e-3):
    """Mask [N,Lq,M,L,P,2] of samples whose pixel coordinates are at least `eps` away from an
    integer: d/d(loc) is discontinuous on grid lines, and two float pipelines (fp32 vs fp64,
    fused vs unfused multiply-add) may floor to different sides there."""
    HW = shapes.to(torch.float64).to(loc.device)
    h_im = loc[..., 1].double() * HW[:, 0].view(1, 1, 1, -1, 1) - 0.5
    w_im = loc[..., 0].double() * HW[:, 1].view(1, 1, 1, -1, 1) - 0.5
    near = ((h_im - h_im.round()).abs() < eps) | ((w_im - w_im.round()).abs() < eps)
    return (~near).unsqueeze(-1).expand(*loc.shape)


def run_hip(M, dev, value, shapes, lsi, loc, attn, grad_out=None):
    v, s, a = value.to(dev), loc.to(dev), attn.to(dev)
    sh, ls = shapes.to(dev), lsi.to(dev)
    out = M.ms_deform_attn_forward(v, sh, ls, s, a, 64)
    if grad_out is None:
        return out.cpu()
    gv, gl, ga = M.ms_deform_attn_backward(v, sh, ls, s, a, grad_out.to(dev).contiguous(), 64)
    return out.cpu(), gv.cpu(), gl.cpu(), ga.cpu()


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_golden_vectors(golden_dir, M, dev, name):
    g = load(golden_dir, name)
    out, gv, gl, ga = run_hip(M, dev, g["value"], g["shapes"], g["lsi"], g["loc"], g["attn"],
                              g["grad_out"])
    t = tol(out.dtype)
    torch.testing.assert_close(out, g["out"], **t)
    torch.testing.assert_close(gv, g["grad_value"], **t)
    torch.testing.assert_close(ga, g["grad_attn"], **t)
    # samples with a pixel coordinate of exactly -1: native-kernel semantics (zero gradient),
    # see tests/test_oracle_msda.py
    H = g["shapes"][:, 0].to(gl.dtype).view(1, 1, 1, -1, 1)
    W = g["shapes"][:, 1].to(gl.dtype).view(1, 1, 1, -1, 1)
    edge = ((g["loc"][..., 0] * W - 0.5) == -1) | ((g["loc"][..., 1] * H - 0.5) == -1)
    keep = ~edge.unsqueeze(-1).expand_as(gl)
    torch.testing.assert_close(gl[keep], g["grad_loc"][keep], **tol(out.dtype, loose=10))
    assert torch.count_nonzero(gl[~keep]) == 0


CASES = [
    # N, Lq, M, D, shapes, P, loc_range, dtype
    (2, 77, 8, 32, [(20, 27), (10, 14), (5, 7), (3, 4)], 4, (-0.2, 1.2), torch.float32),   # fast, K=16, ragged Lq
    (1, 32, 8, 32, [(9, 9), (5, 5), (3, 3), (2, 2)], 4, (0.0, 1.0), torch.float32),
    (3, 45, 4, 16, [(8, 6), (4, 3)], 3, (-0.1, 1.1), torch.float32),                       # LPR=4, K=6
    (2, 19, 2, 64, [(7, 5), (4, 4), (2, 3)], 2, (-0.1, 1.1), torch.float32),               # LPR=16, K=6
    (1, 130, 8, 32, [(16, 16)], 1, (0.0, 1.0), torch.float32),                             # K=1
    (2, 33, 8, 32, [(12, 10), (6, 5), (3, 3), (2, 2), (1, 1)], 5, (-0.1, 1.1), torch.float32),  # K=25
    (1, 9, 3, 30, [(6, 4), (3, 2)], 2, (0.0, 1.0), torch.float32),                         # generic
    (1, 5, 2, 71, [(6, 4), (3, 2)], 2, (0.0, 1.0), torch.float64),
    (1, 3, 2, 1025, [(6, 4), (3, 2)], 2, (0.0, 1.0), torch.float64),
    (1, 2, 1, 2048, [(6, 4), (3, 2)], 2, (0.0, 1.0), torch.float32),
    (2, 50, 8, 32, [(20, 27), (10, 14), (5, 7), (3, 4)], 4, (-0.2, 1.2), torch.float64),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"N{c[0]}q{c[1]}M{c[2]}D{c[3]}L{len(c[4])}P{c[5]}{str(c[7])[-2:]}")
def test_matches_c_oracle(M, O, dev, case):
    N, Lq, Mh, D, shapes, P, rng, dtype = case
    value, sh, lsi, loc, attn = O.random_inputs(N, Lq, Mh, D, shapes, P, seed=7, dtype=dtype,
                                                loc_range=rng)
    go = torch.randn(N, Lq, Mh * D, generator=torch.Generator().manual_seed(2)).to(dtype)
    out, gv, gl, ga = run_hip(M, dev, value, sh, lsi, loc, attn, go)
    t = tol(dtype)
    torch.testing.assert_close(out, O.msda_forward(value, sh, lsi, loc, attn), **t)
    rv, rl, ra = O.msda_backward(value, sh, lsi, loc, attn, go)
    torch.testing.assert_close(gv, rv, **tol(dtype, 10))
    torch.testing.assert_close(ga, ra, **tol(dtype, 10))
    keep = off_grid(loc, sh, eps=1e-4)
    torch.testing.assert_close(gl[keep], rl[keep], **tol(dtype, 100))


def pyramid_locs(shapes, N, Mh, P, spread, seed):
    """Encoder-style sampling locations: every pyramid pixel is a query whose reference point
    is its own centre; offsets ~ N(0, spread px) in the target level's pixels."""
    g = torch.Generator().manual_seed(seed)
    refs = []
    for h, w in shapes:
        ys, xs = torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij")
        refs.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
    ref = torch.cat(refs, 0)
    S, L = ref.shape[0], len(shapes)
    wh = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32).view(1, 1, 1, L, 1, 2)
    off = torch.randn(N, S, Mh, L, P, 2, generator=g) * spread
    return (ref.view(1, S, 1, 1, 1, 2) + off / wh).contiguous()


@pytest.mark.parametrize("shapes,spread", [
    ([(20, 27), (10, 14), (5, 7), (3, 4)], 1.5),
    ([(20, 27), (10, 14), (5, 7), (3, 4)], 12.0),      # mostly out-of-window -> global fetches
    ([(40, 61), (20, 31), (10, 16), (5, 8)], 4.0),     # several regions per axis, ring-sized offsets
    ([(33, 50), (17, 25), (9, 13), (5, 7)], 6.0),      # mixed: part in the windows, part outside
    ([(33, 50), (17, 25)], 3.0),                       # 2 levels: not covered -> row kernel
])
def test_pyramid_region_forward_matches_oracle(M, O, dev, shapes, spread, monkeypatch):
    """Encoder calls take the pyramid-region kernels (csrc/msda_fwd_pyr2.hip: every level's window in LDS,
    out-of-window samples from global memory; csrc/msda_fwd_pyr.hip where no window plan exists): == oracle,
    and == the row kernel up to the order of the fp32 sums."""
    assert M.PYR_FORWARD
    N, Mh, D, P = 2, 8, 32, 4
    value, sh, lsi, _, _ = O.random_inputs(N, 1, Mh, D, shapes, P, seed=13)
    S = value.shape[1]
    loc = pyramid_locs(shapes, N, Mh, P, spread, seed=7)
    g = torch.Generator().manual_seed(8)
    attn = torch.softmax(torch.randn(N, S, Mh, len(shapes) * P, generator=g), -1).view(N, S, Mh, len(shapes), P)
    out = run_hip(M, dev, value, sh, lsi, loc, attn)
    torch.testing.assert_close(out, O.msda_forward(value, sh, lsi, loc, attn), **tol(torch.float32))
    monkeypatch.setattr(M, "PYR_FORWARD", False)
    rows = run_hip(M, dev, value, sh, lsi, loc, attn)
    torch.testing.assert_close(out, rows, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("kind", ["ring", "gauss", "uniform", "border"])
def test_pyramid_region_forward_full_size_n4_against_oracle(M, O, dev, kind):
    """The launch the training step actually makes -- N = 4 (source + target merged), the full
    1333x800 pyramid, Lq = S = 22 223 -- against the C oracle on every element.
    `ring`: pixel-centre reference points + the module's initial offset ring (what the encoder
    produces at initialisation; every sample inside the windows); `gauss`: offsets ~ N(0, 2 px)
    (a few per cent leave the windows); `uniform`: locations anywhere in the image (almost all
    through the slow path); `border`: offsets ~ N(0, 12 px), many beyond the image edges."""
    N, Mh, D, P = 4, 8, 32, 4
    value, sh, lsi, _, _ = O.random_inputs(N, 1, Mh, D, FULL_SHAPES, P, seed=21)
    S = value.shape[1]
    g = torch.Generator().manual_seed(22)
    attn = torch.softmax(torch.randn(N, S, Mh, 4 * P, generator=g), -1).view(N, S, Mh, 4, P)
    if kind == "uniform":
        loc = torch.rand(N, S, Mh, 4, P, 2, generator=g) * 1.1 - 0.05
    elif kind == "ring":
        ring = M.MSDeformAttn(256, 4, Mh, P).sampling_offsets.bias.detach().view(1, 1, Mh, 4, P, 2)
        wh = torch.tensor([[w, h] for h, w in FULL_SHAPES], dtype=torch.float32).view(1, 1, 1, 4, 1, 2)
        centres = pyramid_locs(FULL_SHAPES, N, Mh, P, 0.0, seed=23)
        loc = (centres + ring / wh).contiguous()
    else:
        loc = pyramid_locs(FULL_SHAPES, N, Mh, P, 2.0 if kind == "gauss" else 12.0, seed=23)
    out = run_hip(M, dev, value, sh, lsi, loc, attn)
    torch.testing.assert_close(out, O.msda_forward(value, sh, lsi, loc, attn), **tol(torch.float32))


@pytest.mark.parametrize("kind", ["ring", "gauss"])
def test_encoder_backward_full_size_n4_against_oracle(M, O, dev, kind):
    """The backward launches of the training step's merged encoder pass -- N = 4, Lq = S = 22 223: the
    LDS-window dots kernel (msda_bwd_dots_pyr2_d32) + the value-free sorted scatter (csrc/msda_bwd_pyr.hip),
    route 0 of datr_msda_backward_tiled_f32 -- against the C oracle on EVERY element."""
    N, Mh, D, P = 4, 8, 32, 4
    value, sh, lsi, _, _ = O.random_inputs(N, 1, Mh, D, FULL_SHAPES, P, seed=31)
    S = value.shape[1]
    g = torch.Generator().manual_seed(32)
    attn = torch.softmax(torch.randn(N, S, Mh, 4 * P, generator=g), -1).view(N, S, Mh, 4, P)
    if kind == "ring":
        ring = M.MSDeformAttn(256, 4, Mh, P).sampling_offsets.bias.detach().view(1, 1, Mh, 4, P, 2)
        wh = torch.tensor([[w, h] for h, w in FULL_SHAPES], dtype=torch.float32).view(1, 1, 1, 4, 1, 2)
        # a little noise: on the exact ring most samples sit ON grid lines, where d/d(loc) jumps
        loc = (pyramid_locs(FULL_SHAPES, N, Mh, P, 0.05, seed=33) + ring / wh).contiguous()
    else:
        loc = pyramid_locs(FULL_SHAPES, N, Mh, P, 3.0, seed=33)
    go = torch.randn(N, S, Mh * D, generator=g)
    out, gv, gl, ga = run_hip(M, dev, value, sh, lsi, loc, attn, go)
    torch.testing.assert_close(out, O.msda_forward(value, sh, lsi, loc, attn), **tol(torch.float32))
    rv, rl, ra = O.msda_backward(value, sh, lsi, loc, attn, go)
    scale = float(rv.abs().max())
    torch.testing.assert_close(gv, rv, rtol=1e-3, atol=1e-5 * scale)
    torch.testing.assert_close(ga, ra, **tol(torch.float32, 10))
    keep = off_grid(loc, sh)
    torch.testing.assert_close(gl[keep], rl[keep], **tol(torch.float32, 100))


@pytest.mark.parametrize("env_kind", ["measured", "tight", "lopsided"])
def test_encoder_backward_does_not_depend_on_the_envelope(M, O, dev, env_kind):
    """datr_msda_backward_pyramid_f32 sizes the windows of the pyramid-region backward by the offset
    envelope; samples beyond them take the direct-atomic path: whatever the envelope (measured, far too
    tight, one-sided), the gradients are the oracle's."""
    import numpy as np
    N, Mh, D, P = 2, 8, 32, 4
    value, sh, lsi, _, _ = O.random_inputs(N, 1, Mh, D, FULL_SHAPES, P, seed=41)
    S = value.shape[1]
    g = torch.Generator().manual_seed(42)
    attn = torch.softmax(torch.randn(N, S, Mh, 4 * P, generator=g), -1).view(N, S, Mh, 4, P)
    loc = pyramid_locs(FULL_SHAPES, N, Mh, P, 2.0, seed=43)
    go = torch.randn(N, S, Mh * D, generator=g)
    d = [t.to(dev) for t in (value, sh, lsi, loc, attn, go)]
    if env_kind == "measured":
        env = M.measure_envelope(d[3], d[1])
        assert env is not None
    elif env_kind == "tight":
        env = np.tile(np.array([-1.0, 1.0, -1.0, 1.0], np.float32), (8, 4, 1))
    else:
        env = np.tile(np.array([0.0, 9.0, -0.5, 0.5], np.float32), (8, 4, 1))
    gv, gl, ga = M.ms_deform_attn_backward(d[0], d[1], d[2], d[3], d[4], d[5], 64, envelope=env)
    rv, rl, ra = O.msda_backward(value, sh, lsi, loc, attn, go)
    scale = float(rv.abs().max())
    torch.testing.assert_close(gv.cpu(), rv, rtol=1e-3, atol=1e-5 * scale)
    torch.testing.assert_close(ga.cpu(), ra, **tol(torch.float32, 10))
    keep = off_grid(loc, sh)
    torch.testing.assert_close(gl.cpu()[keep], rl[keep], **tol(torch.float32, 100))


_SINGLE_KERNEL_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from datr_amd import msda as M
d = np.load(sys.argv[2])
dev = torch.device("cuda:0")
t = {k: torch.from_numpy(d[k]).to(dev) for k in d.files}
env = M.measure_envelope(t["loc"], t["sh"])
gv, gl, ga = M.ms_deform_attn_backward(t["value"], t["sh"], t["lsi"], t["loc"], t["attn"], t["go"], 64, envelope=env)
np.savez(sys.argv[3], gv=gv.cpu().numpy(), gl=gl.cpu().numpy(), ga=ga.cpu().numpy())
"""


def test_encoder_backward_two_kernels_match_the_single_kernel(M, O, dev, tmp_path):
    """The product backward of the encoder calls is two kernels (grad_loc / grad_attn out of the forward's
    LDS windows, grad_value by a value-free sorted scatter); DATR_MSDA_BWD_SPLIT=0 keeps the single kernel
    that gathers the corner rows itself (the path of shapes the forward's plan does not cover).  The switch
    is read once per process, so the single kernel runs in a child process on the same inputs."""
    import os
    import subprocess
    import sys
    import numpy as np
    N, Mh, D, P = 2, 8, 32, 4
    value, sh, lsi, _, _ = O.random_inputs(N, 1, Mh, D, FULL_SHAPES, P, seed=51)
    S = value.shape[1]
    g = torch.Generator().manual_seed(52)
    attn = torch.softmax(torch.randn(N, S, Mh, 4 * P, generator=g), -1).view(N, S, Mh, 4, P)
    loc = pyramid_locs(FULL_SHAPES, N, Mh, P, 2.0, seed=53)
    go = torch.randn(N, S, Mh * D, generator=g)
    d = [t.to(dev) for t in (value, sh, lsi, loc, attn, go)]
    env = M.measure_envelope(d[3], d[1])
    gv, gl, ga = M.ms_deform_attn_backward(d[0], d[1], d[2], d[3], d[4], d[5], 64, envelope=env)
    inp, out = str(tmp_path / "in.npz"), str(tmp_path / "out.npz")
    np.savez(inp, value=value.numpy(), sh=sh.numpy(), lsi=lsi.numpy(), loc=loc.numpy(), attn=attn.numpy(), go=go.numpy())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, "-c", _SINGLE_KERNEL_SCRIPT, root, inp, out], check=True, timeout=600,
                   env=dict(os.environ, DATR_MSDA_BWD_SPLIT="0"))
    ref = np.load(out)
    scale = float(np.abs(ref["gv"]).max())
    torch.testing.assert_close(gv.cpu(), torch.from_numpy(ref["gv"]), rtol=1e-3, atol=1e-5 * scale)
    torch.testing.assert_close(ga.cpu(), torch.from_numpy(ref["ga"]), **tol(torch.float32, 10))
    keep = off_grid(loc, sh)
    torch.testing.assert_close(gl.cpu()[keep], torch.from_numpy(ref["gl"])[keep], **tol(torch.float32, 100))


# Cityscapes -> Foggy Cityscapes frames are 1024x2048 after the x1.5 scaling capped at 2048
# (/root/reference/config/DA/Cityscapes2FoggyCityscapes/coco_transformer_C2F.py:1-7):
# the pyramid the mAP clause of the north star would run at.
C2F_SHAPES = [(128, 256), (64, 128), (32, 64), (16, 32)]


@pytest.mark.parametrize("kind", ["ring", "gauss"])
def test_c2f_geometry_forward_backward_against_oracle(M, O, dev, kind):
    """The real C2F geometry (level 0 = 128 x 256, S = 43 520, N = 2: one source + one target image)
    through the pyramid-region forward and backward kernels, every element against the C oracle:
    a different region grid, different window sizes and different query-table sizes than the
    1333x800 pyramid the other full-size tests use."""
    N, Mh, D, P = 2, 8, 32, 4
    value, sh, lsi, _, _ = O.random_inputs(N, 1, Mh, D, C2F_SHAPES, P, seed=41)
    S = value.shape[1]
    assert S == 43520
    g = torch.Generator().manual_seed(42)
    attn = torch.softmax(torch.randn(N, S, Mh, 4 * P, generator=g), -1).view(N, S, Mh, 4, P)
    if kind == "ring":
        ring = M.MSDeformAttn(256, 4, Mh, P).sampling_offsets.bias.detach().view(1, 1, Mh, 4, P, 2)
        wh = torch.tensor([[w, h] for h, w in C2F_SHAPES], dtype=torch.float32).view(1, 1, 1, 4, 1, 2)
        loc = (pyramid_locs(C2F_SHAPES, N, Mh, P, 0.05, seed=43) + ring / wh).contiguous()
    else:
        loc = pyramid_locs(C2F_SHAPES, N, Mh, P, 2.5, seed=43)
    go = torch.randn(N, S, Mh * D, generator=g)
    assert M.PYR_FORWARD
    out, gv, gl, ga = run_hip(M, dev, value, sh, lsi, loc, attn, go)
    torch.testing.assert_close(out, O.msda_forward(value, sh, lsi, loc, attn), **tol(torch.float32))
    rv, rl, ra = O.msda_backward(value, sh, lsi, loc, attn, go)
    scale = float(rv.abs().max())
    torch.testing.assert_close(gv, rv, rtol=1e-3, atol=1e-5 * scale)
    torch.testing.assert_close(ga, ra, **tol(torch.float32, 10))
    keep = off_grid(loc, sh)
    torch.testing.assert_close(gl[keep], rl[keep], **tol(torch.float32, 100))


def test_empty_and_fully_out_of_range(M, O, dev):
    value, sh, lsi, loc, attn = O.random_inputs(1, 0, 8, 32, [(4, 4)], 4, seed=1)
    assert run_hip(M, dev, value, sh, lsi, loc, attn).shape == (1, 0, 256)
    value, sh, lsi, loc, attn = O.random_inputs(2, 40, 8, 32, [(4, 4), (2, 2)], 4, seed=1,
                                                loc_range=(2.0, 3.0))
    go = torch.ones(2, 40, 256)
    out, gv, gl, ga = run_hip(M, dev, value, sh, lsi, loc, attn, go)
    for t_ in (out, gv, gl, ga):
        assert torch.count_nonzero(t_) == 0


def test_nan_location_is_skipped_like_the_reference(M, O, dev):
    # a NaN coordinate fails every comparison of cuh:288 -> the sample contributes nothing
    value, sh, lsi, loc, attn = O.random_inputs(1, 40, 8, 32, [(6, 6), (3, 3)], 2, seed=4)
    loc[0, 3, 2, 1, 0, 0] = float("nan")
    loc[0, 5, 1, 0, 1, 1] = float("inf")
    go = torch.randn(1, 40, 256)
    out, gv, gl, ga = run_hip(M, dev, value, sh, lsi, loc, attn, go)
    assert torch.isfinite(out).all() and torch.isfinite(gv).all()
    assert torch.isfinite(gl).all() and torch.isfinite(ga).all()
    torch.testing.assert_close(out, O.msda_forward(value, sh, lsi, loc, attn), **tol(torch.float32))


def test_autograd_function_and_gradcheck_in_double(M, dev):
    """Restates /root/reference/models/dino/ops/test.py:63-86 (gradcheck in double)."""
    from torch.autograd import gradcheck
    torch.manual_seed(3)
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long, device=dev)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    for D in (30, 32, 64, 71):
        value = (torch.rand(1, S, 2, D, device=dev) * 0.01).double().requires_grad_(True)
        loc = torch.rand(1, 2, 2, 2, 2, 2, device=dev).double().requires_grad_(True)
        attn = torch.rand(1, 2, 2, 2, 2, device=dev) + 1e-5
        attn = (attn / attn.sum(-1, keepdim=True).sum(-2, keepdim=True)).double().requires_grad_(True)
        assert gradcheck(M.MSDeformAttnFunction.apply, (value, shapes, lsi, loc, attn, 2))


@pytest.mark.parametrize("D", [1025, 2048, 3096])
def test_large_channel_directional_derivative(M, dev, D):
    """The reference gradchecks D in {1025, 2048, 3096} (ops/test.py:85); a full numerical
    Jacobian at those widths is minutes of work, so check <grad, direction> against a central
    difference along random directions instead."""
    torch.manual_seed(5)
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long, device=dev)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    value = (torch.rand(1, 30, 2, D, device=dev)).double().requires_grad_(True)
    loc = (torch.rand(1, 2, 2, 2, 2, 2, device=dev) * 0.8 + 0.1).double().requires_grad_(True)
    attn = torch.rand(1, 2, 2, 2, 2, device=dev).double().requires_grad_(True)
    go = torch.randn(1, 2, 2 * D, device=dev).double()
    f = lambda v, l, a: (M.MSDeformAttnFunction.apply(v, shapes, lsi, l, a, 2) * go).sum()
    gv, gl, ga = torch.autograd.grad(f(value, loc, attn), (value, loc, attn))
    eps = 1e-6
    for x, g, args in ((value, gv, 0), (loc, gl, 1), (attn, ga, 2)):
        d = torch.randn_like(x)
        xs = [value.detach(), loc.detach(), attn.detach()]
        xp = list(xs); xp[args] = xs[args] + eps * d
        xm = list(xs); xm[args] = xs[args] - eps * d
        num = (f(*xp) - f(*xm)) / (2 * eps)
        torch.testing.assert_close((g * d).sum(), num, rtol=1e-5, atol=1e-7)


def test_error_behaviour_on_device(M, dev):
    v = torch.zeros(3, 4, 2, 16, device=dev)
    sh = torch.tensor([[2, 2]], device=dev)
    lsi = torch.tensor([0], device=dev)
    loc = torch.zeros(3, 1, 2, 1, 1, 2, device=dev)
    att = torch.zeros(3, 1, 2, 1, 1, device=dev)
    with pytest.raises(RuntimeError, match="must divide"):
        M.ms_deform_attn_forward(v, sh, lsi, loc, att, 2)      # 3 % 2 != 0 (cu:50-52)
    with pytest.raises(RuntimeError, match="contiguous"):
        M.ms_deform_attn_forward(v.transpose(0, 1), sh, lsi, loc, att, 64)
    assert M.ms_deform_attn_forward(v, sh, lsi, loc, att, 64).shape == (3, 1, 32)


# ---------------------------------------------------------------------------------------------
# BASELINE sizes: N=2, S=22223; encoder (Lq=S) and decoder (Lq=1100 / 900) calls
# ---------------------------------------------------------------------------------------------
def full_inputs(O, dev, Lq, seed, loc_range=(0.0, 1.0)):
    value, sh, lsi, loc, attn = O.random_inputs(2, Lq, 8, 32, FULL_SHAPES, 4, seed=seed,
                                                loc_range=loc_range)
    return [t.to(dev) for t in (value, sh, lsi, loc, attn)]


@pytest.mark.parametrize("Lq", [22223, 1100, 900])
def test_full_size_properties(M, O, dev, Lq):
    value, sh, lsi, loc, attn = full_inputs(O, dev, Lq, seed=3, loc_range=(-0.05, 1.05))
    f = lambda v, a: M.ms_deform_attn_forward(v, sh, lsi, loc, a, 64)
    out = f(value, attn)
    # determinism of the forward (no atomics)
    assert torch.equal(out, f(value, attn))
    # linearity in value and in the attention weights
    v2 = torch.rand_like(value) * 0.01
    torch.testing.assert_close(f(value + v2, attn), out + f(v2, attn), rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(f(value, attn * 0.5), out * 0.5, rtol=1e-5, atol=1e-8)
    # fp32 fast path vs the fp64 generic kernels on the same inputs
    out64 = M.ms_deform_attn_forward(value.double(), sh, lsi, loc.double(), attn.double(), 64)
    torch.testing.assert_close(out.double(), out64, rtol=1e-4, atol=1e-7)
    # adjoint identities: out is linear in value and in attn, so
    #   <go, out> == <grad_value, value> == <grad_attn, attn>
    go = torch.randn_like(out)
    gv, gl, ga = M.ms_deform_attn_backward(value, sh, lsi, loc, attn, go, 64)
    lhs = (go.double() * out.double()).sum()
    torch.testing.assert_close((gv.double() * value.double()).sum(), lhs, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close((ga.double() * attn.double()).sum(), lhs, rtol=1e-4, atol=1e-6)
    # grad_loc / grad_attn use no atomics -> bitwise reproducible; grad_value to rounding
    gv2, gl2, ga2 = M.ms_deform_attn_backward(value, sh, lsi, loc, attn, go, 64)
    assert torch.equal(gl, gl2) and torch.equal(ga, ga2)
    scale = float(gv.abs().max())         # hot pixels sum thousands of atomically-added terms
    torch.testing.assert_close(gv, gv2, rtol=1e-3, atol=1e-5 * scale)
    # backward against the fp64 generic kernels
    gv64, gl64, ga64 = M.ms_deform_attn_backward(value.double(), sh, lsi, loc.double(),
                                                 attn.double(), go.double(), 64)
    torch.testing.assert_close(gv.double(), gv64, rtol=1e-3, atol=1e-5 * scale)
    torch.testing.assert_close(ga.double(), ga64, rtol=1e-3, atol=1e-6)
    # d/d(loc) jumps where a pixel coordinate crosses an integer; fp32 and fp64 can floor to
    # different sides there, so leave out samples within 1e-3 px of a grid line
    keep = off_grid(loc, sh)
    torch.testing.assert_close(gl.double()[keep], gl64[keep], rtol=1e-3, atol=1e-4)


def test_full_size_subsample_against_oracle(M, O, dev):
    """Decoder-sized call (Lq=900) at the full pyramid against the C oracle (seconds on CPU)."""
    value, sh, lsi, loc, attn = O.random_inputs(2, 900, 8, 32, FULL_SHAPES, 4, seed=9,
                                                loc_range=(-0.05, 1.05))
    go = torch.randn(2, 900, 256, generator=torch.Generator().manual_seed(4))
    out, gv, gl, ga = run_hip(M, dev, value, sh, lsi, loc, attn, go)
    torch.testing.assert_close(out, O.msda_forward(value, sh, lsi, loc, attn), **tol(torch.float32))
    rv, rl, ra = O.msda_backward(value, sh, lsi, loc, attn, go)
    scale = float(rv.abs().max())
    torch.testing.assert_close(gv, rv, rtol=1e-3, atol=1e-5 * scale)
    torch.testing.assert_close(ga, ra, **tol(torch.float32, 10))
    keep = off_grid(loc, sh)
    torch.testing.assert_close(gl[keep], rl[keep], **tol(torch.float32, 100))


@pytest.mark.parametrize("mode", ["clustered", "spread"])
def test_owner_backward_for_decoder_queries(M, O, dev, mode, monkeypatch):
    """Lq != S, a few hundred queries: workgroups own ranges of value rows (msda_bwd_owner.hip).
    `clustered` piles every query onto the same few pixels (the contention case the kernel
    exists for), `spread` scatters them over all levels / ranges.  The output buffers are
    pre-filled with NaN: the kernel must write every element itself (it does no zero-fill)."""
    shapes = [(40, 53), (20, 27), (10, 14), (5, 7)]               # level 0: 2120 rows = 5 ranges
    N, Lq, Mh, D, P = 2, 333, 8, 32, 4
    value, sh, lsi, loc, attn = O.random_inputs(N, Lq, Mh, D, shapes, P, seed=21, loc_range=(-0.1, 1.1))
    if mode == "clustered":
        g = torch.Generator().manual_seed(4)
        loc = (0.37 + 0.01 * torch.randn(loc.shape, generator=g)).contiguous()
    go = torch.randn(N, Lq, Mh * D, generator=torch.Generator().manual_seed(9)) * 2.0
    real_empty = torch.empty_like

    def nan_empty(t, **kw):
        out = real_empty(t, **kw)
        return out.fill_(float("nan")) if out.is_floating_point() and out.is_cuda else out
    monkeypatch.setattr(torch, "empty_like", nan_empty)
    out, gv, gl, ga = run_hip(M, dev, value, sh, lsi, loc, attn, go)
    monkeypatch.setattr(torch, "empty_like", real_empty)
    rv, rl, ra = O.msda_backward(value, sh, lsi, loc, attn, go)
    assert torch.isfinite(gv).all() and torch.isfinite(gl).all() and torch.isfinite(ga).all()
    scale = float(rv.abs().max())
    # fixed-point step = max|grad_out| * sum|attn over the level| / 2^30 (~3e-7 of max|grad_value|
    # here); ~150 contributions meet on a row of the 5x7 level: allow 5e-6 of the maximum
    torch.testing.assert_close(gv, rv, rtol=1e-4, atol=5e-6 * scale)
    torch.testing.assert_close(ga, ra, **tol(torch.float32, 10))
    keep = off_grid(loc, sh, eps=1e-4)
    torch.testing.assert_close(gl[keep], rl[keep], **tol(torch.float32, 100))


def test_routes_agree_and_the_offset_monitor_switches_them():
    """The three kernel routes of the encoder call (0: pyramid regions, 1: row forward + query-tiled
    backward, 2: row kernels) compute the same op -- compared on offsets wide enough that a large
    share of the samples leaves the pyramid windows -- and datr_amd.msda.OffsetMonitor moves an
    encoder layer from route 0 to the others once it has seen such offsets (asynchronous read-back:
    the switch happens on a later call, never through a host synchronisation inside the call)."""
    from datr_amd import msda
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    shapes_l = [(40, 54), (20, 27), (10, 14), (5, 7)]
    shapes = torch.tensor(shapes_l, dtype=torch.int64, device=dev)
    lsi = torch.cat([shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]])
    S = int(shapes.prod(1).sum())
    N, M, D, L, P = 2, 8, 32, 4, 4
    value = torch.randn(N, S, M, D, device=dev)
    ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h, device=dev) + 0.5) / h,
                                                (torch.arange(w, device=dev) + 0.5) / w, indexing="ij"), -1)
                     .flip(-1).reshape(-1, 2) for h, w in shapes_l], 0)
    wh = torch.tensor([[w, h] for h, w in shapes_l], dtype=torch.float32, device=dev).view(1, 1, 1, L, 1, 2)
    loc = (ref.view(1, S, 1, 1, 1, 2) + torch.randn(N, S, M, L, P, 2, device=dev) * 5.0 / wh).contiguous()
    attn = torch.softmax(torch.randn(N, S, M, L * P, device=dev), -1).view(N, S, M, L, P)
    go = torch.randn(N, S, M * D, device=dev)
    res = []
    for route in (0, 1, 2):
        v, l_, a = (x.clone().requires_grad_(True) for x in (value, loc, attn))
        out = msda.MSDeformAttnFunction.apply(v, shapes, lsi, l_, a, 64, route)
        res.append([out.detach()] + list(torch.autograd.grad(out, (v, l_, a), go)))
    for other in res[1:]:
        for a, b in zip(res[0], other):
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4 * max(1.0, float(b.abs().max())))

    assert [msda.OffsetMonitor.route_for(f) for f in (0.0, 0.2, 0.3, 0.6, 0.7, 1.0)] == [0, 0, 1, 1, 2, 2]
    mod = msda.MSDeformAttn(256, 4, 8, 4).to(dev)
    query = torch.randn(N, S, 256, device=dev)
    src = torch.randn(N, S, 256, device=dev)
    refpts = ref.view(1, S, 1, 2).expand(N, S, L, 2).contiguous()

    def call():
        return mod(query, refpts, src, shapes, lsi)
    y0 = call()                                                   # call 1: measures
    call()
    call()                                                        # call 1 + LAG: the measurement is applied
    mon = msda._MONITORS[mod]
    assert mon.route == 0 and mon.fraction == 0.0                 # ring initialisation: nothing beyond 4 px
    # the envelope the phased forward sizes its windows by: head 0 looks along +x only (1..4 px)
    env = mon.envelope
    assert env is not None and env.shape == (8, 4, 4) and env.dtype == np.float32
    ring = mod.sampling_offsets.bias.detach().view(8, 4, 4, 2).cpu()
    for h in range(8):
        for l in range(4):
            assert env[h, l, 0] <= float(ring[h, l, :, 1].min()) <= float(ring[h, l, :, 1].max()) <= env[h, l, 1]
            assert env[h, l, 2] <= float(ring[h, l, :, 0].min()) <= float(ring[h, l, :, 0].max()) <= env[h, l, 3]
            assert env[h, l, 1] - env[h, l, 0] <= 3.0 + 2 * mon.MARGIN_PX + 2 * mon.GRID_PX + 1e-3   # margin + outward rounding
    assert msda.pyramid_plan(shapes, lsi, N, 8, 32, 4, env)["phased"]
    y0b = call()                                                  # now through the phased kernel
    torch.testing.assert_close(y0b, y0, rtol=1e-4, atol=1e-4)
    with torch.no_grad():
        mod.sampling_offsets.bias.mul_(2.5)                       # up to 10 px: most points leave the halo
    mon.calls = 0                                                 # next call probes again
    y1 = call()
    call()
    y2 = call()                                                   # LAG calls later: new route
    assert mon.fraction > 0.5 and mon.route == msda.OffsetMonitor.route_for(mon.fraction) and mon.route > 0
    torch.testing.assert_close(y1, y2, rtol=1e-4, atol=1e-4)      # route 0 vs the new route: same values
    assert y0.shape == y1.shape


def _phased(M, dev, value, sh, lsi, loc, attn, envelope, force=False):
    """The phased all-LDS forward (csrc/msda_fwd_pyr2.hip).  force: call the kernel's own entry even
    where the public dispatch would prefer round 2's kernel (multi-phase plans)."""
    from datr_amd import _native
    v, s_, a = value.to(dev), loc.to(dev), attn.to(dev)
    if not force:
        return M.ms_deform_attn_forward(v, sh.to(dev), lsi.to(dev), s_, a, 64, envelope=envelope).cpu()
    N, S, Mh, D = v.shape
    out = torch.full((N, S, Mh * D), float("nan"), device=dev)
    shh, lsh = sh.cpu().numpy().copy(), lsi.cpu().numpy().copy()
    fn = _native.lib.datr_internal_msda_fwd_pyr2_d32
    import ctypes
    fn.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int64] * 7 + [ctypes.c_void_p, ctypes.c_void_p]
    fn.restype = ctypes.c_int
    rc = fn(v.data_ptr(), s_.data_ptr(), a.data_ptr(), shh.ctypes.data, lsh.ctypes.data,
            None if envelope is None else envelope.ctypes.data, N, S, Mh, D, 4, S, 4, out.data_ptr(),
            _native.current_stream_ptr(dev))
    assert rc == 0, rc
    return out.cpu()


@pytest.mark.parametrize("shapes,spread,env_kind", [
    ([(20, 27), (10, 14), (5, 7), (3, 4)], 1.5, "measured"),
    ([(40, 61), (20, 31), (10, 16), (5, 8)], 1.0, "measured"),     # several regions per axis
    ([(40, 61), (20, 31), (10, 16), (5, 8)], 4.0, "symmetric"),    # multi-phase plan, forced
    ([(33, 50), (17, 25), (9, 13), (5, 7)], 6.0, "tight"),         # envelope far too small: slow path
    ([(33, 50), (17, 25), (9, 13), (5, 7)], 3.0, "lopsided"),      # per-head directional envelopes
    ([(64, 96), (32, 48), (16, 24), (8, 12)], 2.0, "measured"),
])
def test_phased_pyramid_forward_matches_oracle(M, O, dev, shapes, spread, env_kind):
    """csrc/msda_fwd_pyr2.hip: all four levels out of LDS windows sized by a per-(head, level)
    envelope.  Whatever the envelope says -- measured, symmetric, much too tight (every sample
    through the global-memory slow path), lopsided per head -- the result equals the oracle's."""
    N, Mh, D, P = 2, 8, 32, 4
    value, sh, lsi, _, _ = O.random_inputs(N, 1, Mh, D, shapes, P, seed=13)
    S = value.shape[1]
    loc = pyramid_locs(shapes, N, Mh, P, spread, seed=7)
    if env_kind == "lopsided":            # head m looks along direction m, like the ring initialisation
        ring = M.MSDeformAttn(256, 4, Mh, P).sampling_offsets.bias.detach().view(1, 1, Mh, 4, P, 2)
        wh = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32).view(1, 1, 1, 4, 1, 2)
        loc = (pyramid_locs(shapes, N, Mh, P, 0.3, seed=7) + ring / wh).contiguous()
    g = torch.Generator().manual_seed(8)
    attn = torch.softmax(torch.randn(N, S, Mh, 4 * P, generator=g), -1).view(N, S, Mh, 4, P)
    if env_kind in ("measured", "lopsided"):
        env = M.measure_envelope(loc.to(dev), sh)
    elif env_kind == "symmetric":
        env = None
    else:
        env = np.tile(np.array([-0.4, 0.4, -0.4, 0.4], np.float32), (8, 4, 1))
    want = O.msda_forward(value, sh, lsi, loc, attn)
    plan = M.pyramid_plan(sh, lsi, N, Mh, D, P, env)
    assert plan["forward"]
    if env_kind == "lopsided":
        assert plan["phased"] and float((env[:, :, 1] - env[:, :, 0]).min()) < 2.5      # axis-aligned heads
    out = _phased(M, dev, value, sh, lsi, loc, attn, env, force=True)
    assert not torch.isnan(out).any()
    torch.testing.assert_close(out, want, **tol(torch.float32))
    torch.testing.assert_close(_phased(M, dev, value, sh, lsi, loc, attn, env), want, **tol(torch.float32))


@pytest.mark.parametrize("geometry", ["1333x800", "c2f"])
@pytest.mark.parametrize("kind", ["ring", "gauss", "border"])
def test_phased_pyramid_forward_full_size_against_oracle(M, O, dev, geometry, kind):
    """The launch the training step makes once the monitor has measured the envelope: N = 4 at the
    1333x800 pyramid (N = 2 at the C2F geometry 128x256), every element against the C oracle.
    `ring` is the one-phase plan of the phased kernel (asserted); `gauss` / `border` have wide
    envelopes and go to whichever kernel the dispatch picks -- and are forced through the phased
    kernel's multi-phase form as well."""
    shapes = FULL_SHAPES if geometry == "1333x800" else C2F_SHAPES
    N, Mh, D, P = (4 if geometry == "1333x800" else 2), 8, 32, 4
    value, sh, lsi, _, _ = O.random_inputs(N, 1, Mh, D, shapes, P, seed=21)
    S = value.shape[1]
    g = torch.Generator().manual_seed(22)
    attn = torch.softmax(torch.randn(N, S, Mh, 4 * P, generator=g), -1).view(N, S, Mh, 4, P)
    if kind == "ring":
        ring = M.MSDeformAttn(256, 4, Mh, P).sampling_offsets.bias.detach().view(1, 1, Mh, 4, P, 2)
        wh = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32).view(1, 1, 1, 4, 1, 2)
        loc = (pyramid_locs(shapes, N, Mh, P, 0.05, seed=23) + ring / wh).contiguous()
    else:
        loc = pyramid_locs(shapes, N, Mh, P, 2.0 if kind == "gauss" else 12.0, seed=23)
    env = M.measure_envelope(loc.to(dev), sh)
    plan = M.pyramid_plan(sh, lsi, N, Mh, D, P, env)
    if kind == "ring":
        assert plan["phased"] and plan["phases"] == 1, plan
    want = O.msda_forward(value, sh, lsi, loc, attn)
    torch.testing.assert_close(_phased(M, dev, value, sh, lsi, loc, attn, env), want, **tol(torch.float32))
    if plan["forward"]:
        torch.testing.assert_close(_phased(M, dev, value, sh, lsi, loc, attn, env, force=True), want,
                                   **tol(torch.float32))


def test_padded_rows_are_zeroed_in_place_like_masked_fill():
    """MSDeformAttn with a padding mask: `value.masked_fill(mask[..., None], 0)`
    (ms_deform_attn.py:101-102) done by the in-place row kernel (and again on the gradient) gives the
    module output and every gradient of the masked_fill formulation."""
    from datr_amd import msda
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    shapes = [(12, 17), (6, 9), (3, 5), (2, 3)]
    S = sum(h * w for h, w in shapes)
    attn = msda.MSDeformAttn(256, 4, 8, 4).to(dev)
    with torch.no_grad():
        attn.sampling_offsets.weight.normal_(0, 0.02)
        attn.attention_weights.weight.normal_(0, 0.02)
    src = torch.randn(2, S, 256, device=dev, requires_grad=True)
    ref = torch.rand(2, S, 4, 2, device=dev)
    mask = torch.rand(2, S, device=dev) < 0.3
    ss = torch.tensor(shapes, device=dev)
    lsi = torch.cat([ss.new_zeros(1), (ss[:, 0] * ss[:, 1]).cumsum(0)[:-1]])
    go = torch.randn(2, S, 256, device=dev)
    params = list(attn.parameters())
    calls = []
    real = msda._zero_rows_
    msda._zero_rows_ = lambda x, m: (calls.append(1), real(x, m))[1]
    try:
        own = torch.autograd.grad(attn(src, ref, src, ss, lsi, mask), [src] + params, go)
    finally:
        msda._zero_rows_ = real
    assert len(calls) == 2                                          # forward and backward
    saved = msda.zero_padded_rows
    msda.zero_padded_rows = lambda v, m: v.masked_fill(m[..., None], 0.0)
    try:
        lib = torch.autograd.grad(attn(src, ref, src, ss, lsi, mask), [src] + params, go)
    finally:
        msda.zero_padded_rows = saved
    for a, b in zip(own, lib):                   # (the row kernel's float atomics are not order-stable: no bitwise claim)
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5 * float(b.abs().max()))
    # the op itself, bit for bit, on a tensor of its own
    v = torch.randn(2, S, 256, device=dev)
    want = v.masked_fill(mask[..., None], 0.0)
    got = msda.zero_padded_rows(v.clone().requires_grad_(True) * 1.0, mask)
    assert torch.equal(got, want)


def test_batched_value_projections_match_separate_modules(M, dev):
    """msda.value_projections: the value projections of several attention modules on one memory as one
    autograd node whose backward reads the value gradients out of a shared buffer (written with a row
    stride by datr_msda_backward_strided_f32) -- same outputs, same gradients as the modules one by one
    (the decoder layers of deformable_transformer.py:880-900)."""
    torch.manual_seed(7)
    shapes = [(20, 30), (10, 15), (5, 8), (3, 4)]
    sh = torch.tensor(shapes, dtype=torch.int64, device=dev)
    lsi = torch.cat([sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]])
    S, N, Lq, n = int(sh.prod(1).sum()), 2, 50, 3
    attns = [M.MSDeformAttn(256, 4, 8, 4).to(dev) for _ in range(n)]
    for a in attns:                                   # spread the samples: the ring alone sits on grid lines
        torch.nn.init.normal_(a.sampling_offsets.weight, std=0.02)
        torch.nn.init.normal_(a.attention_weights.weight, std=0.02)
    mem0 = torch.randn(N, S, 256, device=dev)
    qs = [torch.randn(N, Lq, 256, device=dev) for _ in range(n)]
    ref = torch.rand(N, Lq, 4, 4, device=dev) * 0.6 + 0.2
    ws = [torch.randn(N, Lq, 256, device=dev) for _ in range(n)]

    def run(batched):
        for a in attns:
            a.zero_grad(set_to_none=True)
        mem = mem0.clone().requires_grad_(True)
        if batched:
            res = M.value_projections(mem, attns)
            assert res is not None
            values, slab = res
            outs = [a(q, ref, mem, sh, lsi, None, value=values[i], grad_slot=(slab, i))
                    for i, (a, q) in enumerate(zip(attns, qs))]
        else:
            outs = [a(q, ref, mem, sh, lsi, None) for a, q in zip(attns, qs)]
        sum((o * w).sum() for o, w in zip(outs, ws)).backward()
        grads = [mem.grad.clone()]
        for a in attns:
            grads += [a.value_proj.weight.grad.clone(), a.value_proj.bias.grad.clone(),
                      a.sampling_offsets.weight.grad.clone(), a.output_proj.weight.grad.clone()]
        return [o.detach().clone() for o in outs], grads

    o1, g1 = run(False)
    o2, g2 = run(True)
    for a, b in zip(o1, o2):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
    for a, b in zip(g1, g2):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-5 * float(a.abs().max()))


@pytest.mark.parametrize("kind,N", [("ring", 4), ("gauss", 2)])
def test_encoder_backward_query_gradient_rows_against_oracle(M, O, dev, kind, N):
    """datr_msda_backward_pyramid_query_f32 (ms_deform_attn_backward_query_grad): the encoder call's backward
    leaving the gradient of the module's merged query projection -- [grad_sampling_loc | softmax backward of
    grad_attn_weight] in rows of 384 -- at the training step's size against the C oracle's grad_loc / grad_attn
    pushed through the softmax backward in float64 (ms_deform_attn.py:106), and against the two-pass route
    (ms_deform_attn_backward + the prologue's backward kernel) it replaces."""
    Mh, D, P = 8, 32, 4
    value, sh, lsi, _, _ = O.random_inputs(N, 1, Mh, D, FULL_SHAPES, P, seed=51)
    S = value.shape[1]
    g = torch.Generator().manual_seed(52)
    attn = torch.softmax(torch.randn(N, S, Mh, 4 * P, generator=g), -1).view(N, S, Mh, 4, P)
    if kind == "ring":
        ring = M.MSDeformAttn(256, 4, Mh, P).sampling_offsets.bias.detach().view(1, 1, Mh, 4, P, 2)
        wh = torch.tensor([[w, h] for h, w in FULL_SHAPES], dtype=torch.float32).view(1, 1, 1, 4, 1, 2)
        loc = (pyramid_locs(FULL_SHAPES, N, Mh, P, 0.05, seed=53) + ring / wh).contiguous()
    else:
        loc = pyramid_locs(FULL_SHAPES, N, Mh, P, 3.0, seed=53)
    go = torch.randn(N, S, Mh * D, generator=g)
    d = [t.to(dev) for t in (value, sh, lsi, loc, attn, go)]
    import numpy as np
    # ring: the measured envelope, every sample inside its window; gauss (sigma 3 px): windows of +-2.5 px, so that
    # a good part of the samples goes through the kernel's global-memory path (results never depend on the envelope)
    env = M.measure_envelope(d[3], d[1]) if kind == "ring" else np.tile(np.array([-2.5, 2.5, -2.5, 2.5], np.float32), (8, 4, 1))
    plan = M.pyramid_plan(d[1], d[2], N, Mh, D, P, env)
    done = M.ms_deform_attn_backward_query_grad(d[0], d[1], d[2], d[3], d[4], d[5], 64, envelope=env)
    assert plan["phased"] and plan["tasks_per_wave"] == (2 if kind == "ring" else 3), plan     # both kernel variants
    assert done is not None
    gv, gq = done
    assert gq.shape == (N, S, 384)
    rv, rl, ra = O.msda_backward(value, sh, lsi, loc, attn, go)
    scale = float(rv.abs().max())
    torch.testing.assert_close(gv.cpu(), rv, rtol=1e-3, atol=1e-5 * scale)
    a64, g64 = attn.double().view(N, S, Mh, 16), ra.double().view(N, S, Mh, 16)
    dlogit = (a64 * (g64 - (a64 * g64).sum(-1, keepdim=True))).float().view(N, S, 128)
    torch.testing.assert_close(gq[..., 256:].cpu(), dlogit, rtol=1e-4, atol=1e-5 * float(dlogit.abs().max()))
    keep = off_grid(loc, sh)
    torch.testing.assert_close(gq[..., :256].cpu().view(N, S, Mh, 4, P, 2)[keep], rl[keep], **tol(torch.float32, 100))
    # the two-pass route: the same kernels followed by csrc/msda_prologue.hip's backward
    gv2, gl2, ga2 = M.ms_deform_attn_backward(d[0], d[1], d[2], d[3], d[4], d[5], 64, envelope=env)
    ref2 = torch.zeros(N, S, 4, 2, device=dev)
    both2 = M._prologue_backward(gl2, ga2, d[4].view(N, S, Mh, 4, P), ref2, (N, S, 384))
    assert torch.equal(gq[..., :256], both2[..., :256])                  # grad_loc: the same values, another place
    torch.testing.assert_close(gq[..., 256:], both2[..., 256:], rtol=1e-5, atol=1e-6 * float(dlogit.abs().max()))


def test_query_gradient_backward_declines_what_it_does_not_cover(M, O, dev):
    """Decoder-like calls (Lq != S) return None: the module then takes the two-pass route."""
    value, sh, lsi, loc, attn = O.random_inputs(2, 300, 8, 32, [(25, 34), (13, 17), (7, 9), (4, 5)], 4, seed=5)
    d = [t.to(dev) for t in (value, sh, lsi, loc, attn)]
    go = torch.randn(2, 300, 256, device=dev)
    assert M.ms_deform_attn_backward_query_grad(d[0], d[1], d[2], d[3], d[4], go, 64) is None


def test_module_with_query_gradient_node_equals_the_two_node_module(M, dev, monkeypatch):
    """MSDeformAttn.forward on an encoder self-attention call: the one-node form (_PrologueMSDA, backward through
    datr_msda_backward_pyramid_query_f32) against the two nodes (DATR_MSDA_QUERY_GRAD=0): output bitwise equal,
    gradients of query / input / parameters equal to fp32 rounding (float atomics in grad_value)."""
    torch.manual_seed(11)
    shapes = [(100, 167), (50, 84), (25, 42), (13, 21)]
    S = sum(h * w for h, w in shapes)
    attn_mod = M.MSDeformAttn(256, 4, 8, 4).to(dev)
    sh = torch.tensor(shapes, device=dev)
    lsi = torch.cat([sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]])
    ref = torch.cat([torch.stack(torch.meshgrid(
        (torch.arange(h, device=dev) + 0.5) / h, (torch.arange(w, device=dev) + 0.5) / w, indexing="ij"), -1)
        .flip(-1).reshape(-1, 2) for h, w in shapes], 0)[None, :, None, :].expand(2, S, 4, 2).contiguous()
    q = torch.randn(2, S, 256, device=dev, requires_grad=True)
    x = torch.randn(2, S, 256, device=dev, requires_grad=True)
    go = torch.randn(2, S, 256, device=dev)
    res = {}
    for on in (True, False):
        monkeypatch.setattr(M, "QUERY_GRAD_BACKWARD", on)
        y = attn_mod(q, ref, x, sh, lsi, None)
        names = []
        fn = y.grad_fn
        seen, todo = set(), [fn]
        while todo:
            f = todo.pop()
            if f is None or f in seen:
                continue
            seen.add(f)
            names.append(type(f).__name__)
            todo.extend(n for n, _ in f.next_functions)
        assert any("PrologueMSDA" in n for n in names) == on
        res[on] = (y, torch.autograd.grad(y, [q, x] + list(attn_mod.parameters()), go))
    assert torch.equal(res[True][0], res[False][0])
    for a, b in zip(res[True][1], res[False][1]):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-7


def _pyramid_of(height, width):
    """Feature-map sizes of the detector for an image of height x width: ResNet-50 strides 8 / 16 / 32 (every stride-2
    stage rounds up) + the extra 3x3 stride-2 level (dino.py:120-124)."""
    def up(v, n):
        for _ in range(n):
            v = (v + 1) // 2
        return v
    return [(up(height, k), up(width, k)) for k in (3, 4, 5, 6)]


# image sizes a multi-scale training run produces (datasets/coco transforms: shorter side 480 .. 800 in steps of 32,
# longer side <= 1333; C2F scales by 1.5): landscape, portrait, near-square, odd sizes whose levels round differently
MULTISCALE = [(480, 800), (512, 683), (544, 1088), (608, 1013), (672, 1191), (736, 981), (800, 1066), (1201, 800),
              (720, 1280), (901, 1333)]


@pytest.mark.parametrize("size", MULTISCALE, ids=lambda s: f"{s[0]}x{s[1]}")
def test_pyramid_kernels_over_multiscale_geometries(M, O, dev, size):
    """The region / window plans of the pyramid kernels are geometry dependent (region grid, window extents, envelope
    trims, phases, tasks per wave): every image size of a multi-scale run is another plan.  Forward, LDS-window dots and
    sorted scatter against the C oracle on every element, for ring-like and for spread offsets, with the measured
    envelope (what the model passes) and without one."""
    shapes = _pyramid_of(*size)
    N, Mh, D, P = 2, 8, 32, 4
    value, sh, lsi, _, _ = O.random_inputs(N, 1, Mh, D, shapes, P, seed=size[0] + size[1])
    S = value.shape[1]
    assert S == sum(h * w for h, w in shapes)
    g = torch.Generator().manual_seed(size[0])
    attn = torch.softmax(torch.randn(N, S, Mh, 4 * P, generator=g), -1).view(N, S, Mh, 4, P)
    go = torch.randn(N, S, Mh * D, generator=g)
    ring = M.MSDeformAttn(256, 4, Mh, P).sampling_offsets.bias.detach().view(1, 1, Mh, 4, P, 2)
    wh = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32).view(1, 1, 1, 4, 1, 2)
    for kind in ("ring", "spread"):
        if kind == "ring":
            loc = (pyramid_locs(shapes, N, Mh, P, 0.3, seed=size[1]) + ring / wh).contiguous()
        else:
            loc = pyramid_locs(shapes, N, Mh, P, 2.5, seed=size[1] + 1)
        ref = O.msda_forward(value, sh, lsi, loc, attn)
        rv, rl, ra = O.msda_backward(value, sh, lsi, loc, attn, go)
        scale = float(rv.abs().max())
        keep = off_grid(loc, sh)
        v, s, a = value.to(dev), loc.to(dev), attn.to(dev)
        shd, lsd = sh.to(dev), lsi.to(dev)
        for env in (M.measure_envelope(s, shd), None):
            kw = {} if env is None else {"envelope": env}
            out = M.ms_deform_attn_forward(v, shd, lsd, s, a, 64, **kw).cpu()
            torch.testing.assert_close(out, ref, **tol(torch.float32))
            gv, gl, ga = (x.cpu() for x in M.ms_deform_attn_backward(v, shd, lsd, s, a, go.to(dev), 64, **kw))
            torch.testing.assert_close(gv, rv, rtol=1e-3, atol=1e-5 * scale)
            torch.testing.assert_close(ga, ra, **tol(torch.float32, 10))
            torch.testing.assert_close(gl[keep], rl[keep], **tol(torch.float32, 100))
