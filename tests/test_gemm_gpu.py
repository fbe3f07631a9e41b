"""The exact-fp32 MFMA GEMM family with fused epilogues (csrc/gemm_f32.hip) against float64 products:
the three operand forms, every epilogue term, ragged edges, row-slice operands, determinism.
Mirrors the op sequences of /root/reference/models/dino/backbone.py:62-72 (frozen BN around the 1x1
convolutions of torchvision's Bottleneck) and deformable_transformer.py:803-806 (FFN)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _close(got, ref, tol=4e-6):
    scale = max(ref.abs().max().item(), 1e-30)
    err = (got.double() - ref).abs().max().item() / scale
    assert err < tol, err


SHAPES = [(1, 4, 32), (63, 64, 32), (64, 60, 64), (129, 132, 96), (300, 64, 256), (257, 256, 64), (1000, 36, 128),
          (4200, 512, 2048), (515, 2048, 256)]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_nt_forward_epilogues(M, N, K):
    from datr_amd import gemm
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(M * 7 + N)
    x = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g)
    scale = torch.rand(N, device=dev, generator=g) + 0.5
    shift = torch.randn(N, device=dev, generator=g)
    res = torch.randn(M, N, device=dev, generator=g) * 4
    base = x.double() @ w.double().t()
    _close(gemm.gemm_nt(x, w), base)
    _close(gemm.gemm_nt(x, w, shift=shift, relu=True), torch.relu(base + shift.double()))
    y, cs = gemm.gemm_nt(x, w, scale=scale, shift=shift, residual=res, relu=True, colsum=True)
    ref = torch.relu(base * scale.double() + shift.double() + res.double())
    _close(y, ref)
    _close(cs, ref.sum(0), 1e-5)
    # gate without relu
    _close(gemm.gemm_nt(x, w, residual=res, gate=res), (base + res.double()) * (res > 0))


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_nn_data_gradient_epilogues(M, N, K):
    from datr_amd import gemm
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(M * 3 + K)
    dy = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(K, N, device=dev, generator=g)
    y = torch.randn(M, N, device=dev, generator=g)
    r = torch.randn(M, N, device=dev, generator=g)
    base = dy.double() @ w.double()
    _close(gemm.gemm_nn(dy, w), base)
    dz, cs = gemm.gemm_nn(dy, w, gate=y, colsum=True)
    ref = base * (y > 0)
    _close(dz, ref)
    _close(cs, ref.sum(0), 1e-5)
    _close(gemm.gemm_nn(dy, w, residual=r, gate=y), (base + r.double()) * (y > 0))
    # bitwise reproducible (no atomics anywhere)
    dz2, cs2 = gemm.gemm_nn(dy, w, gate=y, colsum=True)
    assert torch.equal(dz, dz2) and torch.equal(cs, cs2)


@pytest.mark.parametrize("P,M,N", [(1, 4, 4), (31, 64, 64), (100, 128, 64), (1000, 256, 64), (4200, 512, 2048),
                                   (16800, 256, 1024), (66800, 128, 512), (5000, 36, 260)])
def test_tn_weight_gradient(P, M, N):
    from datr_amd import gemm
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(P + M)
    dy = torch.randn(P, M, device=dev, generator=g)
    x = torch.randn(P, N, device=dev, generator=g)
    rs = torch.rand(M, device=dev, generator=g) + 0.5
    ref = dy.double().t() @ x.double()
    dw = gemm.gemm_tn(dy, x)
    _close(dw, ref, 5e-6)
    dws = gemm.gemm_tn(dy, x, rowscale=rs)
    _close(dws, ref * rs.double()[:, None], 5e-6)
    assert torch.equal(dw, gemm.gemm_tn(dy, x))
    # the bias gradient beside the weight gradient: column sums of dy out of the A fragments
    dwb, db = gemm.gemm_tn(dy, x, bias_grad=True)
    assert torch.equal(dwb, dw)
    _close(db, dy.double().sum(0), 1e-5)
    assert torch.equal(db, gemm.gemm_tn(dy, x, bias_grad=True)[1])


def test_row_slices_and_strided_output():
    """Operands / outputs that are column slices of wider matrices (leading dimension > width)."""
    from datr_amd import gemm
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(5)
    big = torch.randn(300, 512, device=dev, generator=g)
    x = big[:, 128:384]                                  # [300, 256], ld 512
    w = torch.randn(192, 256, device=dev, generator=g)
    out_big = torch.zeros(300, 400, device=dev)
    out = out_big[:, 100:292]
    gemm.gemm_nt(x, w, out=out)
    _close(out, x.double() @ w.double().t())
    assert float(out_big[:, :100].abs().max()) == 0 and float(out_big[:, 292:].abs().max()) == 0
    dw = gemm.gemm_tn(x, big[:, :128])
    _close(dw, x.double().t() @ big[:, :128].double(), 5e-6)


def test_unsupported_shapes_raise():
    from datr_amd import gemm
    dev = _dev()
    x = torch.randn(10, 48, device=dev)                  # K % 32 != 0
    w = torch.randn(8, 48, device=dev)
    with pytest.raises(RuntimeError):
        gemm.gemm_nt(x, w)


@pytest.mark.parametrize("M,N,K", [(300, 64, 256), (4200, 512, 2048), (515, 2048, 256), (1000, 64, 128)])
def test_split_bf16_inner_product_is_fp32_grade(M, N, K, monkeypatch):
    """The experimental inner product (DATR_GEMM_SPLIT_BF16=1: operands split exactly into three bf16 pieces, six
    bf16-MFMA products, fp32 accumulation -- off by default) against float64 for the three forms, scaled by
    sum |a||b| (the natural error scale of a dot product): within 2x of the fp32-MFMA path's error, and both
    below 1e-6."""
    from datr_amd import gemm
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    x = torch.randn(M, K, device=dev, generator=g) * 3 + 0.5
    w = torch.randn(N, K, device=dev, generator=g)
    dy = torch.randn(M, N, device=dev, generator=g)

    def errors():
        out = []
        ref, sc = x.double() @ w.double().t(), x.double().abs() @ w.double().abs().t()
        out.append(((gemm.gemm_nt(x, w).double() - ref).abs() / sc).max().item())
        ref, sc = dy.double() @ w.double(), dy.double().abs() @ w.double().abs()
        out.append(((gemm.gemm_nn(dy, w).double() - ref).abs() / sc).max().item())
        ref, sc = dy.double().t() @ x.double(), dy.double().abs().t() @ x.double().abs()
        out.append(((gemm.gemm_tn(dy, x).double() - ref).abs() / sc).max().item())
        return out

    monkeypatch.delenv("DATR_GEMM_SPLIT_BF16", raising=False)
    exact = errors()
    monkeypatch.setenv("DATR_GEMM_SPLIT_BF16", "1")
    split = errors()
    for e, s_ in zip(exact, split):
        assert e < 1e-6 and s_ < 1e-6 and s_ < 2 * e + 1e-7, (exact, split)


def test_nan_travels_through_relu_and_gate_epilogues():
    """A NaN accumulator must come out NaN -- as through torch.relu / addmm (forward) and through
    threshold_backward under a NaN gate (backward) -- so that the loss guard of the training loop
    (/root/reference/engine.py:81-84) sees a diverged run; fmaxf / `gate > 0` would have returned 0."""
    from datr_amd import gemm
    dev = _dev()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(300, 64, generator=g).to(dev)
    w = torch.randn(96, 64, generator=g).to(dev)
    shift = torch.randn(96, generator=g).to(dev)
    x[7, 3] = float("nan")
    y = gemm.gemm_nt(x, w, shift=shift, relu=True)
    ref = torch.relu(x @ w.t() + shift)
    assert torch.isnan(y[7]).all() and torch.isnan(ref[7]).all()
    assert not torch.isnan(y[torch.arange(300, device=dev) != 7]).any()
    y2 = gemm.gemm_nt(x, w, shift=shift, relu=False)
    assert torch.isnan(y2[7]).all() and not torch.isinf(y2).any()
    # data gradient with a gate: dz = (dy @ w) * [h > 0] in torch's threshold_backward semantics (h <= 0 ? 0 : dz)
    dy = torch.randn(300, 96, generator=g).to(dev)
    h = torch.randn(300, 64, generator=g).to(dev)
    h[11, 5] = float("nan")
    dz = gemm.gemm_nn(dy, w, gate=h)
    ref = torch.where(h <= 0, torch.zeros((), device=dev), dy @ w)
    torch.testing.assert_close(dz, ref, rtol=1e-4, atol=1e-4, equal_nan=True)
    assert dz[11, 5] != 0 and not torch.isnan(dz[11, 5])     # the gradient passes a NaN gate (h <= 0 is false)
    dy[20, 0] = float("nan")
    dz = gemm.gemm_nn(dy, w, gate=h.abs() + 1.0)
    assert torch.isnan(dz[20]).all()
