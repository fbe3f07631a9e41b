"""csrc/wino.hip (Winograd F(2x2,3x3) on the MFMA units) and the discriminator path built on it,
against float64 convolutions under autograd (DA_utils.py:33-79 semantics)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _cl(x):
    return x.contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("cin,cout,sizes", [
    (8, 64, [(5, 7)]),
    (16, 64, [(16, 16), (17, 33)]),
    (128, 128, [(25, 42), (13, 21), (1, 1), (2, 3)]),
    (256, 128, [(50, 84), (7, 11)]),
])
def test_wino_conv_forward_and_data_gradient_match_float64(cin, cout, sizes):
    """Forward (scale, shift, LeakyReLU) and the transposed-filter call (gate, out_scale) of
    wino_conv3x3 vs F.conv2d / its autograd data gradient in float64; ragged sizes exercise the zero
    padding and the partial tiles, several levels share one launch."""
    from datr_amd.domain import wino_conv3x3, wino_filter
    dev = torch.device("cuda:0")
    torch.manual_seed(cin + cout)
    N = 2
    w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
    shift = torch.randn(cout, device=dev)
    scale = torch.rand(cout, device=dev) + 0.5
    xs = [_cl(torch.randn(N, cin, h, ww, device=dev)) for h, ww in sizes]
    ys = wino_conv3x3(xs, wino_filter(w), cout, shift=shift, scale=scale, slope=0.2)
    for x, y in zip(xs, ys):
        assert y.is_contiguous(memory_format=torch.channels_last)
        ref = F.leaky_relu(F.conv2d(x.double(), w.double(), padding=1) * scale.double().view(1, -1, 1, 1)
                           + shift.double().view(1, -1, 1, 1), 0.2)
        torch.testing.assert_close(y.double(), ref, rtol=2e-5, atol=2e-5)
    # data gradient with a gate and the reversal sign: needs cin % 64 == 0 as the gradient's Cout
    if cin % 64 == 0:
        dys = [_cl(torch.randn(N, cout, h, ww, device=dev)) for h, ww in sizes]
        gates = [_cl(torch.randn(N, cin, h, ww, device=dev)) for h, ww in sizes]
        dxs = wino_conv3x3(dys, wino_filter(w, True), cin, gates=gates, gate_slope=0.2, out_scale=-1.0)
        for x, dy, g, dx in zip(xs, dys, gates, dxs):
            xd = x.double().requires_grad_(True)
            (gx,) = torch.autograd.grad(F.conv2d(xd, w.double(), padding=1), xd, dy.double())
            ref = -torch.where(g.double() > 0, gx, gx * 0.2)
            torch.testing.assert_close(dx.double(), ref, rtol=2e-5, atol=2e-5)


def test_wino_filter_handles_channels_last_weights():
    from datr_amd.domain import wino_filter
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    w = torch.randn(64, 16, 3, 3, device=dev)
    for dg in (False, True):
        if dg and w.shape[1] % 64:
            continue
        assert torch.equal(wino_filter(w, dg), wino_filter(_cl(w), dg))


@pytest.mark.parametrize("sizes", [[(20, 27), (10, 14), (5, 7), (3, 4)], [(33, 18)]])
def test_discriminator_pyramid_matches_library_convolutions(sizes, monkeypatch):
    """FCDiscriminator_img.reversed_pyramid (GRL + 4 convs, own Winograd path) against the same
    module on float64 library convolutions under autograd: logits, every parameter gradient and the
    (reversed) input gradients; the input gradients of two backward runs are bitwise identical."""
    from datr_amd import domain
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    d = domain.FCDiscriminator_img(256).to(dev)
    xs = [torch.randn(2, 256, h, w, device=dev, requires_grad=True) for h, w in sizes]
    gos = [torch.randn(2, 1, h, w, device=dev) for h, w in sizes]
    assert domain.OWN_D_IMG
    outs = d.reversed_pyramid(xs)
    params = list(d.parameters())
    grads = torch.autograd.grad(outs, params + xs, gos, retain_graph=True)
    grads2 = torch.autograd.grad(outs, params + xs, gos)
    for a, b in zip(grads[len(params):], grads2[len(params):]):    # own kernels: bitwise reproducible
        assert torch.equal(a, b)                                   # (the library's weight gradient is not)

    d64 = domain.FCDiscriminator_img(256).to(dev).double()
    d64.load_state_dict({k: v.double() for k, v in d.state_dict().items()})
    xs64 = [x.detach().double().requires_grad_(True) for x in xs]
    monkeypatch.setattr(domain, "OWN_D_IMG", False)
    outs64 = d64.reversed_pyramid(xs64)
    ref = torch.autograd.grad(outs64, list(d64.parameters()) + xs64, [g.double() for g in gos])
    for o, r in zip(outs, outs64):
        torch.testing.assert_close(o.double(), r, rtol=1e-4, atol=1e-4)
    for g, r in zip(grads, ref):
        torch.testing.assert_close(g.double(), r, rtol=1e-4, atol=1e-4 * max(1.0, float(r.abs().max())))


@pytest.mark.parametrize("c,hw,trainable", [(64, (40, 52), False), (128, (25, 42), True), (256, (13, 21), True)])
def test_conv3x3_bn_relu_matches_float64(c, hw, trainable, monkeypatch):
    """datr_amd.wino.conv3x3_bn_relu (conv2 + frozen BN + ReLU of a ResNet bottleneck as ONE
    Winograd/MFMA launch; backward = fused gate pass, own data gradient, library weight gradient)
    against float64 ops.  The reference gradient uses the ReLU mask of the kernel's own output, so a
    pre-activation within rounding of zero cannot flip the comparison."""
    from datr_amd import wino
    from datr_amd.wino import conv3x3_bn_relu
    monkeypatch.setattr(wino, "OWN_BACKBONE_3X3_MAX_CH", 4096)
    dev = torch.device("cuda:0")
    torch.manual_seed(c)
    w = (torch.randn(c, c, 3, 3, device=dev) / (3 * c ** 0.5)).contiguous(memory_format=torch.channels_last)
    w.requires_grad_(trainable)
    scale, shift = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1
    x = torch.randn(2, c, *hw, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(trainable)
    y = conv3x3_bn_relu(x, w, scale, shift)
    assert y is not None and y.is_contiguous(memory_format=torch.channels_last)
    z = F.conv2d(x.detach().double(), w.detach().double(), padding=1) * scale.double().view(1, -1, 1, 1) \
        + shift.double().view(1, -1, 1, 1)
    torch.testing.assert_close(y.detach().double(), z.clamp(min=0), rtol=2e-5, atol=2e-5)
    assert conv3x3_bn_relu(x.detach().contiguous(), w, scale, shift) is None          # NCHW: library path
    if not trainable:
        return
    go = torch.randn_like(y)
    gx, gw = torch.autograd.grad(y, (x, w), go)
    xd, wd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    zd = F.conv2d(xd, wd, padding=1)
    dz = go.double() * (y.detach() > 0) * scale.double().view(1, -1, 1, 1)
    rx, rw = torch.autograd.grad(zd, (xd, wd), dz)
    torch.testing.assert_close(gx.double(), rx, rtol=1e-4, atol=1e-4 * float(rx.abs().max()))
    torch.testing.assert_close(gw.double(), rw, rtol=1e-4, atol=1e-4 * float(rw.abs().max()))


def test_backbone_bottleneck_routes_conv2_through_the_own_kernel(monkeypatch):
    """A channels_last bottleneck gives the same output with conv2 + bn2 + relu on the own kernel as
    on the library convolution + fused frozen-BN pass, and really calls the kernel."""
    from datr_amd import backbone, wino
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    blk = backbone.Bottleneck(512, 128, 1, backbone.FrozenBatchNorm2d, downsample=False).to(dev)
    blk = blk.to(memory_format=torch.channels_last)
    x = torch.randn(2, 512, 25, 42, device=dev).contiguous(memory_format=torch.channels_last)
    calls = []
    real = wino.wino_conv3x3
    monkeypatch.setattr(wino, "wino_conv3x3", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    with torch.no_grad():
        y_own = blk(x)
        assert calls
        monkeypatch.setattr(wino, "OWN_BACKBONE_3X3", False)
        y_lib = blk(x)
    torch.testing.assert_close(y_own, y_lib, rtol=1e-4, atol=1e-4)


def test_wide_bottleneck_weight_gradient_in_the_winograd_domain(monkeypatch):
    """A layer3-sized bottleneck (256-channel conv2) on the per-op path: the weight gradient of conv2 comes
    from csrc/wino_wgrad.hip (one call) -- same gradients as the all-library block."""
    from datr_amd import backbone, wino
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    blk = backbone.Bottleneck(1024, 256, 1, backbone.FrozenBatchNorm2d, downsample=False).to(dev)
    blk = blk.to(memory_format=torch.channels_last)
    x = torch.randn(2, 1024, 25, 42, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    go = torch.randn(2, 1024, 25, 42, device=dev).contiguous(memory_format=torch.channels_last)
    calls = []
    real = wino.wino_wgrad
    monkeypatch.setattr(wino, "wino_wgrad", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    params = [blk.conv1.weight, blk.conv2.weight, blk.conv3.weight]

    def reference(acts):
        """The block's op sequence (torchvision's Bottleneck under FrozenBatchNorm2d, backbone.py:62-72) in
        float64 under autograd, its ReLUs opened where the float32 run's were (a pre-activation within rounding of
        zero gates differently in two evaluation orders and moves single gradient entries by O(1): not an error of
        either side, and not something an element-wise tolerance should have to absorb)."""
        def bn(m, t):
            sc, sh = m.scale_shift()
            return t * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
        xd = x.detach().double().requires_grad_(True)
        ws = [p.detach().double().requires_grad_(True) for p in params]
        o = bn(blk.bn1, F.conv2d(xd, ws[0])) * (acts[0] > 0)
        o = bn(blk.bn2, F.conv2d(o, ws[1], padding=1)) * (acts[1] > 0)
        o = (bn(blk.bn3, F.conv2d(o, ws[2])) + xd) * (acts[2] > 0)
        return o, torch.autograd.grad(o, [xd] + ws, go.double())

    for own_path in (True, False):
        monkeypatch.setattr(wino, "OWN_BACKBONE_3X3", own_path)
        acts = []
        monkeypatch.setattr(backbone, "_CAPTURE", acts)
        y = blk(x)
        got = torch.autograd.grad(y, [x] + params, go)
        monkeypatch.setattr(backbone, "_CAPTURE", None)
        assert len(calls) == 1                       # the own weight gradient ran in the first pass only
        yr, ref = reference(acts[0])
        torch.testing.assert_close(y.detach().double(), yr.detach(), rtol=1e-4, atol=1e-5 * float(yr.detach().abs().max()))
        for a, b in zip(got, ref):
            torch.testing.assert_close(a.double(), b, rtol=1e-4, atol=1e-4 * float(b.abs().max()))


@pytest.mark.parametrize("cin,cout,hw,relu,bias", [(64, 256, (40, 52), False, False), (256, 64, (25, 42), True, True),
                                                   (512, 256, (13, 21), False, True), (2048, 512, (7, 11), True, True)])
def test_conv1x1_as_gemm_matches_float64_convolution(cin, cout, hw, relu, bias, monkeypatch):
    """datr_amd.pointwise.conv1x1 (1x1 convolution of a channels_last tensor as a GEMM on the
    [pixels, C] view, bias / ReLU in the epilogue) against F.conv2d in float64: output, input, weight
    and bias gradients -- with the weight gradient once through the GEMM and once through the library
    convolution (the routing threshold is moved across the test size)."""
    from datr_amd import pointwise
    dev = torch.device("cuda:0")
    torch.manual_seed(cin + cout)
    x = torch.randn(2, cin, *hw, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5).requires_grad_(True)
    b = torch.randn(cout, device=dev).requires_grad_(True) if bias else None
    go = torch.randn(2, cout, *hw, device=dev).contiguous(memory_format=torch.channels_last)
    xd, wd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    bd = b.detach().double().requires_grad_(True) if bias else None
    ref = F.conv2d(xd, wd, bd)
    ref = ref.relu() if relu else ref
    rg = torch.autograd.grad(ref, [xd, wd] + ([bd] if bias else []), go.double())
    for max_px in (0, 1 << 30):
        monkeypatch.setattr(pointwise, "WGRAD_GEMM_MAX_PIXELS", max_px)
        y = pointwise.conv1x1(x, w, b, relu=relu)
        assert y is not None and y.is_contiguous(memory_format=torch.channels_last)
        g = torch.autograd.grad(y, [x, w] + ([b] if bias else []), go)
        torch.testing.assert_close(y.double(), ref, rtol=1e-4, atol=1e-4)
        for a, r in zip(g, rg):
            torch.testing.assert_close(a.double(), r, rtol=1e-4, atol=1e-4 * max(1.0, float(r.abs().max())))
    assert pointwise.conv1x1(x.detach().contiguous(), w, b, relu=relu) is None      # NCHW: library path


def test_fold_frozen_bn_folds_and_backpropagates():
    from datr_amd.pointwise import fold_frozen_bn
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    ws = [torch.randn(c, 8, 1, 1, device=dev, requires_grad=(i != 1)) for i, c in enumerate((4, 8, 12))]
    ss = [torch.rand(c, device=dev) + 0.5 for c in (4, 8, 12)]
    f = fold_frozen_bn(list(zip(ws, ss)))
    for w, s, o in zip(ws, ss, f):
        torch.testing.assert_close(o, w.detach().reshape(w.shape[0], -1) * s[:, None])
    (f[0].sum() * 2 + f[2].sum()).backward()
    torch.testing.assert_close(ws[0].grad, (2 * ss[0]).view(-1, 1, 1, 1).expand_as(ws[0]))
    torch.testing.assert_close(ws[2].grad, ss[2].view(-1, 1, 1, 1).expand_as(ws[2]))
    assert ws[1].grad is None and f[1] is fold_frozen_bn([(ws[1], ss[1])])[0]       # frozen: cached


# ---- weight gradient in the Winograd domain (csrc/wino_wgrad.hip) ----------------------------------------
def _wgrad_reference(xs, dys, w):
    """float64 weight gradient of conv2d(x, w, padding=1), summed over the levels (autograd, CPU)."""
    w64 = w.detach().double().cpu().requires_grad_(True)
    total = 0.0
    for x, dy in zip(xs, dys):
        y = F.conv2d(x.detach().double().cpu(), w64, padding=1)
        total = total + (y * dy.detach().double().cpu()).sum()
    (g,) = torch.autograd.grad(total, w64)
    return g


@pytest.mark.parametrize("cin,cout,n,sizes", [
    (64, 64, 1, [(8, 8)]),                       # one workgroup, tiles = one chunk per row
    (64, 128, 2, [(7, 9)]),                      # odd sizes: half tiles at the right / bottom edge
    (128, 64, 3, [(5, 3), (1, 1)]),              # tile rows shorter than a chunk, a 1-pixel level
    (256, 128, 2, [(25, 42), (13, 21), (7, 11), (4, 6)]),   # a pyramid, several segments per level
])
def test_wino_wgrad_matches_float64_autograd(cin, cout, n, sizes):
    from datr_amd.wino import wino_wgrad
    g = torch.Generator().manual_seed(cin + cout + n)
    w = torch.randn(cout, cin, 3, 3, generator=g).to("cuda:0")
    xs = [torch.randn(n, cin, h, ww, generator=g).to("cuda:0").contiguous(memory_format=torch.channels_last) for h, ww in sizes]
    dys = [torch.randn(n, cout, h, ww, generator=g).to("cuda:0").contiguous(memory_format=torch.channels_last) for h, ww in sizes]
    got = wino_wgrad(xs, dys, w)
    assert got.shape == w.shape and got.stride() == w.stride()
    want = _wgrad_reference(xs, dys, w)
    err = (got.double().cpu() - want).abs().max().item()
    assert err <= 2e-5 * want.abs().max().item() + 1e-5, err
    # deterministic: the split-K segments are added in a fixed order
    assert torch.equal(got, wino_wgrad(xs, dys, w))
    # channels_last weights (the backbone's): same values at the weight's own strides
    wcl = w.contiguous(memory_format=torch.channels_last)
    got_cl = wino_wgrad(xs, dys, wcl)
    assert got_cl.stride() == wcl.stride() and torch.equal(got_cl, got)


def test_wino_wgrad_accuracy_is_that_of_the_library_at_the_discriminator_shape():
    """256 -> 256 on the 50x84 level of 4 images: error against float64 no worse than 2x the
    library's fp32 weight-gradient convolution."""
    from datr_amd.wino import wino_wgrad
    g = torch.Generator().manual_seed(5)
    w = torch.randn(256, 256, 3, 3, generator=g).to("cuda:0")
    x = torch.randn(4, 256, 50, 84, generator=g).to("cuda:0").contiguous(memory_format=torch.channels_last)
    dy = (torch.randn(4, 256, 50, 84, generator=g) * 1e-3).to("cuda:0").contiguous(memory_format=torch.channels_last)
    got = wino_wgrad([x], [dy], w)
    _, lib, _ = torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                    [False, True, False])
    want = _wgrad_reference([x], [dy], w)
    e_own = (got.double().cpu() - want).abs().max().item()
    e_lib = (lib.double().cpu() - want).abs().max().item()
    assert e_own <= max(2 * e_lib, 1e-6 * want.abs().max().item()), (e_own, e_lib)


def test_wino_wgrad_refuses_other_channel_counts():
    from datr_amd.wino import wino_wgrad
    w = torch.zeros(48, 64, 3, 3, device="cuda:0")
    x = torch.zeros(1, 64, 8, 8, device="cuda:0").contiguous(memory_format=torch.channels_last)
    dy = torch.zeros(1, 48, 8, 8, device="cuda:0").contiguous(memory_format=torch.channels_last)
    assert wino_wgrad([x], [dy], w) is None


def test_wino_wgrad_full_size_pyramid_against_the_library():
    """The shape the training step runs: 256 -> 128 over the four pyramid levels of four images at
    1333x800 (one split-K launch, segments crossing level boundaries) against the library's
    weight-gradient convolution summed over the levels, and linearity in dY (a size-independent
    property: wgrad(x, a dy1 + dy2) = a wgrad(x, dy1) + wgrad(x, dy2))."""
    from datr_amd.wino import wino_wgrad
    g = torch.Generator().manual_seed(9)
    sizes = [(100, 167), (50, 84), (25, 42), (13, 21)]
    w = torch.randn(128, 256, 3, 3, generator=g).to("cuda:0")
    xs = [_cl(torch.randn(4, 256, h, ww, generator=g).to("cuda:0")) for h, ww in sizes]
    d1 = [_cl(torch.randn(4, 128, h, ww, generator=g).to("cuda:0")) for h, ww in sizes]
    d2 = [_cl(torch.randn(4, 128, h, ww, generator=g).to("cuda:0")) for h, ww in sizes]
    got = wino_wgrad(xs, d1, w)
    lib = None
    for x, dy in zip(xs, d1):
        _, gw, _ = torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                       [False, True, False])
        lib = gw if lib is None else lib + gw
    scale = float(lib.abs().max())
    assert float((got - lib).abs().max()) <= 2e-5 * scale
    mix = wino_wgrad(xs, [0.5 * a + b for a, b in zip(d1, d2)], w)
    assert float((mix - (0.5 * got + wino_wgrad(xs, d2, w))).abs().max()) <= 2e-5 * scale


@pytest.mark.parametrize("sizes", [[(100, 167), (50, 84), (25, 42), (13, 21)], [(1, 1), (2, 3), (7, 5)], [(33, 17)]])
def test_one_channel_classifier_matches_float64_autograd(sizes):
    """csrc/conv_cout1.hip -- the discriminator's 128 -> 1 classifier (DA_utils.py:67,78) on all pyramid
    levels: forward, the LeakyReLU-gated data gradient, weight and bias gradient summed over the
    levels, against F.conv2d under autograd in float64; 1x1 / ragged maps exercise the zero padding;
    two runs are bitwise equal (fixed-order reduction)."""
    from datr_amd.domain import classifier_backward, classifier_forward
    dev = torch.device("cuda:0")
    torch.manual_seed(len(sizes))
    N, C, slope = 2, 128, 0.2
    w = torch.randn(1, C, 3, 3, device=dev) / (3 * C ** 0.5)
    b = torch.randn(1, device=dev)
    zs = [torch.randn(N, C, h, ww, device=dev) for h, ww in sizes]            # conv3's pre-activations
    acts = [_cl(F.leaky_relu(z, slope)) for z in zs]
    outs = classifier_forward(acts, w, b)
    douts = [torch.randn(N, 1, h, ww, device=dev) for h, ww in sizes]
    dzs, dw, db = classifier_backward(acts, douts, w, slope)
    dzs2, dw2, db2 = classifier_backward(acts, douts, w, slope)
    assert torch.equal(dw, dw2) and torch.equal(db, db2) and all(torch.equal(a, b_) for a, b_ in zip(dzs, dzs2))
    wd, bd = w.double().requires_grad_(True), b.double().requires_grad_(True)
    zd = [z.double().requires_grad_(True) for z in zs]
    ref = [F.conv2d(F.leaky_relu(z, slope), wd, bd, padding=1) for z in zd]
    grads = torch.autograd.grad(ref, [wd, bd] + zd, [d.double() for d in douts])
    for o, r in zip(outs, ref):
        assert o.shape == r.shape
        torch.testing.assert_close(o.double(), r.detach(), rtol=1e-5, atol=1e-5)
    scale = float(grads[0].abs().max())
    torch.testing.assert_close(dw.double(), grads[0], rtol=1e-4, atol=1e-5 * scale)
    torch.testing.assert_close(db.double(), grads[1], rtol=1e-4, atol=1e-5 * float(grads[1].abs().max()))
    for dz, g in zip(dzs, grads[2:]):
        assert dz.is_contiguous(memory_format=torch.channels_last) or dz.shape[2] * dz.shape[3] == 1
        torch.testing.assert_close(dz.double(), g, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("co,ci", [(64, 64), (128, 256), (512, 512)])
def test_filter_pair_equals_the_two_separate_transforms(co, ci):
    """datr_wino_weights_pair_f32: the forward and the data-gradient filter of a weight from one launch (mirroring
    the taps permutes the transform's rows / columns): the forward one bitwise, the mirrored one up to the order of
    the three-term sums (w0 + w1 + w2 against w2 + w1 + w0)."""
    from datr_amd import wino
    dev = torch.device("cuda:0")
    torch.manual_seed(co + ci)
    w = torch.randn(co, ci, 3, 3, device=dev)
    u, uf = wino.wino_filter_pair(w)
    assert torch.equal(u, wino.wino_filter(w))
    torch.testing.assert_close(uf, wino.wino_filter(w, True), rtol=1e-6, atol=1e-6)
    wt = w.permute(1, 0, 2, 3)                      # a strided weight
    u2, uf2 = wino.wino_filter_pair(wt)
    assert torch.equal(u2, wino.wino_filter(wt))
    torch.testing.assert_close(uf2, wino.wino_filter(wt, True), rtol=1e-6, atol=1e-6)
