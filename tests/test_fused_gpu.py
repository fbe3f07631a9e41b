"""GPU: fused frozen-BN (+residual) (+ReLU) kernel against the reference's op-by-op formula
(/root/reference/models/dino/backbone.py:62-72 followed by add / relu)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def reference(x, w, b, rm, rv, res, relu):
    scale = w * (rv + 1e-5).rsqrt()
    y = x * scale.view(1, -1, 1, 1) + (b - rm * scale).view(1, -1, 1, 1)
    if res is not None:
        y = y + res
    return torch.relu(y) if relu else y


@pytest.mark.parametrize("shape", [(2, 64, 50, 84), (1, 2048, 25, 42), (3, 7, 5, 3), (4, 256, 8, 8)])
@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("with_res", [True, False])
@pytest.mark.parametrize("nhwc", [False, True])
def test_frozen_bn_act(shape, relu, with_res, nhwc):
    from datr_amd.backbone import FrozenBatchNorm2d
    from datr_amd.fused import frozen_bn_act
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(sum(shape))
    C = shape[1]
    bn = FrozenBatchNorm2d(C)
    bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
    bn.bias.copy_(torch.randn(C, generator=g))
    bn.running_mean.copy_(torch.randn(C, generator=g))
    bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    bn.to(dev)
    fmt = torch.channels_last if nhwc else torch.contiguous_format
    x = torch.randn(shape, generator=g).to(dev).contiguous(memory_format=fmt).requires_grad_(True)
    res = (torch.randn(shape, generator=g).to(dev).contiguous(memory_format=fmt).requires_grad_(True)
           if with_res else None)
    go = torch.randn(shape, generator=g).to(dev)
    y = frozen_bn_act(x, *bn.scale_shift(), residual=res, relu=relu)
    y.backward(go)
    gx, gr = x.grad.clone(), None if res is None else res.grad.clone()
    x.grad = None
    if res is not None:
        res.grad = None
    yr = reference(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, res, relu)
    yr.backward(go)
    torch.testing.assert_close(y, yr, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(gx, x.grad, rtol=1e-6, atol=1e-6)
    if res is not None:
        torch.testing.assert_close(gr, res.grad, rtol=0, atol=0)


def test_scale_shift_cache_invalidates_on_load():
    from datr_amd.backbone import FrozenBatchNorm2d
    bn = FrozenBatchNorm2d(4).cuda()
    s1, _ = bn.scale_shift()
    assert bn.scale_shift()[0] is s1                       # cached
    bn.load_state_dict({"weight": torch.full((4,), 2.0), "bias": torch.zeros(4),
                        "running_mean": torch.zeros(4), "running_var": torch.ones(4)})
    s2, _ = bn.scale_shift()
    torch.testing.assert_close(s2, torch.full((4,), 2.0, device="cuda") / (1 + 1e-5) ** 0.5)


@pytest.mark.parametrize("rows,d,dff", [(1000, 256, 2048), (37, 64, 128), (2200, 256, 1024), (1, 32, 64)])
def test_ffn_relu_matches_autograd(rows, d, dff):
    """Fused FFN (bias+ReLU GEMM epilogue forward; in-place relu-backward + bias-gradient pass
    backward) vs the reference's op sequence linear2(relu(linear1(x)))
    (/root/reference/models/dino/deformable_transformer.py:803-806) under autograd."""
    from datr_amd.fused import ffn_relu
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(rows)
    l1, l2 = torch.nn.Linear(d, dff).to(dev), torch.nn.Linear(dff, d).to(dev)
    x = torch.randn(2, rows, d, generator=g).to(dev).requires_grad_(True)
    go = torch.randn(2, rows, d, generator=g).to(dev)
    y = ffn_relu(x, l1, l2)
    y.backward(go)
    got = [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in (*l1.parameters(), *l2.parameters())]
    x.grad = None
    for p in (*l1.parameters(), *l2.parameters()):
        p.grad = None
    yr = l2(torch.relu(l1(x)))
    yr.backward(go)
    ref = [yr.detach(), x.grad] + [p.grad for p in (*l1.parameters(), *l2.parameters())]
    for a, b in zip(got, ref):
        scale = max(float(b.abs().max()), 1e-6)
        assert float((a - b).abs().max()) <= 2e-5 * scale, (a.shape, float((a - b).abs().max()), scale)
    # bias gradient is reproducible bit for bit
    x.grad = None
    l1.bias.grad = None
    ffn_relu(x, l1, l2).backward(go)
    assert torch.equal(l1.bias.grad, got[3])


@pytest.mark.parametrize("shape", [(2, 1000, 256), (1100, 4, 256), (3, 256), (1, 1, 256)])
def test_add_layer_norm_matches_autograd(shape):
    """Fused residual add + LayerNorm (csrc/layernorm.hip) vs norm(x + res) under autograd
    (/root/reference/models/dino/deformable_transformer.py:796-806)."""
    from datr_amd.fused import add_layer_norm
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(shape[0])
    norm = torch.nn.LayerNorm(256).to(dev)
    with torch.no_grad():
        norm.weight.copy_(torch.rand(256, generator=g) + 0.5)
        norm.bias.copy_(torch.randn(256, generator=g))
    x = (torch.randn(shape, generator=g) * 2 + 0.3).to(dev).requires_grad_(True)
    r = torch.randn(shape, generator=g).to(dev).requires_grad_(True)
    go = torch.randn(shape, generator=g).to(dev)
    y = add_layer_norm(x, r, norm)
    y.backward(go)
    got = [y.detach().clone(), x.grad.clone(), r.grad.clone(), norm.weight.grad.clone(), norm.bias.grad.clone()]
    x.grad = r.grad = norm.weight.grad = norm.bias.grad = None
    yr = norm(x + r)
    yr.backward(go)
    ref = [yr.detach(), x.grad, r.grad, norm.weight.grad, norm.bias.grad]
    for a, b in zip(got, ref):
        scale = max(float(b.abs().max()), 1e-6)
        assert float((a - b).abs().max()) <= 1e-5 * scale + 1e-6, (a.shape, float((a - b).abs().max()), scale)
    # gamma / beta gradients are reproducible bit for bit
    x.grad = r.grad = norm.weight.grad = norm.bias.grad = None
    add_layer_norm(x, r, norm).backward(go)
    assert torch.equal(norm.weight.grad, got[3]) and torch.equal(norm.bias.grad, got[4])


@pytest.mark.parametrize("rows,cin,cout", [(4400, 256, 256), (88892, 256, 384), (7, 32, 4), (300, 64, 128)])
def test_fast_linear_matches_autograd(rows, cin, cout):
    """FastLinear: nn.Linear's forward / data-gradient GEMMs; weight + bias gradient from the library GEMM +
    column-sum kernel (few rows) or from the own deterministic split-K kernel whose A fragments give the
    bias gradient (csrc/gemm_f32.hip, >= 1 024 rows)."""
    from datr_amd.fused import FastLinear
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(rows)
    lin = FastLinear(cin, cout).to(dev)
    x = torch.randn(2, rows, cin, generator=g).to(dev).requires_grad_(True)
    go = torch.randn(2, rows, cout, generator=g).to(dev)
    y = lin(x)
    y.backward(go)
    got = [y.detach().clone(), x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone()]
    x.grad = lin.weight.grad = lin.bias.grad = None
    yr = torch.nn.functional.linear(x, lin.weight, lin.bias)
    yr.backward(go)
    ref = [yr.detach(), x.grad, lin.weight.grad, lin.bias.grad]
    from datr_amd import gemm
    if gemm.own_big(x.detach().reshape(-1, cin), lin.weight):
        # a row count the library's selections do not cover (or the "own" backend): forward and data gradient are the
        # own NT / NN forms -- the same products in another fp32 summation order
        for a, e in ((got[0], torch.nn.functional.linear(x.detach().double(), lin.weight.double(), lin.bias.double())),
                     (got[1], go.double() @ lin.weight.double())):
            assert float((a.double() - e).abs().max()) <= 4e-6 * float(e.abs().max())
    else:
        for a, b in zip(got[:2], ref[:2]):
            assert torch.equal(a, b)                                 # the same GEMM calls
    exact_w = (go.reshape(-1, cout).double().t() @ x.detach().reshape(-1, cin).double())
    wscale = float(exact_w.abs().max())
    assert float((got[2].double() - exact_w).abs().max()) <= max(5e-6 * wscale, float((ref[2].double() - exact_w).abs().max()) * 2)
    scale = max(float(ref[3].abs().max()), 1e-6)
    assert float((got[3] - ref[3]).abs().max()) <= 2e-5 * scale
    with torch.no_grad():
        # without autograd the library call from 16 384 rows on is the own NT form (fused.linear): same product,
        # another fp32 summation order
        yn = lin(x)
        if 2 * rows >= 16384:
            exact = torch.nn.functional.linear(x.detach().double(), lin.weight.double(), lin.bias.double())
            assert float((yn.double() - exact).abs().max()) <= 4e-6 * float(exact.abs().max())
        else:
            assert torch.equal(yn, yr.detach())


def test_fast_linear_takes_own_weight_gradient_kernel_and_matches_autograd():
    from datr_amd import fused
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    lin = fused.FastLinear(256, 256).to(dev)
    ref = torch.nn.Linear(256, 256).to(dev)
    ref.load_state_dict(lin.state_dict())
    x = torch.randn(2, 9000, 256, device=dev, requires_grad=True)
    xr = x.detach().clone().requires_grad_(True)
    gy = torch.randn(2, 9000, 256, device=dev)
    assert x.numel() // 256 >= fused.OWN_WGRAD_MIN_ROWS
    lin(x).backward(gy)
    ref(xr).backward(gy)
    torch.testing.assert_close(x.grad, xr.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(lin.weight.grad, ref.weight.grad, rtol=1e-4, atol=2e-3)
    torch.testing.assert_close(lin.bias.grad, ref.bias.grad, rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("nc", [2, 4])
def test_sine_embed_kernel_matches_torch_formulation(nc):
    """csrc/sine_embed.hip against the torch restatement of gen_sineembed_for_position
    (utils.py:138-163), which a tensor that requires grad still takes."""
    from datr_amd.transformer import gen_sineembed_for_position
    dev = torch.device("cuda:0")
    torch.manual_seed(nc)
    pos = torch.rand(1100, 4, nc, device=dev) * 1.2 - 0.1
    fused = gen_sineembed_for_position(pos)
    ref = gen_sineembed_for_position(pos.clone().requires_grad_(True)).detach()
    assert fused.shape == (1100, 4, 128 * nc)
    torch.testing.assert_close(fused, ref, rtol=0, atol=2e-6)


def test_cached_proposals_equal_uncached():
    from datr_amd.transformer import gen_encoder_output_proposals
    dev = torch.device("cuda:0")
    shapes = [(20, 27), (10, 14), (5, 7), (3, 4)]
    S = sum(h * w for h, w in shapes)
    torch.manual_seed(0)
    mem = torch.randn(2, S, 256, device=dev)
    mask = torch.zeros(2, S, dtype=torch.bool, device=dev)
    m0, p0 = gen_encoder_output_proposals(mem, mask, shapes)
    for _ in range(2):                                     # second call hits the cache
        m1, p1 = gen_encoder_output_proposals(mem, mask, shapes, no_padding=True)
        assert torch.equal(m0, m1) and torch.equal(p0, p1)


@pytest.mark.parametrize("L", [300, 301, 1100])
def test_decoder_self_attention_matches_nn_multihead_attention(L):
    """transformer._self_attention (merged q/k projection) against nn.MultiheadAttention with
    query = key = tgt + pos, value = tgt and the DN attention mask: outputs and every gradient."""
    from datr_amd.transformer import _plain_mha, _self_attention
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    N, E = 3, 256
    mha = torch.nn.MultiheadAttention(E, 8, dropout=0.0).to(dev)
    assert _plain_mha(mha)
    tgt = torch.randn(L, N, E, device=dev)
    pos = torch.randn(L, N, E, device=dev)
    mask = torch.zeros(L, L, dtype=torch.bool, device=dev)
    mask[100:, :100] = True
    mask[:50, 50:100] = True
    mask[50:100, :50] = True
    go = torch.randn(L, N, E, device=dev)
    outs, grads = [], []
    for fused_path in (True, False):
        t = tgt.clone().requires_grad_(True)
        p = pos.clone().requires_grad_(True)
        mha.zero_grad()
        q = t + p
        y = (_self_attention(mha, q, t, mask) if fused_path
             else mha(q, q, t, attn_mask=mask, need_weights=False)[0])
        y.backward(go)
        outs.append(y.detach())
        grads.append([t.grad, p.grad] + [x.grad.clone() for x in mha.parameters()])
    torch.testing.assert_close(outs[0], outs[1], rtol=1e-4, atol=1e-5)
    for a, b in zip(*grads):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("shape", [(2, 256, 25, 42), (3, 256, 13, 21), (1, 256, 1, 3), (2, 64, 40, 52)])
def test_groupnorm_nhwc_matches_float64_group_norm(shape):
    """csrc/groupnorm.hip behind fused.GroupNormNHWC against F.group_norm in float64 (output, input
    gradient, gamma / beta gradients) on inputs with a large common offset (mean >> std, the case the
    shifted sums exist for); output and gradient stay channels_last; NCHW inputs take nn.GroupNorm."""
    from datr_amd.fused import GroupNormNHWC
    dev = torch.device("cuda:0")
    torch.manual_seed(sum(shape))
    N, C, H, W = shape
    gn = GroupNormNHWC(32 if C % 128 == 0 else 16, C).to(dev)
    with torch.no_grad():
        gn.weight.uniform_(0.5, 1.5)
        gn.bias.normal_(0, 0.3)
    x = (torch.randn(shape, device=dev) * 0.7 + 11.0).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    go = torch.randn(shape, device=dev).contiguous(memory_format=torch.channels_last)
    y = gn(x)
    assert y.is_contiguous(memory_format=torch.channels_last)
    gx, gw, gb = torch.autograd.grad(y, (x, gn.weight, gn.bias), go)
    assert gx.is_contiguous(memory_format=torch.channels_last)
    xd = x.detach().double().contiguous().requires_grad_(True)
    wd, bd = gn.weight.detach().double().requires_grad_(True), gn.bias.detach().double().requires_grad_(True)
    ref = torch.nn.functional.group_norm(xd, gn.num_groups, wd, bd, gn.eps)
    rx, rw, rb = torch.autograd.grad(ref, (xd, wd, bd), go.double())
    torch.testing.assert_close(y.double(), ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(gx.double(), rx, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(gw.double(), rw, rtol=1e-4, atol=1e-4 * max(1.0, float(rw.abs().max())))
    torch.testing.assert_close(gb.double(), rb, rtol=1e-4, atol=1e-4 * max(1.0, float(rb.abs().max())))
    # an NCHW input is nn.GroupNorm's business (and gives an NCHW output)
    assert gn(x.detach().contiguous()).is_contiguous()


@pytest.mark.parametrize("L,masked", [(64, False), (300, True), (301, True), (1100, True), (1100, False)])
def test_attention_d32_forward_and_backward_match_float64_math(L, masked):
    """csrc/mha_fwd.hip + csrc/mha_bwd.hip against softmax(q k^T / sqrt(32) + mask) v written out in
    float64 under autograd: q and k are column slices of one merged buffer (the decoder's layout),
    the gradient arrives non-contiguous; two backward runs are bitwise identical (no atomics)."""
    from datr_amd.fused import attention_d32
    dev = torch.device("cuda:0")
    torch.manual_seed(L)
    N, H = 3, 8
    E = 32 * H
    qk = torch.randn(L, N, 2 * E, device=dev).requires_grad_(True)
    v = torch.randn(L, N, E, device=dev).requires_grad_(True)
    mask = None
    if masked:
        mask = torch.zeros(L, L, device=dev)
        mask[100:, :100] = float("-inf")
        mask[:50, 50:100] = float("-inf")
        mask[50:100, :50] = float("-inf")
        mask += 0.1 * torch.randn(L, L, device=dev)            # a finite additive part as well
    go = torch.randn(N, L, E, device=dev).transpose(0, 1)      # non-contiguous incoming gradient
    q, k = qk.split(E, dim=-1)
    out = attention_d32(q, k, v, mask, H)
    g_qk, g_v = torch.autograd.grad(out, (qk, v), go, retain_graph=True)
    g_qk2, g_v2 = torch.autograd.grad(out, (qk, v), go)
    assert torch.equal(g_qk, g_qk2) and torch.equal(g_v, g_v2)

    qd = qk.detach().double().requires_grad_(True)
    vd = v.detach().double().requires_grad_(True)
    q4, k4 = (x.reshape(L, N, H, 32).permute(1, 2, 0, 3) for x in qd.split(E, dim=-1))
    v4 = vd.reshape(L, N, H, 32).permute(1, 2, 0, 3)
    s = q4 @ k4.transpose(-1, -2) / 32 ** 0.5
    if mask is not None:
        s = s + mask.double()
    ref = (s.softmax(-1) @ v4).permute(2, 0, 1, 3).reshape(L, N, E)
    r_qk, r_v = torch.autograd.grad(ref, (qd, vd), go.double())
    torch.testing.assert_close(out.double(), ref, rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(g_qk.double(), r_qk, rtol=1e-5, atol=5e-6)
    torch.testing.assert_close(g_v.double(), r_v, rtol=1e-5, atol=5e-6)


@pytest.mark.parametrize("batch_major", [False, True])
def test_attention_qk_d32_equals_attention_d32_in_both_memory_orders(batch_major):
    """The merged-projection entry (one qk tensor in, one gradient out) runs the same kernels as attention_d32:
    bitwise the same output and gradients, for sequence-major memory and for [L, N, .] views of batch-major
    tensors (the decoder's batch-first layout); output and gradients come in the memory order of qk."""
    from datr_amd.fused import attention_d32, attention_qk_d32
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    L, N, H = 300, 4, 8
    E = 32 * H
    mask = 0.1 * torch.randn(L, L, device=dev)
    mask[100:, :100] = float("-inf")
    if batch_major:
        qk = torch.randn(N, L, 2 * E, device=dev).transpose(0, 1).requires_grad_(True)
        v = torch.randn(N, L, E, device=dev).transpose(0, 1).requires_grad_(True)
        go = torch.randn(N, L, E, device=dev).transpose(0, 1)
    else:
        qk = torch.randn(L, N, 2 * E, device=dev).requires_grad_(True)
        v = torch.randn(L, N, E, device=dev).requires_grad_(True)
        go = torch.randn(L, N, E, device=dev)
    out = attention_qk_d32(qk, v, mask, H)
    g_qk, g_v = torch.autograd.grad(out, (qk, v), go)
    assert out.stride() == v.stride() and g_qk.stride() == qk.stride() and g_v.stride() == v.stride()
    q, k = qk.split(E, dim=-1)
    ref = attention_d32(q, k, v, mask, H)
    r_qk, r_v = torch.autograd.grad(ref, (qk, v), go)
    assert torch.equal(out, ref) and torch.equal(g_qk, r_qk) and torch.equal(g_v, r_v)


def test_decoder_layer_batch_first_equals_sequence_first(monkeypatch):
    """DeformableTransformerDecoderLayer with batch_first=True ([bs, nq, C] in and out) against the reference
    order ([nq, bs, C]): the same kernels on the same rows -- outputs and parameter gradients agree to fp32
    GEMM rounding (the library may pick another kernel for another row order).  The FFN's hidden activation of the
    first run is handed to the second one (after checking that the second run computed the same values to
    rounding): both runs then open the same ReLUs -- a hidden unit within rounding of zero would otherwise gate
    differently under two GEMM row orders and move the gradients behind it by a finite amount."""
    from datr_amd import fused
    from datr_amd.transformer import DeformableTransformerDecoderLayer
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    shapes = [(20, 27), (10, 14), (5, 7), (3, 4)]
    S = sum(h * w for h, w in shapes)
    spatial = torch.tensor(shapes, dtype=torch.int64, device=dev)
    lsi = torch.cat([spatial.new_zeros(1), (spatial[:, 0] * spatial[:, 1]).cumsum(0)[:-1]])
    layer = DeformableTransformerDecoderLayer(256, 512, 0.0, "relu", 4, 8, 4).to(dev)
    bs, nq = 3, 130
    tgt = torch.randn(bs, nq, 256, device=dev)
    pos = torch.randn(bs, nq, 256, device=dev)
    ref = torch.rand(bs, nq, 4, 4, device=dev) * 0.6 + 0.2
    ref[..., 2:] *= 0.3
    memory = torch.randn(bs, S, 256, device=dev)
    mask = torch.zeros(nq, nq, device=dev)
    mask[30:, :30] = float("-inf")
    go = torch.randn(bs, nq, 256, device=dev)
    res, hidden = [], []
    real_hidden = fused._ffn_hidden

    def shared_hidden(x2, w1, b1):
        h = real_hidden(x2, w1, b1)
        if not hidden:                                     # batch-first run: rows (b, q)
            hidden.append(h)
            return h
        h1 = hidden[0].view(bs, nq, -1).transpose(0, 1).reshape(nq * bs, -1).contiguous()     # rows (q, b)
        torch.testing.assert_close(h, h1, rtol=1e-4, atol=1e-5)
        return h1
    monkeypatch.setattr(fused, "_ffn_hidden", shared_hidden)
    for bf in (True, False):
        t = tgt.clone().requires_grad_(True)
        tr = (lambda x: x) if bf else (lambda x: x.transpose(0, 1))
        out = layer(tgt=tr(t), tgt_query_pos=tr(pos), tgt_reference_points=tr(ref).contiguous(),
                    memory=memory.transpose(0, 1), memory_level_start_index=lsi, memory_spatial_shapes=spatial,
                    self_attn_mask=mask, batch_first=bf)
        out = tr(out)
        grads = torch.autograd.grad(out, [t] + list(layer.parameters()), go)
        res.append((out.detach(), grads))
    assert len(hidden) == 1, "the layer's FFN must run through fused._ffn_hidden in both orders"
    (o1, g1), (o2, g2) = res
    torch.testing.assert_close(o1, o2, rtol=1e-4, atol=1e-5)
    for a, b in zip(g1, g2):
        torch.testing.assert_close(a, b, rtol=1e-3, atol=2e-4 * max(1.0, float(b.abs().max())))


@pytest.mark.parametrize("rows,classes", [(1, 9), (4099, 9), (777, 16), (300, 1)])
def test_layer_norm_class_max_matches_float64(rows, classes):
    """csrc/layernorm.hip::ln_class_max_kernel: max_c(head(norm(x))) per row against float64."""
    from datr_amd.fused import layer_norm_class_max
    dev = torch.device("cuda:0")
    torch.manual_seed(rows)
    x = torch.randn(rows, 256, device=dev) * 3 + 0.5
    norm = torch.nn.LayerNorm(256).to(dev)
    head = torch.nn.Linear(256, classes).to(dev)
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
        norm.bias.normal_()
        head.bias.normal_()
    got = layer_norm_class_max(x.view(1, rows, 256), norm, head)
    assert got.shape == (1, rows)
    ref = torch.nn.functional.linear(
        torch.nn.functional.layer_norm(x.double(), (256,), norm.weight.double(), norm.bias.double(), norm.eps),
        head.weight.double(), head.bias.double()).max(-1)[0]
    torch.testing.assert_close(got.view(-1).double(), ref, rtol=1e-5, atol=2e-6)
    assert layer_norm_class_max(x, norm, torch.nn.Linear(256, 17).to(dev)) is None      # too many classes: caller's path
    # masked rows are evaluated on the fill vector
    mask = torch.rand(rows, device=dev) < 0.3
    fill = torch.randn(256, device=dev)
    got_m = layer_norm_class_max(x, norm, head, row_mask=mask, row_fill=fill)
    xm = torch.where(mask[:, None], fill[None, :], x)
    torch.testing.assert_close(got_m, layer_norm_class_max(xm, norm, head), rtol=0, atol=0)


def test_fused_contrast_loss_matches_the_op_sequence(monkeypatch):
    """csrc/prototypes.hip::contrast_loss_kernel against loss_contrast_da's torch ops (normalize, two products,
    cross entropy against eye * class_map): value and both gradients, with absent classes (zero prototypes, mask 0)."""
    from datr_amd import criterion as crit
    dev = torch.device("cuda:0")
    torch.manual_seed(4)
    K = 9
    res = []
    for fused_path in (True, False):
        monkeypatch.setattr(crit, "FUSED_CONTRAST", fused_path)
        torch.manual_seed(4)
        q_s = torch.randn(K, 256, device=dev)
        q_t = torch.randn(K, 256, device=dev) * 3
        m_s = (torch.rand(K, device=dev) > 0.3).float()
        m_t = (torch.rand(K, device=dev) > 0.3).float()
        q_s[m_s == 0] = 0                                  # an absent class has a zero prototype
        q_t[m_t == 0] = 0
        q_s.requires_grad_(True)
        q_t.requires_grad_(True)
        g = torch.randn(K, 256, device=dev)
        g[5] = 0
        out = {"output_source": q_s, "outputs_target": q_t, "query_mask_source": m_s, "query_mask_target": m_t,
               "global_proto": g}
        loss = crit.SetCriterion.loss_contrast_da(None, out)
        gs, gt = torch.autograd.grad(loss * 1.7, (q_s, q_t))
        res.append((loss.detach(), gs, gt))
    for a, b in zip(*res):
        torch.testing.assert_close(a, b, rtol=2e-5, atol=1e-6)


def test_class_prototypes_kernel_matches_the_op_sequence(monkeypatch):
    """csrc/prototypes.hip against the torch op sequence of get_prototype_class_wise (DA_utils.py:82-120): labels /
    present / counts / one-hot exactly, class means and the running global prototypes to fp32 rounding (the
    reference's GEMM fixes no summation order), the gradient to the features likewise; two successive calls (the
    source and the target half of a step) with some classes absent."""
    from datr_amd import domain
    dev = torch.device("cuda:0")
    torch.manual_seed(21)
    K = 9
    res = []
    for own in (True, False):
        monkeypatch.setattr(domain, "OWN_PROTOTYPES", own)
        torch.manual_seed(21)
        g_proto, g_amount = torch.zeros(K, 256, device=dev), torch.zeros(K, device=dev)
        outs = []
        for call in range(2):
            q = torch.randn(2, 900, 256, device=dev, requires_grad=True)
            logits = torch.randn(2, 900, K, device=dev)
            logits[..., 7] = -50.0                        # a class no query takes
            if call == 0:
                logits[..., 2] = -50.0
            p, present, g_proto, g_amount, onehot = domain.get_prototype_class_wise(q, logits, K, g_proto, g_amount)
            w = torch.randn(K, 256, device=dev)
            (gq,) = torch.autograd.grad((p * w).sum(), q)
            outs.append((p.detach(), present, g_proto, g_amount, onehot, gq))
        res.append(outs)
    for a, b in zip(*res):
        assert torch.equal(a[1], b[1]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
        for i in (0, 2, 5):
            torch.testing.assert_close(a[i], b[i], rtol=1e-5, atol=1e-6)
        assert not a[2].requires_grad


@pytest.mark.parametrize("scaled", [False, True])
def test_stack_linear_matches_torch_cat_and_scale(scaled):
    """csrc/stack_linear.hip: [diag(s) wa ; wb], [s * ba ; bb] and the gradients of the four parameters against the
    torch formulation (two multiplies + two concatenations) -- products of two floats, so bitwise equal."""
    from datr_amd.msda import stack_linear
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    wa, ba = torch.randn(256, 256, device=dev, requires_grad=True), torch.randn(256, device=dev, requires_grad=True)
    wb, bb = torch.randn(128, 256, device=dev, requires_grad=True), torch.randn(128, device=dev, requires_grad=True)
    s_ = (torch.rand(256, device=dev) + 0.01) if scaled else None
    gw, gb = torch.randn(384, 256, device=dev), torch.randn(384, device=dev)
    w, b = stack_linear(wa, ba, wb, bb, s_)
    got = torch.autograd.grad([w, b], [wa, ba, wb, bb], [gw, gb])
    wa2, ba2 = (wa * s_[:, None], ba * s_) if scaled else (wa, ba)
    w_ref, b_ref = torch.cat([wa2, wb], 0), torch.cat([ba2, bb], 0)
    ref = torch.autograd.grad([w_ref, b_ref], [wa, ba, wb, bb], [gw, gb])
    assert torch.equal(w, w_ref) and torch.equal(b, b_ref)
    for a, r in zip(got, ref):
        assert torch.equal(a, r)


@pytest.mark.parametrize("ref_dim,Lq", [(2, 701), (4, 300)])
def test_fused_sampling_prologue_matches_torch_ops(ref_dim, Lq):
    """csrc/msda_prologue.hip (softmax + sampling locations from the merged query projection,
    forward and backward) against the torch op sequence of MSDeformAttn.forward
    (ms_deform_attn.py:96-117) inside the whole module: output and all gradients."""
    from datr_amd import msda
    dev = torch.device("cuda:0")
    torch.manual_seed(ref_dim)
    shapes = [(20, 27), (10, 14), (5, 7), (3, 4)]
    S = sum(h * w for h, w in shapes)
    spatial = torch.tensor(shapes, dtype=torch.int64, device=dev)
    lsi = torch.cat([spatial.new_zeros(1), (spatial[:, 0] * spatial[:, 1]).cumsum(0)[:-1]])
    mod = msda.MSDeformAttn(256, 4, 8, 4).to(dev)
    with torch.no_grad():
        mod.sampling_offsets.weight.normal_(0, 0.02)
        mod.attention_weights.weight.normal_(0, 0.05)
    N = 2
    if ref_dim == 2:
        Lq = S
    query = torch.randn(N, Lq, 256, device=dev)
    src = torch.randn(N, S, 256, device=dev)
    ref = torch.rand(N, Lq, 4, ref_dim, device=dev) * 0.8 + 0.1
    if ref_dim == 4:
        ref[..., 2:] *= 0.3
    go = torch.randn(N, Lq, 256, device=dev)
    res = []
    for fused_path in (True, False):
        msda.FUSED_PROLOGUE = fused_path
        try:
            q = query.clone().requires_grad_(True)
            x = src.clone().requires_grad_(True)
            mod.zero_grad()
            y = mod(q, ref, x, spatial, lsi, None)
            y.backward(go)
            res.append([y.detach(), q.grad, x.grad] + [p.grad.clone() for p in mod.parameters()])
        finally:
            msda.FUSED_PROLOGUE = True
    for a, b in zip(*res):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-5 * max(1.0, float(b.abs().max())))


def test_plain_layer_norm_through_the_fused_kernel_matches_autograd():
    """fused.layer_norm (csrc/layernorm.hip with a null residual: enc_output_norm, decoder norm)
    against nn.LayerNorm under autograd."""
    from datr_amd.fused import layer_norm
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    ln = torch.nn.LayerNorm(256).to(dev)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5)
        ln.bias.normal_(0, 0.2)
    x = (torch.randn(3, 701, 256, device=dev) * 2 + 0.5).requires_grad_(True)
    go = torch.randn_like(x)
    got = torch.autograd.grad(layer_norm(x, ln), (x, ln.weight, ln.bias), go)
    y_ref = ln(x)
    ref = torch.autograd.grad(y_ref, (x, ln.weight, ln.bias), go)
    torch.testing.assert_close(layer_norm(x, ln), y_ref, rtol=1e-5, atol=1e-5)
    for a, b in zip(got, ref):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4 * max(1.0, float(b.abs().max())))


def test_level_embedding_add_matches_the_per_level_formula():
    """transformer._level_positions (one cat, one broadcast add, column-sum backward) against the
    reference's per-level `pos.flatten(2).transpose(1, 2) + level_embed[lvl]` and cat
    (deformable_transformer.py:283-286): identical values, level_embed gradient to float32 rounding;
    the flattened table is reused only for the very same embedding tensors."""
    from datr_amd import transformer as T
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    from tests.helpers import build_model
    tr = build_model()[1].transformer.to(dev)
    sizes = [(13, 21), (7, 11), (4, 6), (2, 3)]
    pos = [torch.randn(3, 256, h, w, device=dev) for h, w in sizes]
    go = torch.randn(3, sum(h * w for h, w in sizes), 256, device=dev)
    for no_padding in (True, False):
        tr.no_padding = no_padding
        out = tr._level_positions(pos)
        ref = torch.cat([p.flatten(2).transpose(1, 2) + tr.level_embed[l].view(1, 1, -1) for l, p in enumerate(pos)], 1)
        assert torch.equal(out, ref)
        (g,) = torch.autograd.grad(out, tr.level_embed, go)
        (gr,) = torch.autograd.grad(ref, tr.level_embed, go)
        torch.testing.assert_close(g, gr, rtol=1e-5, atol=1e-4)
    tr.no_padding = True
    a = tr._level_positions(pos)
    pos2 = [p.clone() + 1 for p in pos]
    b = tr._level_positions(pos2)                                   # other tensors: no stale table
    assert torch.equal(b, torch.cat([p.flatten(2).transpose(1, 2) + tr.level_embed[l].view(1, 1, -1)
                                     for l, p in enumerate(pos2)], 1))
    assert torch.equal(tr._level_positions(pos), a)


@pytest.mark.parametrize("rows", [(3, 700), (2, 4400)])
def test_ffn_block_as_one_node_matches_composition_and_float64(rows):
    """fused._FFNAddNorm (FFN + residual + LayerNorm as one autograd node, the residual gradient
    riding in the last GEMM's beta term) against the composition of the two fused halves and against
    the reference's op sequence in float64 (deformable_transformer.py:803-806)."""
    import torch.nn.functional as F
    from datr_amd import transformer as T
    dev = torch.device("cuda:0")
    torch.manual_seed(rows[1])
    lin1, lin2, norm = torch.nn.Linear(256, 2048).to(dev), torch.nn.Linear(2048, 256).to(dev), torch.nn.LayerNorm(256).to(dev)
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
        norm.bias.normal_(0, 0.1)
    drop = torch.nn.Dropout(0.0)
    x = torch.randn(*rows, 256, device=dev, requires_grad=True)
    go = torch.randn(*rows, 256, device=dev)
    params = [lin1.weight, lin1.bias, lin2.weight, lin2.bias, norm.weight, norm.bias]
    outs = {}
    for one in (True, False):
        T.FUSED_FFN_BLOCK = one
        y = T._ffn_block(x, lin1, F.relu, drop, lin2, drop, norm)
        outs[one] = (y, torch.autograd.grad(y, [x] + params, go))
    T.FUSED_FFN_BLOCK = True
    assert "FFNAddNorm" in type(outs[True][0].grad_fn).__name__
    assert torch.equal(outs[True][0], outs[False][0])                      # same kernels forward
    xd = x.detach().double().requires_grad_(True)
    pd = [p.detach().double().requires_grad_(True) for p in params]
    yd = F.layer_norm(xd + F.linear(F.relu(F.linear(xd, pd[0], pd[1])), pd[2], pd[3]), (256,), pd[4], pd[5])
    gd = torch.autograd.grad(yd, [xd] + pd, go.double())
    for a, b, r in zip(outs[True][1], outs[False][1], gd):
        scale = float(r.abs().max())
        # against float64 a handful of ReLU gates sit within float32 rounding of zero and flip (each taking the 256 values of its row along): isolated
        # outliers, not a tolerance; against the composed float32 path (same h) there are none
        off = (a.double() - r).abs() > 2e-5 * scale + 1e-6
        assert float(off.double().mean()) <= 2e-3 and float((a.double() - r).abs().max()) <= 0.05 * scale
        assert float((a - b).abs().max()) <= 2e-5 * scale + 1e-6


@pytest.mark.parametrize("n", [2, 3, 6, 8])
@pytest.mark.parametrize("shape,transposed", [((4, 1111, 256), False), ((3, 5, 7), False), ((4, 300, 256), True)])
def test_fan_out_sums_the_consumers_gradients_in_argument_order(n, shape, transposed):
    """fused.fan_out: n aliases of a tensor whose gradients meet in one csrc/addn.hip pass -- the same
    sum, in the same order, as autograd's pairwise accumulation (deformable_transformer.py:796-806: src
    feeds the query, the value projection and the residual)."""
    from datr_amd.fused import fan_out
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(n)
    x = torch.randn(shape, generator=g).to(dev).requires_grad_(True)
    ws = [torch.randn(shape, generator=g).to(dev) for _ in range(n)]
    xin = x.transpose(0, 1) if transposed else x
    parts = fan_out(xin, n)
    assert all(p.data_ptr() == xin.data_ptr() for p in parts)
    sum((p.transpose(0, 1) if transposed else p).mul(w).sum() for p, w in zip(parts, ws)).backward()
    expect = ws[0].clone()
    for w in ws[1:]:
        expect = expect + w                     # argument order
    assert torch.equal(x.grad, expect)


def test_fan_out_passes_through_without_gradient():
    from datr_amd.fused import fan_out
    x = torch.randn(5, 3, device="cuda:0")
    a, b = fan_out(x, 2)
    assert a is x and b is x


@pytest.mark.parametrize("nhwc", [False, True])
@pytest.mark.parametrize("with_res", [True, False])
def test_frozen_bn_act_two_handles_sum_their_gradients_in_the_kernel(nhwc, with_res):
    """frozen_bn_act(..., twice=True): the output as two handles (next bottleneck's conv1 + identity branch,
    torchvision Bottleneck.forward); datr_affine_act_backward2_f32 adds their gradients on the fly -- same
    result as one handle used twice."""
    from datr_amd.fused import frozen_bn_act
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    fmt = torch.channels_last if nhwc else torch.contiguous_format
    x0 = torch.randn(2, 64, 9, 11, generator=g).to(dev).contiguous(memory_format=fmt)
    r0 = torch.randn(2, 64, 9, 11, generator=g).to(dev).contiguous(memory_format=fmt)
    scale = (torch.rand(64, generator=g) + 0.5).to(dev)
    shift = torch.randn(64, generator=g).to(dev)
    wa, wb = (torch.randn(2, 64, 9, 11, generator=g).to(dev) for _ in range(2))

    def run(twice):
        x, r = x0.clone().requires_grad_(True), r0.clone().requires_grad_(with_res)
        out = frozen_bn_act(x, scale, shift, residual=r if with_res else None, relu=True, twice=twice)
        a, b = out if twice else (out, out)
        ((a * wa).sum() + (b * wb).sum()).backward()
        return a.detach(), x.grad, (r.grad if with_res else None)

    y1, gx1, gr1 = run(False)
    y2, gx2, gr2 = run(True)
    assert torch.equal(y1, y2)
    torch.testing.assert_close(gx1, gx2, rtol=1e-6, atol=1e-6)
    if with_res:
        torch.testing.assert_close(gr1, gr2, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("shape,ref_grad", [((6, 2, 1100, 4), True), ((1100, 4, 4), False), ((3, 5), True)])
def test_refine_boxes_matches_the_reference_op_sequence(shape, ref_grad):
    """sigmoid(delta + inverse_sigmoid(ref)) in one launch (csrc/refine.hip) against the reference's ops
    (/root/reference/models/dino/deformable_transformer.py:738-744, util/misc.py:587-591) in float64 under
    autograd, with reference values outside [0, 1], on the clamp thresholds and at the eps bounds."""
    from datr_amd.fused import refine_boxes
    from datr_amd.nested import inverse_sigmoid
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(sum(shape))
    delta = torch.randn(*shape, generator=g).to(dev).requires_grad_(True)
    ref = (torch.rand(*shape, generator=g) * 1.2 - 0.1)
    ref.view(-1)[:6] = torch.tensor([0.0, 1.0, 1e-3, 1 - 1e-3, 5e-4, 0.9997])
    ref = ref.to(dev).requires_grad_(ref_grad)
    go = torch.randn(*shape, generator=g).to(dev)
    out = refine_boxes(delta, ref)
    out.backward(go)
    d64, r64 = delta.detach().double().requires_grad_(True), ref.detach().double().requires_grad_(ref_grad)
    exp = (d64 + inverse_sigmoid(r64)).sigmoid()
    exp.backward(go.double())
    torch.testing.assert_close(out.double(), exp, rtol=1e-6, atol=2e-7)                    # float32 kernel vs float64
    torch.testing.assert_close(delta.grad.double(), d64.grad, rtol=1e-5, atol=1e-6)
    if ref_grad:
        # 1 - x near 1e-3 carries a relative float32 rounding error of ~6e-5 (x itself is rounded at 6e-8)
        scale = float(r64.grad.abs().max())
        torch.testing.assert_close(ref.grad.double(), r64.grad, rtol=3e-4, atol=1e-6 * max(scale, 1.0))


@pytest.mark.parametrize("rows,cin,cout", [(4400, 256, 256), (37, 64, 128), (13200, 256, 256)])
def test_linear_relu_matches_autograd(rows, cin, cout):
    """relu(linear(x)) as one node (bias + ReLU epilogue; gate pass + weight / bias gradient backward)."""
    from datr_amd.fused import linear_relu
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(rows)
    lin = torch.nn.Linear(cin, cout).to(dev)
    x = torch.randn(rows, cin, generator=g).to(dev).requires_grad_(True)
    go = torch.randn(rows, cout, generator=g).to(dev)
    y = linear_relu(x, lin.weight, lin.bias)
    y.backward(go)
    got = [y.detach().clone(), x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone()]
    x.grad = lin.weight.grad = lin.bias.grad = None
    yr = torch.relu(torch.nn.functional.linear(x, lin.weight, lin.bias))
    yr.backward(go)
    assert torch.equal(got[0], yr.detach())
    for a, b in zip(got[1:], [x.grad, lin.weight.grad, lin.bias.grad]):
        scale = max(float(b.abs().max()), 1e-6)
        assert float((a - b).abs().max()) <= 3e-5 * scale


@pytest.mark.parametrize("rows,live", [((2, 1111), (1, 1, 1)), ((4, 700), (1, 0, 1)), ((1, 260), (0, 1, 0))])
def test_ffn_block_with_next_query_equals_the_separate_ops_bitwise(rows, live):
    """fused._FFNAddNorm with `next_pos`: the node hands out the next encoder layer's (query = output + pos, value
    input, residual) handles -- deformable_transformer.py:789-798 -- and sums their gradients as it loads them.
    Against the plain node followed by fan_out and the ATen add: the same values bit for bit, forward and backward
    (gradient order (query + value) + residual, as csrc/addn.hip sums), with any subset of the handles used."""
    import torch.nn.functional as F
    from datr_amd import transformer as T
    from datr_amd.fused import fan_out
    dev = torch.device("cuda:0")
    torch.manual_seed(rows[1])
    lin1, lin2, norm = torch.nn.Linear(256, 1024).to(dev), torch.nn.Linear(1024, 256).to(dev), torch.nn.LayerNorm(256).to(dev)
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
        norm.bias.normal_(0, 0.1)
    drop = torch.nn.Dropout(0.0)
    x = torch.randn(*rows, 256, device=dev, requires_grad=True)
    pos = torch.randn(*rows, 256, device=dev, requires_grad=True)
    gs = [torch.randn(*rows, 256, device=dev) for _ in range(3)]
    params = [lin1.weight, lin1.bias, lin2.weight, lin2.bias, norm.weight, norm.bias]
    res = {}
    for fusedq in (True, False):
        if fusedq:
            q, v, r = T._ffn_block(x, lin1, F.relu, drop, lin2, drop, norm, next_pos=pos)
            assert "FFNAddNorm" in type(q.grad_fn).__name__ and v.data_ptr() == r.data_ptr()
        else:
            y = T._ffn_block(x, lin1, F.relu, drop, lin2, drop, norm)
            s_q, v, r = fan_out(y, 3)
            q = s_q + pos
        loss = sum((t * g).sum() for t, g, on in zip((q, v, r), gs, live) if on)
        grads = torch.autograd.grad(loss, [x, pos] + params, allow_unused=True)
        res[fusedq] = ((q, v, r), grads)
    for a, b in zip(res[True][0], res[False][0]):
        assert torch.equal(a, b)
    for a, b in zip(res[True][1], res[False][1]):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a, b)
    assert (res[True][1][1] is None) == (not live[0])            # the position table's gradient = the query's


def test_encoder_chain_of_next_queries_equals_layers_with_separate_adds(monkeypatch):
    """TransformerEncoder.forward with the layers chained through the FFN node's three handles against the same
    encoder with DATR_FUSED_NEXT_QUERY off: outputs bitwise equal, every gradient (tokens, position table,
    parameters) equal to fp32 rounding."""
    from datr_amd import transformer as T
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    layer = T.DeformableTransformerEncoderLayer(256, 512, 0.0, "relu", 4, 8, 4)
    enc = T.TransformerEncoder(layer, 3, None, d_model=256).to(dev)
    shapes = [(20, 28), (10, 14), (5, 7), (3, 4)]
    S = sum(h * w for h, w in shapes)
    spatial = torch.tensor(shapes, device=dev)
    lsi = torch.cat([spatial.new_zeros(1), spatial.prod(1).cumsum(0)[:-1]])
    src = torch.randn(2, S, 256, device=dev, requires_grad=True)
    pos = torch.randn(2, S, 256, device=dev, requires_grad=True)
    vr = torch.ones(2, 4, 2, device=dev)
    go = torch.randn(2, S, 256, device=dev)
    out = {}
    for on in (True, False):
        monkeypatch.setattr(T, "FUSED_NEXT_QUERY", on)
        y = enc(src, pos, spatial, lsi, vr, None, shapes_list=shapes)[0]
        out[on] = (y, torch.autograd.grad(y, [src, pos] + list(enc.parameters()), go))
    assert torch.equal(out[True][0], out[False][0])
    # (the MSDA backward's float atomics make grad_value reproducible to fp32 rounding only)
    for a, b in zip(out[True][1], out[False][1]):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-7
