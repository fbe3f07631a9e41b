"""Shared test helpers: golden loaders and the oracle-backed MSDA stand-in used ONLY by the
CPU tests (the product package itself has no CPU path)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)


def load_npz(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def t(x):
    return torch.from_numpy(np.asarray(x))


def patch_msda_with_oracle(monkeypatch, kind="c"):
    """Route datr_amd.msda's two native entry points through the oracle so that the host
    logic (model, criterion, DDP) can be exercised on a CPU-only box.
    kind="c": oracle/msda_ref.c;  kind="grid_sample": the torch formulation, which is the
    arithmetic the golden vectors were produced with (the reference's CPU path)."""
    from datr_amd import msda
    from oracle import msda_oracle as O

    if kind == "c":
        def fwd(value, shapes, lsi, loc, attn, im2col_step):
            return O.msda_forward(value, shapes, lsi, loc, attn).to(value.device)

        def bwd(value, shapes, lsi, loc, attn, grad_output, im2col_step):
            return [g.to(value.device)
                    for g in O.msda_backward(value, shapes, lsi, loc, attn, grad_output)]
    else:
        def fwd(value, shapes, lsi, loc, attn, im2col_step):
            with torch.no_grad():
                return O.msda_grid_sample(value, shapes, loc, attn)

        def bwd(value, shapes, lsi, loc, attn, grad_output, im2col_step):
            with torch.enable_grad():
                v = value.detach().requires_grad_(True)
                s = loc.detach().requires_grad_(True)
                a = attn.detach().requires_grad_(True)
                out = O.msda_grid_sample(v, shapes, s, a)
                return list(torch.autograd.grad(out, (v, s, a), grad_output))

    monkeypatch.setattr(msda, "ms_deform_attn_forward", fwd)
    monkeypatch.setattr(msda, "ms_deform_attn_backward", bwd)
    # the fused focal-loss kernel has no CPU path either: stand in the reference's formulation
    from datr_amd import criterion
    from oracle import focal_oracle
    monkeypatch.setattr(criterion, "focal_loss_sums", focal_oracle.focal_sums_torch)


def build_model(device="cpu", task="c2f"):
    import synth
    from datr_amd import config
    from datr_amd.detector import build_dino
    args = {"c2f": config.c2f_args, "sim10k": config.sim10k_args}[task](device=device)
    torch.manual_seed(0)
    model, criterion, post = build_dino(args)
    synth.synth_init_(model)
    return args, model.to(device), criterion, post


def canonical_grad_norms(model):
    sd = model.state_dict(keep_vars=True)
    names_of = {}
    for k, v in sd.items():
        names_of.setdefault(v.data_ptr(), []).append(k)
    out = {}
    for names in names_of.values():
        p = sd[names[0]]
        if p.requires_grad:
            out[min(names)] = None if p.grad is None else float(p.grad.double().norm())
    return out


def force_reference_selection(model, golden, device):
    """Test-side substitution of DeformableTransformer.select_queries: return the reference's
    top-900 token indices (source pass first, then target pass) instead of re-deriving them."""
    calls = [t(golden["topk_source"]).to(device), t(golden["topk_target"]).to(device)]
    state = {"i": 0}

    def forced(scores):
        if scores.shape[0] == calls[0].shape[0] + calls[1].shape[0]:   # merged decoder pass
            return torch.cat(calls, 0)
        idx = calls[state["i"] % 2]
        state["i"] += 1
        return idx
    model.transformer.select_queries = forced


def run_training_step(model, criterion, device, golden, channels_last=False, num_classes=9):
    """One training forward + criterion + backward with the golden's CDN noise injected.
    channels_last: the images (and whatever the caller did to the model) in NHWC -- bench.py's layout."""
    import synth
    from datr_amd.nested import nested_tensor_from_tensor_list
    imgs, targets = synth.synth_batch(num_classes=num_classes)
    samples = nested_tensor_from_tensor_list([i.to(device) for i in imgs])
    if channels_last:
        samples.tensors = samples.tensors.contiguous(memory_format=torch.channels_last)
    targets = [{k: v.to(device) for k, v in tg.items()} for tg in targets]
    model.train()
    criterion.train()
    model.dn_noise_override = {
        "label_p": t(golden["noise_label_p"]), "new_label": t(golden["noise_new_label"]),
        "rand_sign": t(golden["noise_rand_sign"]), "rand_part": t(golden["noise_rand_part"])}
    out = model(samples, targets)
    loss_dict, indices_list = criterion(out, targets, return_indices=True)
    wd = criterion.weight_dict
    total = sum(loss_dict[k] * wd[k] for k in loss_dict if k in wd)
    model.zero_grad()
    total.backward()
    return out, loss_dict, indices_list, total


def check_training_step(model, out, loss_dict, indices_list, total, g, logit_tol=1e-3,
                        loss_rtol=2e-3):
    """Compare against tests/golden/model_step.npz.  Tolerances: BASELINE north_star --
    logits within 1e-3 fp32, index selection bit-exact."""
    cpu = lambda x: x.detach().float().cpu()
    close = lambda a, b, **kw: torch.testing.assert_close(cpu(a), t(b), **kw)
    kw = dict(rtol=logit_tol, atol=logit_tol)
    close(out["pred_logits"], g["pred_logits"], **kw)
    close(out["pred_boxes"], g["pred_boxes"], **kw)
    close(torch.stack([a["pred_logits"] for a in out["aux_outputs"]]), g["aux_logits"], **kw)
    close(torch.stack([a["pred_boxes"] for a in out["aux_outputs"]]), g["aux_boxes"], **kw)
    close(out["interm_outputs"]["pred_logits"], g["interm_logits"], **kw)
    close(out["interm_outputs"]["pred_boxes"], g["interm_boxes"], **kw)
    close(out["interm_outputs_for_matching_pre"]["pred_boxes"], g["init_box_proposal"], **kw)
    known = out["dn_meta"]["output_known_lbs_bboxes"]
    assert out["dn_meta"]["pad_size"] == int(g["dn_pad_size"])
    assert out["dn_meta"]["num_dn_group"] == int(g["dn_groups"])
    close(known["pred_logits"], g["dn_logits"], **kw)
    close(known["pred_boxes"], g["dn_boxes"], **kw)
    da = out["da_output"]
    close(da["backbone_DA"], g["backbone_DA"], **kw)
    close(da["proto_DA"]["da_protos"], g["da_protos"], **kw)
    assert torch.equal(cpu(da["proto_DA"]["class_map_source"]), t(g["class_map_source"]))
    assert torch.equal(cpu(da["proto_DA"]["class_map_target"]), t(g["class_map_target"]))
    close(da["global_proto_DA"]["output_source"], g["proto_source"], **kw)
    close(da["global_proto_DA"]["outputs_target"], g["proto_target"], **kw)
    close(model.global_proto, g["global_proto"], **kw)
    assert torch.equal(cpu(model.Amount), t(g["Amount"]))
    # Hungarian indices of all 7 matcher calls: bit-exact
    mine = np.stack([np.stack([np.stack([s.cpu().numpy(), tt.cpu().numpy()]) for s, tt in call])
                     for call in indices_list])
    assert mine.shape == g["indices"].shape
    assert (mine == g["indices"]).all(), "Hungarian assignment differs from the reference"
    # loss dict: same keys in the same order, values to 1e-3 relative
    assert list(loss_dict.keys()) == [str(k) for k in g["loss_keys"]]
    mine_vals = torch.tensor([float(v.detach()) for v in loss_dict.values()], dtype=torch.float64)
    torch.testing.assert_close(mine_vals, t(g["loss_values"]), rtol=loss_rtol, atol=loss_rtol * 0.1)
    torch.testing.assert_close(float(total), float(g["total_loss"]), rtol=loss_rtol, atol=1e-3)


def check_gradients(model, g, rtol=2e-2, outlier_fraction=0.0):
    """Per-parameter gradient norms and a few full gradients against the golden step.
    outlier_fraction > 0: bilinear sampling is piecewise linear in the sampling location, so its
    gradient JUMPS where a location crosses a pixel boundary; a last-bit difference in a location
    (other GEMM / convolution kernels upstream) can put one sample on the other side, which moves a
    handful of elements of the encoder-side gradients by a few per cent (observed run to run:
    tools/probes/nhwc_repro.py).  Then up to that fraction of a tensor's elements may miss the
    element-wise tolerance, by no more than 15 % of the tensor's largest element."""
    norms = canonical_grad_norms(model)
    assert sorted(norms) == sorted(str(k) for k in g["grad_keys"])
    assert all(v is not None for v in norms.values()), "a trainable parameter got no gradient"
    ref = {str(k): float(v) for k, v in zip(g["grad_keys"], g["grad_norms"])}
    bad = [(k, norms[k], ref[k]) for k in ref
           if abs(norms[k] - ref[k]) > rtol * abs(ref[k]) + 1e-5]
    assert not bad, bad[:10]
    sd = model.state_dict(keep_vars=True)
    for key in g:
        if key.startswith("grad::"):
            p = sd[key[6:]]
            mine = p.grad if p.grad.numel() < 5000 else p.grad.flatten()[:5000]
            scale = float(np.abs(g[key]).max()) + 1e-12
            mine, want = mine.detach().float().cpu(), t(g[key])
            if outlier_fraction > 0:
                err = (mine - want).abs()
                miss = err > 2e-2 * want.abs() + 2e-3 * scale
                assert float(miss.float().mean()) <= outlier_fraction and float(err.max()) <= 0.15 * scale, \
                    (key, float(miss.float().mean()), float(err.max()), scale)
                continue
            torch.testing.assert_close(mine, want, rtol=2e-2, atol=2e-3 * scale)


DA_LOSS_KEYS = ("loss_backbone_DA", "loss_proto_DA", "loss_global_proto_DA")


def run_source_only_step(model, criterion, device, golden, channels_last=False):
    """BASELINE configs 1-2 (SURVEY.md 8d "Mapping BASELINE configs"): the non-reference switch
    `model.domain_adaptation = False` -- no D_img / prototypes / target pass, B source images only.
    The source-side outputs and every non-DA loss of the reference's step do not depend on the DA
    branch in the forward pass, so the golden step pins them.  The source image of the synthetic
    pair is the larger one: alone in a batch it is unpadded exactly as inside the pair."""
    import synth
    from datr_amd.nested import nested_tensor_from_tensor_list
    imgs, targets = synth.synth_batch()
    samples = nested_tensor_from_tensor_list([imgs[0].to(device)])
    if channels_last:
        samples.tensors = samples.tensors.contiguous(memory_format=torch.channels_last)
    targets = [{k: v.to(device) for k, v in tg.items()} for tg in targets]
    model.domain_adaptation = False
    model.train()
    criterion.train()
    model.dn_noise_override = {
        "label_p": t(golden["noise_label_p"]), "new_label": t(golden["noise_new_label"]),
        "rand_sign": t(golden["noise_rand_sign"]), "rand_part": t(golden["noise_rand_part"])}
    src_idx = t(golden["topk_source"]).to(device)
    model.transformer.select_queries = lambda scores: src_idx
    out = model(samples, targets)
    loss_dict, indices_list = criterion(out, targets, return_indices=True)
    wd = criterion.weight_dict
    total = sum(loss_dict[k] * wd[k] for k in loss_dict if k in wd)
    model.zero_grad()
    total.backward()
    return out, loss_dict, indices_list, total


def check_source_side(out, loss_dict, indices_list, g, logit_tol=1e-3, loss_rtol=2e-3):
    """Source-side outputs, Hungarian indices and all non-DA losses against model_step.npz."""
    cpu = lambda x: x.detach().float().cpu()
    close = lambda a, b, **kw: torch.testing.assert_close(cpu(a), t(b), **kw)
    kw = dict(rtol=logit_tol, atol=logit_tol)
    assert "da_output" not in out
    close(out["pred_logits"], g["pred_logits"], **kw)
    close(out["pred_boxes"], g["pred_boxes"], **kw)
    close(torch.stack([a["pred_logits"] for a in out["aux_outputs"]]), g["aux_logits"], **kw)
    close(torch.stack([a["pred_boxes"] for a in out["aux_outputs"]]), g["aux_boxes"], **kw)
    close(out["interm_outputs"]["pred_logits"], g["interm_logits"], **kw)
    close(out["interm_outputs"]["pred_boxes"], g["interm_boxes"], **kw)
    close(out["interm_outputs_for_matching_pre"]["pred_boxes"], g["init_box_proposal"], **kw)
    known = out["dn_meta"]["output_known_lbs_bboxes"]
    assert out["dn_meta"]["pad_size"] == int(g["dn_pad_size"])
    close(known["pred_logits"], g["dn_logits"], **kw)
    close(known["pred_boxes"], g["dn_boxes"], **kw)
    mine = np.stack([np.stack([np.stack([s.cpu().numpy(), tt.cpu().numpy()]) for s, tt in call])
                     for call in indices_list])
    assert mine.shape == g["indices"].shape and (mine == g["indices"]).all()
    ref_keys = [str(k) for k in g["loss_keys"]]
    want_keys = [k for k in ref_keys if k not in DA_LOSS_KEYS]
    assert list(loss_dict.keys()) == want_keys and len(want_keys) == 79
    ref = dict(zip(ref_keys, g["loss_values"].tolist()))
    mine_vals = torch.tensor([float(loss_dict[k].detach()) for k in want_keys], dtype=torch.float64)
    want_vals = torch.tensor([ref[k] for k in want_keys], dtype=torch.float64)
    torch.testing.assert_close(mine_vals, want_vals, rtol=loss_rtol, atol=loss_rtol * 0.1)
