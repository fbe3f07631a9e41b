"""GPU: the assembled model on the HIP path (libdatr_hip.so MSDA kernels + ROCm kernels for
the dense layers) against the golden training step captured from the reference.

Index selection downstream of `topk(900)` over the 1700 encoder tokens is discontinuous: two
tokens whose scores differ by less than the fp32 noise between two correct implementations
(different GEMM summation orders on CPU vs GPU) swap ranks, and because DINO pairs the k-th
selected box with the k-th learned content query (embed_init_tgt) the swapped queries are
genuinely different (SURVEY.md 7.2 "Bit-exact index selection with 1e-3 logits").  So:
  * everything UPSTREAM of the selection is compared element-wise (1e-3),
  * selection, matching and post-processing are pinned as functions on identical inputs
    (tests/test_model_cpu.py: bit-exact against the reference),
  * downstream of it we require the bulk of the queries to agree to 1e-3 and the loss dict to
    agree to a few percent, and two GPU runs to agree bit-for-bit in the forward pass.
"""
import numpy as np
import pytest
import torch

from helpers import build_model, canonical_grad_norms, load_npz, run_training_step, t

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def step():
    dev = torch.device("cuda:0")
    g = load_npz("model_step.npz")
    _, model, criterion, _ = build_model("cuda:0")
    out, loss_dict, indices_list, total = run_training_step(model, criterion, dev, g)
    return g, model, out, loss_dict, indices_list, total


def test_upstream_of_selection_matches_reference(step):
    g, model, out, *_ = step
    # the image-level discriminator sees backbone+input_proj features of all images: no topk
    torch.testing.assert_close(out["da_output"]["backbone_DA"].float().cpu(), t(g["backbone_DA"]),
                               rtol=1e-3, atol=1e-3)


def test_bulk_of_queries_and_losses_match_reference(step):
    g, model, out, loss_dict, indices_list, total = step
    assert list(loss_dict.keys()) == [str(k) for k in g["loss_keys"]] and len(loss_dict) == 82
    diff = (out["pred_logits"].float().cpu() - t(g["pred_logits"])).abs().amax(-1)[0]
    frac = float((diff < 1e-3).float().mean())
    assert frac > 0.80, f"only {frac:.2%} of the 900 queries agree with the reference to 1e-3"
    dn = out["dn_meta"]["output_known_lbs_bboxes"]
    diff_dn = (dn["pred_logits"].float().cpu() - t(g["dn_logits"])).abs().amax(-1)[0]
    assert float((diff_dn < 2e-3).float().mean()) > 0.80
    mine = {k: float(v.detach()) for k, v in loss_dict.items()}
    ref = {str(k): float(v) for k, v in zip(g["loss_keys"], g["loss_values"])}
    for k in ("loss_ce", "loss_bbox", "loss_giou", "loss_ce_dn", "loss_bbox_dn", "loss_giou_dn",
              "loss_ce_interm", "loss_backbone_DA", "loss_proto_DA", "loss_global_proto_DA"):
        assert abs(mine[k] - ref[k]) <= 0.05 * abs(ref[k]) + 1e-3, (k, mine[k], ref[k])
    assert abs(float(total) - float(g["total_loss"])) <= 0.02 * float(g["total_loss"])
    assert int(out["dn_meta"]["pad_size"]) == int(g["dn_pad_size"])


def test_every_trainable_parameter_gets_a_finite_gradient(step):
    g, model, *_ = step
    norms = canonical_grad_norms(model)
    assert sorted(norms) == sorted(str(k) for k in g["grad_keys"])
    assert all(v is not None and np.isfinite(v) for v in norms.values())
    ref = {str(k): float(v) for k, v in zip(g["grad_keys"], g["grad_norms"])}
    # aggregate agreement: backbone / encoder gradients are upstream-dominated
    for k in ("backbone.0.body.layer2.0.conv1.weight", "input_proj.0.0.weight",
              "D_img.conv1.weight", "transformer.encoder.layers.0.linear1.weight"):
        assert abs(norms[k] - ref[k]) <= 0.1 * ref[k] + 1e-6, (k, norms[k], ref[k])


def test_forward_is_bitwise_reproducible_on_gpu():
    dev = torch.device("cuda:0")
    g = load_npz("model_step.npz")
    outs = []
    for _ in range(2):
        _, model, criterion, _ = build_model("cuda:0")
        out, loss_dict, indices_list, _ = run_training_step(model, criterion, dev, g)
        outs.append((out["pred_logits"].detach().clone(), out["pred_boxes"].detach().clone(),
                     indices_list))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for a, b in zip(outs[0][2], outs[1][2]):
        for (s1, t1), (s2, t2) in zip(a, b):
            assert torch.equal(s1, s2) and torch.equal(t1, t2)
