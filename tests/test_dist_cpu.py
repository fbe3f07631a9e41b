"""CPU, world_size 2 over gloo: the flat-bucket gradient all-reduce (datr_amd/dist.py) and the
criterion's num_boxes all-reduce, run as real processes (torch.multiprocessing.spawn)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(8, 16)
        self.shared = nn.Linear(16, 16)
        self.heads = nn.ModuleList([self.shared for _ in range(3)])     # aliased, like DINO's heads
        self.unused = nn.Linear(16, 4)                                  # never in the graph
        self.sometimes = nn.Linear(16, 4)                               # only on rank 0

    def forward(self, x, use_sometimes):
        h = torch.relu(self.a(x))
        for m in self.heads:
            h = m(h)
        out = h.sum()
        if use_sometimes:
            out = out + self.sometimes(h).sum()
        return out


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from datr_amd.dist import GradAllReducer, init_distributed
    init_distributed(backend="gloo")
    torch.manual_seed(0)
    model = Toy()
    ref = Toy()
    ref.load_state_dict(model.state_dict())
    # tiny buckets so that several are exercised
    red = GradAllReducer(model, bucket_mb=0.0005, first_bucket_mb=0.0002)
    assert len(red.buckets) >= 3
    assert sum(len(b.params) for b in red.buckets) == len(list(model.parameters()))
    for step in range(3):
        x = torch.randn(5, 8, generator=torch.Generator().manual_seed(100 * step + rank))
        red.zero_grad()
        model(x, use_sometimes=(rank == 0)).backward()
        red.finish()
        # expected: average over ranks of the local gradients (missing grads count as zeros)
        ref.zero_grad()
        ref(x, use_sometimes=(rank == 0)).backward()
        for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            local = torch.zeros_like(q) if q.grad is None else q.grad.clone()
            dist.all_reduce(local)
            torch.testing.assert_close(p.grad, local / world, rtol=1e-6, atol=1e-7, msg=n)
        # grads are still views of the flat buffers (no copy-out happened)
        for b in red.buckets:
            for p in b.params:
                assert p.grad.untyped_storage().data_ptr() == b.flat.untyped_storage().data_ptr()
    # criterion: num_boxes is the world average of the per-rank counts, clamped at 1
    from datr_amd import criterion as crit_mod
    from datr_amd.criterion import SetCriterion
    from oracle import focal_oracle
    crit_mod.focal_loss_sums = focal_oracle.focal_sums_torch      # no GPU in this test

    class M(nn.Module):
        def forward(self, outputs, targets):
            return [(torch.arange(len(t["labels"])), torch.arange(len(t["labels"]))) for t in targets]

        def forward_many(self, outs, targets):
            return [self.forward(o, targets) for o in outs]
    crit = SetCriterion(3, M(), {}, 0.25, ["labels", "boxes", "cardinality"])
    n_gt = 1 if rank == 0 else 3
    out = {"pred_logits": torch.zeros(1, 5, 3), "pred_boxes": torch.full((1, 5, 4), 0.5),
           "dn_meta": None}
    tg = [{"labels": torch.ones(n_gt, dtype=torch.long), "boxes": torch.full((n_gt, 4), 0.25)}]
    crit.eval()
    losses = crit(out, tg)
    # L1 = |0.5-0.25| * 4 * n_gt / num_boxes with num_boxes = (1+3)/2 = 2
    torch.testing.assert_close(losses["loss_bbox"], torch.tensor(1.0 * n_gt / 2.0))
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")


def test_flat_bucket_allreduce_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_reducer_single_process_views_and_unused():
    from datr_amd.dist import GradAllReducer
    torch.manual_seed(0)
    model = Toy()
    red = GradAllReducer(model, bucket_mb=0.0005, first_bucket_mb=0.0002)
    red.zero_grad()
    model(torch.randn(3, 8), use_sometimes=False).backward()
    red.finish()
    assert torch.count_nonzero(model.unused.weight.grad) == 0
    assert torch.count_nonzero(model.sometimes.weight.grad) == 0
    assert torch.count_nonzero(model.shared.weight.grad) > 0
    names = [n for n, _ in model.named_parameters()]
    assert [n for n, u in zip(names, red.used_flags().tolist()) if not u] == \
        ["unused.weight", "unused.bias", "sometimes.weight", "sometimes.bias"]
    # every gradient view starts on a 256-byte boundary of its bucket (the multi-tensor optimizer /
    # norm kernels vectorise only for aligned pointers), also behind the odd-sized [4] biases
    for b in red.buckets:
        for v in b.views:
            assert (v.data_ptr() - b.flat.data_ptr()) % 256 == 0
            assert b.flat.data_ptr() <= v.data_ptr() and v.data_ptr() + v.numel() * 4 <= b.flat.data_ptr() + b.numel * 4
    ref = Toy()
    ref.load_state_dict(model.state_dict())
    torch.manual_seed(0)
    Toy()                                                          # same generator position as above
    ref(torch.randn(3, 8), use_sometimes=False).backward()
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        if q.grad is not None:
            assert torch.equal(p.grad, q.grad), n
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2)
    before = model.a.weight.detach().clone()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1)
    opt.step()
    assert not torch.equal(before, model.a.weight)


def test_reducer_keeps_channels_last_parameter_layout():
    """A conv weight in torch.channels_last (bench.py's NHWC backbone) must get a gradient view
    with the SAME strides: fused AdamW rejects mismatched layouts, and autograd then accumulates
    the NHWC weight gradient in place.  Values equal the plain (no reducer) gradients."""
    from datr_amd.dist import GradAllReducer
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(4, 6, 3, padding=1), torch.nn.ReLU(),
                              torch.nn.Conv2d(6, 2, 1)).to(memory_format=torch.channels_last)
    x = torch.randn(2, 4, 5, 7).contiguous(memory_format=torch.channels_last)
    net(x).square().sum().backward()
    ref = [p.grad.clone() for p in net.parameters()]
    for p in net.parameters():
        p.grad = None
    red = GradAllReducer(net, bucket_mb=0.0005, first_bucket_mb=0.0002)
    for _ in range(2):                      # second round: views survive zero_grad()
        red.zero_grad()
        net(x).square().sum().backward()
        red.finish()
        for p, g in zip(net.parameters(), ref):
            assert p.grad.stride() == p.stride() and p.grad.shape == p.shape
            torch.testing.assert_close(p.grad, g)
    w = net[0].weight
    assert w.is_contiguous(memory_format=torch.channels_last) and not w.is_contiguous()
    assert any(b.flat.data_ptr() <= w.grad.data_ptr() < b.flat.data_ptr() + 4 * b.numel
               for b in red.buckets)          # still a view into a flat bucket
    torch.optim.AdamW(net.parameters(), lr=1e-3).step()


def _unused_worker(rank, world, port, tmp):
    """DDP(find_unused_parameters=True) semantics (/root/reference/main.py:156): a parameter NO rank used
    keeps .grad = None, so AdamW skips it (no weight decay, no state); a parameter only rank 0 used gets
    the average of its gradient and zero.  Compared with a single-process AdamW fed the averaged
    gradients (None where no rank produced one)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from datr_amd.dist import GradAllReducer, init_distributed
    from datr_amd.engine import _backward_and_step
    init_distributed(backend="gloo")
    torch.manual_seed(0)
    model, ref = Toy(), Toy()
    ref.load_state_dict(model.state_dict())
    red = GradAllReducer(model, bucket_mb=0.0005, first_bucket_mb=0.0002)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2, weight_decay=0.1)
    ropt = torch.optim.AdamW(ref.parameters(), lr=1e-2, weight_decay=0.1)
    for step in range(3):
        x = torch.randn(5, 8, generator=torch.Generator().manual_seed(100 * step + rank))
        sometimes = rank == 0 and step != 1                 # step 1: `sometimes` is unused on EVERY rank
        _backward_and_step(model, opt, model(x, sometimes), 0.1, None, False, red)
        flags = red.used_flags().tolist()
        names = [n for n, _ in model.named_parameters()]
        for n, u in zip(names, flags):
            expect = not n.startswith("unused") and (not n.startswith("sometimes") or step != 1)
            assert bool(u) == expect, (step, n, u)
        # single-process reference: average of the ranks' gradients, None where nobody had one
        ropt.zero_grad()
        ref(x, sometimes).backward()
        for q in ref.parameters():
            have = torch.tensor([0.0 if q.grad is None else 1.0])
            dist.all_reduce(have)
            local = torch.zeros_like(q) if q.grad is None else q.grad.clone()
            dist.all_reduce(local)
            q.grad = local / world if float(have) > 0 else None
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.1)
        ropt.step()
        for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            torch.testing.assert_close(p, q, rtol=1e-6, atol=1e-7, msg=f"step {step} {n}")
    assert model.unused.weight not in opt.state or not opt.state[model.unused.weight]      # never touched
    assert float(opt.state[model.sometimes.weight]["step"]) == 2                           # skipped once
    assert torch.equal(model.unused.weight, ref.unused.weight)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")


def test_globally_unused_parameters_keep_no_gradient_world2(tmp_path):
    port = _free_port()
    mp.spawn(_unused_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


# ---------------------------------------------------------------------------------------------
# ordered launching: a rank-dependent parameter that fills a bucket of its own
# ---------------------------------------------------------------------------------------------
class Staggered(nn.Module):
    """Three independent branches, each big enough to fill a bucket; rank 1 skips the middle
    one, so its buckets complete in a different order than rank 0's."""

    def __init__(self):
        super().__init__()
        self.a = nn.Linear(16, 16)
        self.b = nn.Linear(16, 16)
        self.c = nn.Linear(16, 16)

    def forward(self, x, use_b):
        out = self.c(x).sum() + self.a(x).sum()
        if use_b:
            out = out + self.b(x).sum()
        return out


def _ordered_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from datr_amd.dist import GradAllReducer, init_distributed
    init_distributed(backend="gloo")
    torch.manual_seed(rank)                     # different weights per rank: the constructor
    model = Staggered()                         # must broadcast rank 0's
    red = GradAllReducer(model, bucket_mb=272 * 4 / (1 << 20), first_bucket_mb=0.0)
    w = model.a.weight.detach().clone()
    dist.broadcast(w, 0)
    assert torch.equal(w, model.a.weight), "parameters were not broadcast from rank 0"
    assert len(red.buckets) == 3 and all(len(b.params) == 2 for b in red.buckets)
    ref = Staggered()
    ref.load_state_dict(model.state_dict())
    for step in range(4):
        x = torch.randn(5, 16, generator=torch.Generator().manual_seed(10 * step + rank))
        use_b = rank == 0 or step == 3          # rank 1 leaves `b` without gradient for 3 steps
        red.zero_grad()
        model(x, use_b).backward()
        red.finish()
        ref.zero_grad()
        ref(x, use_b).backward()
        for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            local = torch.zeros_like(q) if q.grad is None else q.grad.clone()
            dist.all_reduce(local)
            torch.testing.assert_close(p.grad, local / world, rtol=1e-6, atol=1e-7, msg=f"{step} {n}")
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")


def test_buckets_launch_in_fixed_order_when_ranks_use_different_parameters(tmp_path):
    """with arrival-order launching, rank 1 (no gradient for `b`) issues the
    collectives of `a` and `c` before `b`'s while rank 0 issues them as they arrive -- gloo
    aborts on the size mismatch, RCCL hangs.  Fixed-order launching cannot mis-order."""
    port = _free_port()
    mp.spawn(_ordered_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


# ---------------------------------------------------------------------------------------------
# the real detector through the epoch function, two ranks
# ---------------------------------------------------------------------------------------------
def _dino_worker(rank, world, port, tmp):
    import copy
    import sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(4)
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.join(here, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import synth
    from helpers import build_model
    from datr_amd import criterion as crit_mod, msda
    from datr_amd.config import get_param_dict
    from datr_amd.criterion import weighted_total
    from datr_amd.dist import init_distributed
    from datr_amd.engine import train_one_epoch
    from datr_amd.nested import nested_tensor_from_tensor_list
    from oracle import focal_oracle, msda_oracle as O
    # no GPU here: the two native ops are stood in by the oracle (test infrastructure)
    msda.ms_deform_attn_forward = lambda v, sh, lsi, loc, a, step: O.msda_forward(v, sh, lsi, loc, a)
    msda.ms_deform_attn_backward = lambda v, sh, lsi, loc, a, go, step: \
        list(O.msda_backward(v, sh, lsi, loc, a, go))
    crit_mod.focal_loss_sums = focal_oracle.focal_sums_torch
    init_distributed(backend="gloo")

    args, model, criterion, _ = build_model()
    if rank == 1:                                 # the reference seeds ranks differently
        with torch.no_grad():                     # (main.py:138) and relies on DDP's broadcast
            for p in model.parameters():
                p.add_(0.01 * torch.randn_like(p))
    optimizer = torch.optim.AdamW(get_param_dict(args, model), lr=args.lr,
                                  weight_decay=args.weight_decay)
    ref = None
    for step in range(2):
        # rank 0: 3 boxes; rank 1: an image pair WITHOUT any box (no de-noising queries, no
        # matched pairs: fewer gradients than rank 0 produces)
        imgs, targets = synth.synth_batch(seed=1 + 2 * step + rank, num_gt=3 if rank == 0 else 0)
        batch = (nested_tensor_from_tensor_list(imgs), tuple(targets), None, None)
        if ref is None:
            # engine creates the reducer on first use (-> broadcast); build the reference copy
            # from the post-broadcast weights by running the reducer's constructor first
            from datr_amd.dist import reducer_for
            red = reducer_for(model, args)
            assert red is not None and red.world == 2
            assert any(b.side for b in red.buckets), "D_img must sit in the side-stream bucket"
            w = model.transformer.level_embed.detach().clone()
            dist.broadcast(w, 0)
            assert torch.equal(w, model.transformer.level_embed)
            ref = copy.deepcopy(model)       # the reducer lives in a weak registry, not on the model
            from datr_amd.dist import attach_reducer
            attach_reducer(ref, False)
        else:
            ref.load_state_dict(model.state_dict())
            ref.global_proto, ref.Amount = proto_before
        proto_before = (model.global_proto.clone(), model.Amount.clone())
        ref.global_proto, ref.Amount = proto_before[0].clone(), proto_before[1].clone()
        nb_before = len(red.buckets)
        torch.manual_seed(100 + step)             # CDN noise
        stats = train_one_epoch(model, criterion, [batch], optimizer, torch.device("cpu"), 0,
                                max_norm=0, args=args)     # max_norm 0: .grad stays the reduced one
        assert stats["loss"] == stats["loss"]
        # single-process gradients of the same weights on this rank's batch, averaged by hand
        ref.train()
        torch.manual_seed(100 + step)
        criterion.prefetch_num_boxes(batch[1], torch.device("cpu"))
        out = ref(batch[0], list(batch[1]))
        loss = weighted_total(criterion(out, list(batch[1])), criterion.weight_dict)
        ref.zero_grad()
        loss.backward()
        worst = 0.0
        for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            if not p.requires_grad:
                continue
            local = torch.zeros_like(q) if q.grad is None else q.grad.clone()
            dist.all_reduce(local)
            local /= world
            assert p.grad is not None, n
            scale = float(local.abs().max()) + 1e-12
            err = float((p.grad - local).abs().max()) / scale
            worst = max(worst, err)
            assert err < 1e-4, (step, n, err)
        if step == 0:
            assert len(red.buckets) >= 3
            order_sizes = [b.numel for b in red.buckets]
            t = torch.tensor(order_sizes)
            dist.broadcast(t, 0)
            assert t.tolist() == order_sizes, "bucket layout differs between ranks"
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, f"ok{rank}"), "w").write(f"ok {worst:.2e} buckets {nb_before}")


def test_dino_epoch_function_world2_matches_single_process_gradients(tmp_path):
    """the real detector at 256x320 through `engine.train_one_epoch` on two
    gloo ranks with rank-different targets (rank 1 has NO boxes), the side-stream bucket, the
    num_boxes prefetch, the constructor broadcast and the bucket rebuild after step 1; the
    reduced gradients equal the average of the single-process gradients."""
    port = _free_port()
    mp.spawn(_dino_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


# ---------------------------------------------------------------------------------------------
# four ranks, the teacher-student epoch function, many buckets, rank-dependent pseudo labels
# ---------------------------------------------------------------------------------------------
def _selftrain_worker(rank, world, port, tmp):
    import copy
    import sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.join(here, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import synth
    from helpers import build_model
    from datr_amd import criterion as crit_mod, msda
    from datr_amd.config import get_param_dict
    from datr_amd.dist import GradAllReducer, attach_reducer, init_distributed, reducer_for
    from datr_amd.ema import ModelEMA
    from datr_amd.engine import train_one_epoch_with_self_training
    from datr_amd.nested import nested_tensor_from_tensor_list
    from oracle import focal_oracle, msda_oracle as O
    msda.ms_deform_attn_forward = lambda v, sh, lsi, loc, a, step: O.msda_forward(v, sh, lsi, loc, a)
    msda.ms_deform_attn_backward = lambda v, sh, lsi, loc, a, go, step: \
        list(O.msda_backward(v, sh, lsi, loc, a, go))
    crit_mod.focal_loss_sums = focal_oracle.focal_sums_torch
    init_distributed(backend="gloo")

    args, model, criterion, _ = build_model()
    # ranks 0 and 2 find pseudo labels, ranks 1 and 3 find none (threshold above every score):
    # the target-domain criterion then returns {} there and fewer gradients are produced
    args.pseudo_label_threshold = 0.02 if rank % 2 == 0 else 0.999
    teacher = ModelEMA(model, decay=args.ema_decay_teacher)
    red = GradAllReducer(model, bucket_mb=24.0, first_bucket_mb=2.0)
    attach_reducer(model, red)
    assert len(red.buckets) >= 6, len(red.buckets)
    ref = copy.deepcopy(model)
    attach_reducer(ref, False)
    assert reducer_for(ref, args) is None and reducer_for(model, args) is red

    imgs, targets = synth.synth_batch(seed=3 + rank, sizes=((192, 240), (184, 232)), num_gt=2 + rank % 2)
    g = torch.Generator().manual_seed(50 + rank)
    strong = [imgs[0], imgs[1] + 0.3 * torch.randn(imgs[1].shape, generator=g)]
    meta = [{"image_id": torch.tensor([rank]), "orig_size": torch.tensor([368, 464]),
             "size": torch.tensor([184, 232]), "boxes": torch.zeros(0, 4),
             "labels": torch.zeros(0, dtype=torch.long)}]
    loader = [(nested_tensor_from_tensor_list(imgs), tuple(targets), tuple(meta),
               nested_tensor_from_tensor_list(strong))]

    def run(m):
        opt = torch.optim.SGD(get_param_dict(args, m), lr=0.0)        # gradients only
        torch.manual_seed(200 + rank)                                  # CDN noise
        return train_one_epoch_with_self_training(m, teacher, criterion, loader, loader, opt,
                                                  torch.device("cpu"), 0, max_norm=0, args=args)
    s_ref = run(ref)                     # single-process gradients of this rank's batch
    s_red = run(model)                   # the same step with the reducer
    assert s_ref["_last"]["num_pseudo_images"] == s_red["_last"]["num_pseudo_images"] == (1 if rank % 2 == 0 else 0)
    n_pseudo = torch.tensor([s_red["_last"]["num_pseudo_images"]])
    dist.all_reduce(n_pseudo)
    assert int(n_pseudo) == 2, "two of the four ranks must have produced pseudo labels"
    worst, missing = 0.0, 0
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        if not p.requires_grad:
            continue
        missing += q.grad is None
        local = torch.zeros_like(q) if q.grad is None else q.grad.clone()
        dist.all_reduce(local)
        local /= world
        assert p.grad is not None, n
        scale = float(local.abs().max()) + 1e-12
        err = float((p.grad - local).abs().max()) / scale
        worst = max(worst, err)
        assert err < 1e-4, (n, err)
    sizes = torch.tensor([b.numel for b in red.buckets])
    dist.broadcast(sizes, 0)
    assert sizes.tolist() == [b.numel for b in red.buckets], "bucket layout differs between ranks"
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, f"ok{rank}"), "w").write(f"ok {worst:.2e} buckets {len(red.buckets)} missing {missing}")


def test_selftraining_epoch_function_world4_many_buckets_rank_dependent_pseudo_labels(tmp_path):
    """four gloo ranks through `engine.train_one_epoch_with_self_training` with the
    real detector (tiny images), >= 6 gradient buckets, and pseudo labels on ranks 0 / 2 only -- the
    ranks without them skip the target-domain criterion and produce fewer gradients, yet every rank
    issues the same collectives in the same order and ends with the average of the four
    single-process gradients."""
    port = _free_port()
    mp.spawn(_selftrain_worker, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(4))


# ---------------------------------------------------------------------------------------------
# eight ranks (the size of the driver's SCALE run), the real detector, two ranks without boxes
# ---------------------------------------------------------------------------------------------
def _world8_worker(rank, world, port, tmp):
    import copy
    import sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.join(here, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import synth
    from helpers import build_model
    from datr_amd import criterion as crit_mod, msda
    from datr_amd.config import get_param_dict
    from datr_amd.dist import GradAllReducer, attach_reducer, init_distributed
    from datr_amd.engine import train_one_epoch
    from datr_amd.nested import nested_tensor_from_tensor_list
    from oracle import focal_oracle, msda_oracle as O
    msda.ms_deform_attn_forward = lambda v, sh, lsi, loc, a, step: O.msda_forward(v, sh, lsi, loc, a)
    msda.ms_deform_attn_backward = lambda v, sh, lsi, loc, a, go, step: \
        list(O.msda_backward(v, sh, lsi, loc, a, go))
    crit_mod.focal_loss_sums = focal_oracle.focal_sums_torch
    init_distributed(backend="gloo")

    args, model, criterion, _ = build_model()
    with torch.no_grad():                         # ranks start from different weights (main.py:138)
        g = torch.Generator().manual_seed(1000 + rank)
        for p in model.parameters():
            p.add_(0.01 * rank * torch.randn(p.shape, generator=g))
    red = GradAllReducer(model, bucket_mb=24.0, first_bucket_mb=2.0)     # constructor broadcast
    attach_reducer(model, red)
    assert red.world == 8 and len(red.buckets) >= 8, len(red.buckets)
    w = model.transformer.level_embed.detach().clone()
    dist.broadcast(w, 0)
    assert torch.equal(w, model.transformer.level_embed), "rank 0's weights on every rank"
    ref = copy.deepcopy(model)
    attach_reducer(ref, False)

    boxless = rank in (2, 5)
    imgs, targets = synth.synth_batch(seed=11 + rank, sizes=((192, 240), (184, 232)),
                                      num_gt=0 if boxless else 1 + rank % 3)
    batch = (nested_tensor_from_tensor_list(imgs), tuple(targets), None, None)
    opt = torch.optim.SGD(get_param_dict(args, model), lr=0.0)           # gradients only
    torch.manual_seed(300 + rank)                                         # CDN noise
    stats = train_one_epoch(model, criterion, [batch], opt, torch.device("cpu"), 0, max_norm=0, args=args)
    assert stats["loss"] == stats["loss"]
    opt_ref = torch.optim.SGD(get_param_dict(args, ref), lr=0.0)
    torch.manual_seed(300 + rank)
    train_one_epoch(ref, criterion, [batch], opt_ref, torch.device("cpu"), 0, max_norm=0, args=args)

    worst, missing = 0.0, 0
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        if not p.requires_grad:
            continue
        missing += q.grad is None
        local = torch.zeros_like(q) if q.grad is None else q.grad.clone()
        dist.all_reduce(local)
        local /= world
        assert p.grad is not None, n
        scale = float(local.abs().max()) + 1e-12
        err = float((p.grad - local).abs().max()) / scale
        worst = max(worst, err)
        assert err < 1e-4, (n, err)
    sizes = torch.tensor([b.numel for b in red.buckets])
    dist.broadcast(sizes, 0)
    assert sizes.tolist() == [b.numel for b in red.buckets], "bucket layout differs between ranks"
    flags = red.used_flags().clone()
    f0 = flags.clone()
    dist.broadcast(f0, 0)
    assert torch.equal(f0, flags), "MAX-reduced used flags are the same on every rank"
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, f"ok{rank}"), "w").write(f"ok {worst:.2e} buckets {len(red.buckets)} missing {missing}")


def test_epoch_function_world8_two_boxless_ranks(tmp_path):
    """the reducer at the world size of the driver's SCALE run -- eight gloo ranks with
    the real detector (tiny images) through `engine.train_one_epoch`, >= 8 gradient buckets, ranks 2 and 5
    WITHOUT any box (no de-noising queries, no matched pairs: fewer gradients); every rank issues the same
    collectives in the same order and ends with the average of the eight single-process gradients and the
    same used-parameter flags."""
    port = _free_port()
    mp.spawn(_world8_worker, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(8))
