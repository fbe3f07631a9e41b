"""Where the step's SMALL kernels (< 20 us) come from: GPU time and launch count per forward source
line (datr_amd / bench frames) and per autograd backward node (torch.profiler with stacks)."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datr_amd import training as bench  # noqa: E402


class A:
    flat_grads = False
    tuned_gemm = True
    channels_last = True


LIMIT_US = 20.0
dev = torch.device("cuda:0")
tr = bench.Stepper(dev)
samples, targets = bench.synthetic_batch(2, 800, 1333, 10, dev, seed=1)
samples.tensors = samples.tensors.contiguous(memory_format=torch.channels_last)
# no python stacks in this build's profiler: label regions by wrapping the callables
import functools
from torch.profiler import record_function
import datr_amd.denoising, datr_amd.detector, datr_amd.transformer as T, datr_amd.criterion as Cr
import datr_amd.matcher as Mt


def label(obj, name, tag=None):
    fn = getattr(obj, name)

    @functools.wraps(fn)
    def wrapped(*a, **k):
        with record_function("REGION:" + (tag or name)):
            return fn(*a, **k)
    setattr(obj, name, wrapped)


model = tr.model.module if hasattr(tr.model, "module") else tr.model
label(datr_amd.detector, "prepare_for_cdn")
label(datr_amd.detector, "dn_post_process")
label(T, "gen_sineembed_for_position")
label(T, "gen_encoder_output_proposals")
label(T.TransformerEncoder, "forward", "encoder")
label(T.DeformableTransformerDecoderLayer, "forward_sa", "dec.self_attn")
label(T.DeformableTransformerDecoderLayer, "forward_ca", "dec.cross_attn")
label(T.DeformableTransformerDecoderLayer, "forward_ffn", "dec.ffn")
label(T.TransformerDecoder, "forward", "decoder(other)")
label(T.DeformableTransformer, "encode", "transformer.encode(other)")
label(T.DeformableTransformer, "decode", "transformer.decode(other)")
label(Mt.HungarianMatcher, "cost_matrix", "matcher.cost")
label(Mt.HungarianMatcher, "forward_many", "matcher(other)")
label(Cr.SetCriterion, "_family_losses", "criterion.family_losses")
label(Cr.SetCriterion, "loss_da")
label(Cr.SetCriterion, "loss_proto_da")
label(Cr.SetCriterion, "loss_contrast_da")
label(Cr.SetCriterion, "forward", "criterion(other)")
label(type(model), "_heads", "heads")
label(type(model.backbone), "forward", "backbone+posembed")
label(type(model), "forward", "detector(other)")
for _ in range(3):
    tr.step(samples, targets)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(samples, targets)
    torch.cuda.synchronize()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fwd, bwd = collections.defaultdict(lambda: [0.0, 0]), collections.defaultdict(lambda: [0.0, 0])
total = [0.0, 0]


def region_of(ev):
    p = ev.cpu_parent
    while p is not None:
        if p.name.startswith("REGION:"):
            return p.name[7:]
        p = p.cpu_parent
    return None


# autograd sequence numbers tie a backward node to the forward op that created it
seq2region = {}
for ev in prof.events():
    if ev.device_type.name == "CPU" and getattr(ev, "sequence_nr", -1) is not None \
            and getattr(ev, "sequence_nr", -1) >= 0 and "evaluate_function" not in ev.name \
            and "Backward" not in ev.name:
        r = region_of(ev)
        if r is not None:
            seq2region.setdefault(ev.sequence_nr, r)
bwd_region = collections.defaultdict(lambda: [0.0, 0])
for ev in prof.events():
    if ev.device_type.name != "CPU" or not getattr(ev, "kernels", None):
        continue
    if ev.name.startswith("hip") or any(getattr(c, "kernels", None) and not c.name.startswith("hip")
                                        for c in ev.cpu_children):
        continue                             # count kernels at the innermost op only
    small = [k for k in ev.kernels if k.duration < LIMIT_US]
    if not small:
        continue
    t, n = sum(k.duration for k in small), len(small)
    total[0] += t
    total[1] += n
    region, p = None, ev.cpu_parent
    while p is not None and region is None:
        if p.name.startswith("REGION:"):
            region = p.name[7:]
        p = p.cpu_parent
    if region is not None:
        fwd[region + " :: " + ev.name][0] += t
        fwd[region + " :: " + ev.name][1] += n
        fwd["== " + region][0] += t
        fwd["== " + region][1] += n
    else:
        p, name, seq = ev.cpu_parent, ev.name, None
        while p is not None:
            if "evaluate_function" in p.name or "Backward" in p.name or "Optimizer" in p.name:
                name = p.name
                if getattr(p, "sequence_nr", -1) is not None and getattr(p, "sequence_nr", -1) >= 0:
                    seq = p.sequence_nr
            p = p.cpu_parent
        bwd[name][0] += t
        bwd[name][1] += n
        if seq is not None:
            r = seq2region.get(seq, "(unattributed)")
            bwd_region[r][0] += t
            bwd_region[r][1] += n
            bwd_region[r + " :: " + name.replace("autograd::engine::evaluate_function: ", "")][0] += t
            bwd_region[r + " :: " + name.replace("autograd::engine::evaluate_function: ", "")][1] += n
import itertools
for ev in itertools.islice((e for e in prof.events() if e.stack), 3):
    print("sample stack:", ev.name, ev.stack[:6])
print(f"small kernels (< {LIMIT_US} us): {total[0] / 1e3:.2f} ms in {total[1]} launches")
print("forward, by source line:")
for k, (t, n) in sorted(fwd.items(), key=lambda x: -x[1][0])[:70]:
    print(f"  {t / 1e3:6.3f} ms {n:4d}  {k[:110]}")
print("backward / other, by autograd node:")
for k, (t, n) in sorted(bwd.items(), key=lambda x: -x[1][0])[:40]:
    print(f"  {t / 1e3:6.3f} ms {n:4d}  {k[:110]}")
print("backward, by the forward region that created the node:")
for k, (t, n) in sorted(bwd_region.items(), key=lambda x: -x[1][0])[:60]:
    print(f"  {t / 1e3:6.3f} ms {n:4d}  {k[:110]}")
