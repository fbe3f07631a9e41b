"""EXPERIMENT: the training step with the GEMM family's split-bf16 inner product (DATR_GEMM_SPLIT_BF16=1) against
the default exact-fp32 MFMA one -- same seed, same batches: per-step total loss of both runs, and how far the
parameters have drifted apart after the last step, beside the same comparison between two DEFAULT runs whose only
difference is the GEMM tile plan (another fp32 summation order): the yardstick for "differs by rounding".
    python tools/probes/split_bf16/step_parity.py [--steps 8]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
from datr_amd import engine, training  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda:0")
batches = [training.synthetic_batch(2, 800, 1333, 10, dev, seed=s) for s in range(4)]


def run(env):
    for k in ("DATR_GEMM_SPLIT_BF16", "DATR_GEMM_PLAN"):
        os.environ.pop(k, None)
    os.environ.update(env)
    torch.manual_seed(0)
    st = training.Stepper(dev, seed=0)
    losses = []
    for i in range(a.steps):
        torch.manual_seed(100 + i)                       # the de-noising noise of the step
        stats = st.step(*batches[i % len(batches)])
        losses.append(float(stats["loss"]) if isinstance(stats, dict) and "loss" in stats else float("nan"))
    torch.cuda.synchronize()
    params = {n: p.detach().clone() for n, p in st.model.named_parameters()}
    for k in env:
        os.environ.pop(k, None)
    return losses, params


def drift(p, q):
    num = sum(float((p[n].double() - q[n].double()).pow(2).sum()) for n in p)
    den = sum(float(p[n].double().pow(2).sum()) for n in p)
    worst = max(float((p[n] - q[n]).abs().max()) for n in p)
    return (num / den) ** 0.5, worst


base_l, base_p = run({})
plan_l, plan_p = run({"DATR_GEMM_PLAN": "2,2,32,0"})
split_l, split_p = run({"DATR_GEMM_SPLIT_BF16": "1"})
print("step   default        default, other GEMM tile plan   split-bf16 x6")
for i, (x, y, z) in enumerate(zip(base_l, plan_l, split_l)):
    print(f"{i:3d}   {x:12.6f}   {y:12.6f} ({abs(y - x) / abs(x):.1e})      {z:12.6f} ({abs(z - x) / abs(x):.1e})")
d1, d2 = drift(base_p, plan_p), drift(base_p, split_p)
print(f"parameters after {a.steps} steps, relative L2 distance / largest element difference from the default run:")
print(f"   other tile plan (fp32 MFMA, another summation order): {d1[0]:.3e} / {d1[1]:.3e}")
print(f"   split-bf16 x6:                                        {d2[0]:.3e} / {d2[1]:.3e}")
