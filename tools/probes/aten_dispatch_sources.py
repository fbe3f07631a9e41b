"""Which source lines of datr_amd/ issue the ATen element-wise / copy / reduction ops of one training step:
a TorchDispatchMode records every aten op with the innermost datr_amd/ frame of the Python stack (autograd
runs single-threaded so that the backward's ops are seen too; ops of native autograd nodes have no such frame
and are listed under their op name only).  GEMM / convolution ops are skipped.  Counts, not times: join with
tools/probes/aten_sources.py (device time per op and shape).
    python tools/probes/aten_dispatch_sources.py [--rows 80]"""
import argparse
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from datr_amd import training as bench  # noqa: E402

SKIP = ("mm", "addmm", "bmm", "_addmm_activation", "convolution", "view", "_unsafe_view", "reshape", "t", "transpose",
        "permute", "expand", "as_strided", "detach", "alias", "slice", "select", "unbind", "split", "split_with_sizes",
        "squeeze", "unsqueeze", "empty", "empty_like", "empty_strided", "unflatten", "flatten", "_reshape_alias",
        "contiguous", "is_same_size", "size", "stride", "record_stream", "lift_fresh", "new_empty", "chunk", "unfold",
        "view_as", "narrow", "movedim", "result_type", "sym_size", "sym_stride", "sym_numel", "is_pinned", "diagonal")


class Rec(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.Counter()
        self.big = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__ if hasattr(func, "overloadpacket") else str(func)
        out = func(*args, **(kwargs or {}))
        if name in SKIP:
            return out
        dev = [a for a in args if torch.is_tensor(a) and a.is_cuda]
        outs = out if isinstance(out, (tuple, list)) else (out,)
        if not dev and not any(torch.is_tensor(o) and o.is_cuda for o in outs):
            return out
        where = "?"
        for fr in reversed(traceback.extract_stack(limit=40)):
            if "datr_amd/" in fr.filename:
                where = f"{fr.filename.split('datr_amd/')[-1]}:{fr.lineno} {fr.name}"
                break
        node = torch._C._current_autograd_node()
        if node is not None and "_backward_and_step" in where:
            # a native autograd node: anomaly mode kept the forward stack that created it
            where = "bwd " + node.name()
            tb = node.metadata.get("traceback_", None)
            if tb:
                for line in reversed(tb):
                    if "datr_amd/" in line and "File" in line:
                        f_ = line.split("datr_amd/")[-1].split('"')[0]
                        ln = line.split("line ")[1].split(",")[0]
                        fn = line.split(" in ")[1].split("\n")[0] if " in " in line else ""
                        where = f"bwd {node.name()} <- {f_}:{ln} {fn}"
                        break
        big = max([a.numel() for a in args if torch.is_tensor(a)] + [o.numel() for o in outs if torch.is_tensor(o)] + [0])
        self.rows[(where, name, "big" if big >= 1 << 20 else "small")] += 1
        if big >= 1 << 20:
            shp = next((tuple(o.shape) for o in outs if torch.is_tensor(o) and o.numel() == big), None) or \
                next(tuple(a.shape) for a in args if torch.is_tensor(a) and a.numel() == big)
            self.big[(where, name, shp)] += 1
        return out


ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=80)
a = ap.parse_args()
dev = torch.device("cuda:0")
tr = bench.Stepper(dev)
samples, targets = bench.synthetic_batch(2, 800, 1333, 10, dev, seed=1)
for _ in range(3):
    tr.step(samples, targets)
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)
rec = Rec()
with torch.autograd.detect_anomaly(check_nan=False), rec:
    tr.step(samples, targets)
torch.cuda.synchronize()
total = sum(rec.rows.values())
print(f"{total} device ATen ops (views / GEMMs / convolutions excluded) in one step")
by_where = collections.Counter()
for (w, n, s), c in rec.rows.items():
    by_where[w] += c
print("---- by source line")
for w, c in by_where.most_common(a.rows):
    ops = collections.Counter({n + ("*" if s == "big" else ""): c2 for (w2, n, s), c2 in rec.rows.items() if w2 == w})
    print(f"{c:5d}  {w:84s} " + " ".join(f"{n}x{k}" for n, k in ops.most_common(6)))
print("---- ops on >= 1M elements: source, op, shape")
for (w, n, shp), c in sorted(rec.big.items(), key=lambda kv: -kv[1] * __import__("math").prod(kv[0][2])):
    print(f"{c:4d}  {n:24s} {str(shp):28s} {w}")
