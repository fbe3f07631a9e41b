"""Which library convolutions still run in the teacher-student stage (bench.py --stage teacher): every
aten::convolution / miopen call of one step with its input shapes and the python frame that issued it.
Usage: python tools/probes/teacher_conv_calls.py"""
import argparse
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
args = argparse.Namespace(batch=2, height=800, width=1333, num_gt=10, warmup=2, steps=1, channels_last=True)
orig_sync = torch.cuda.synchronize
state = {"n": 0}


def sync_and_profile(*a, **k):
    orig_sync(*a, **k)


# run the stage once for warm-up, then once under the profiler
bench.teacher_student_stage(args, dev)
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    bench.teacher_student_stage(argparse.Namespace(**{**vars(args), "warmup": 1, "steps": 1}), dev)
seen = {}
for ev in prof.events():
    if ev.name in ("aten::convolution", "aten::miopen_convolution", "aten::_convolution", "aten::convolution_backward",
                   "aten::miopen_depthwise_convolution", "aten::conv2d"):
        stack = [s for s in (ev.stack or []) if "datr_amd" in s or "bench.py" in s]
        if ev.name == "aten::conv2d" and not state["n"]:
            state["n"] = 1
            print("full stack of the first aten::conv2d:", *(ev.stack or [])[:25], sep="\n    ")
        key = (ev.name, str(ev.input_shapes)[:120], stack[0][:110] if stack else "?")
        seen[key] = seen.get(key, 0) + 1
for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(v, k)
