"""List every host<->device synchronisation inside one training step (torch's sync debug mode).

    python tools/sync_audit.py            # on a GPU box
"""
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import argparse
    dev = torch.device("cuda:0")
    args = argparse.Namespace(tuned_gemm=True, channels_last=True, flat_grads=False)
    tr = bench.Trainer(args, dev, False)
    samples, targets = bench.synthetic_batch(2, 800, 1333, 10, dev, seed=1)
    for _ in range(3):
        tr.step(samples, targets)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("warn")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        tr.step(samples, targets)
    torch.cuda.set_sync_debug_mode("default")
    import traceback
    seen = {}
    for x in w:
        key = f"{x.filename}:{x.lineno}  {str(x.message)[:80]}"
        seen[key] = seen.get(key, 0) + 1
    for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
        print(v, k)
    print("total synchronising calls in one step:", len(w))


if __name__ == "__main__":
    main()
