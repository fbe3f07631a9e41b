"""List every host<->device synchronisation inside one training step (torch's sync debug mode).

    python tools/sync_audit.py            # on a GPU box
"""
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datr_amd import training as bench  # noqa: E402


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--dist", action="store_true",
                    help="one-rank RCCL process group + flat-bucket reducer with every collective issued "
                         "(DATR_DIST_FORCE_COLLECTIVES=1 must be set in the environment)")
    ap.add_argument("--small", action="store_true", help="256x320 images instead of 800x1333")
    cli = ap.parse_args()
    dev = torch.device("cuda:0")
    if cli.dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=0, world_size=1)
    args = argparse.Namespace(tuned_gemm=True, channels_last=True, flat_grads=False)
    tr = bench.Stepper(dev, reducer=bool(cli.dist))
    hh, ww = (256, 320) if cli.small else (800, 1333)
    samples, targets = bench.synthetic_batch(2, hh, ww, 10, dev, seed=1)
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
    if cli.dist:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
