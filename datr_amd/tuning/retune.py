"""Re-create gemm_mi355x.csv on a MI355X:  python -m datr_amd.tuning.retune
Runs two warm-up steps of bench.py's training step with TunableOp tuning enabled and writes
the selections next to this file."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import bench
    from datr_amd import tuning
    t = torch.cuda.tunable
    t.enable(True)
    t.tuning_enable(True)
    t.set_max_tuning_duration(60)
    t.set_filename("/tmp/datr_tunableop_scratch.csv", insert_device_ordinal=False)

    class A:
        flat_grads = False
        tuned_gemm = False
        channels_last = True
    dev = torch.device("cuda:0")
    tr = bench.Trainer(A, dev, distributed=False)
    for size, b, gt in (((800, 1333), 2, 10), ((640, 640), 1, 5)):
        samples, targets = bench.synthetic_batch(b, size[0], size[1], gt, dev, seed=1)
        for _ in range(2):
            tr.step(samples, targets)
    torch.cuda.synchronize()
    with open(tuning.RESULTS, "w") as f:        # same layout TunableOp itself writes at exit
        for key, val in t.get_validators():
            f.write(f"Validator,{key},{val}\n")
        for op, params, solution, ms in t.get_results():
            f.write(f"{op},{params},{solution},{ms}\n")
    print("wrote", tuning.RESULTS, "with", len(t.get_results()), "entries")


if __name__ == "__main__":
    main()
