"""Re-create gemm_mi355x.csv on a MI355X:  python -m datr_amd.tuning.retune
Runs two steps of the training step bench.py times (datr_amd.training.Stepper) with TunableOp tuning enabled and writes
the selections next to this file."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from datr_amd import tuning
    from datr_amd.training import Stepper, synthetic_batch
    t = torch.cuda.tunable
    t.enable(True)
    t.tuning_enable(True)
    t.set_max_tuning_duration(int(os.environ.get("DATR_TUNE_MS", "60")))
    if os.environ.get("DATR_TUNE_ITERS"):
        t.set_max_tuning_iterations(int(os.environ["DATR_TUNE_ITERS"]))
    t.set_filename("/tmp/datr_tunableop_scratch.csv", insert_device_ordinal=False)
    dev = torch.device("cuda:0")
    tr = Stepper(dev, tuned_gemm=False, channels_last=True)
    tuning.enable_miopen_db()
    for size, b, gt in (((800, 1333), 2, 10), ((640, 640), 1, 5)):
        samples, targets = synthetic_batch(b, size[0], size[1], gt, dev, seed=1)
        for _ in range(2):
            tr.step(samples, targets)
        # the eval-mode forward of the teacher / the evaluation loop (target half only: other M)
        tr.model.eval()
        with torch.no_grad():
            half = samples.tensors.shape[0] // 2
            tr.model(samples.tensors[half:].contiguous(memory_format=torch.channels_last))
        tr.model.train()
    # BASELINE config 2 read literally (bench.py --stage source-only): B source images, DA branch off -- the encoder's
    # GEMMs then have half the rows of the DA step's, in both directions
    tr.model.domain_adaptation = False
    samples, targets = synthetic_batch(2, 800, 1333, 10, dev, seed=1, source_only=True)
    for _ in range(2):
        tr.step(samples, targets)
    tr.model.domain_adaptation = True
    torch.cuda.synchronize()
    results = {(op, params): (solution, ms) for op, params, solution, ms in t.get_results()}
    kept = 0
    # shapes not seen in this run keep their selection; DATR_TUNE_KEEP=1: so do the shapes that already have one
    # (adding the shapes of a new workload without re-rolling the selections the committed measurements were made with)
    keep_all = os.environ.get("DATR_TUNE_KEEP", "0") != "0"
    if os.path.exists(tuning.RESULTS):
        for line in open(tuning.RESULTS):
            p = line.rstrip("\n").split(",")
            if len(p) >= 4 and p[0] != "Validator" and (keep_all or (p[0], p[1]) not in results):
                results[(p[0], p[1])] = (p[2], p[3])
                kept += 1
    with open(tuning.RESULTS, "w") as f:        # same layout TunableOp itself writes at exit
        for key, val in t.get_validators():
            f.write(f"Validator,{key},{val}\n")
        for (op, params), (solution, ms) in results.items():
            f.write(f"{op},{params},{solution},{ms}\n")
    print("wrote", tuning.RESULTS, "with", len(results), "entries,", kept, "kept from the previous file")


if __name__ == "__main__":
    main()
