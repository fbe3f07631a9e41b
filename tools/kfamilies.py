"""Kernel time by family from a tools/kstats.py step table.
    python tools/kfamilies.py profiles/r04_step_kernels.csv"""
import csv
import sys
from collections import defaultdict

FAMILIES = [
    ("hipBLASLt/rocBLAS GEMM", ("Cijk_",)),
    ("own GEMM family", ("gemm_f32_kernel", "gemm_splitk_fold", "gemm_colsum_finish")),
    ("MIOpen", ("igemm", "SubTensorOp", "miopen", "Im2")),
    ("own Winograd", ("wino_",)),
    ("own stride-2 / stem", ("tap_", "stem_conv", "even_pixels")),
    ("own MSDA", ("msda_", "prologue_", "owner_")),
    ("own attention", ("mha_",)),
    ("own LN / GN / affine / ffn / addn / colsum", ("add_ln", "gn_", "affine_", "relu_bwd_bias", "add_n_kernel", "colsum")),
    ("own criterion / matcher / misc", ("focal", "box_loss", "match_cost", "lsap", "topk", "sine_embed", "nms", "ema_", "conv_c1", "c1_", "cout1", "zero_rows")),
    ("ATen / rocclr", ("at::", "__amd_rocclr", "rocclr")),
]


def main():
    rows = [r for r in csv.reader(l for l in open(sys.argv[1]) if not l.startswith("#"))][1:]
    fam = defaultdict(lambda: [0.0, 0.0])
    for name, calls, ms, *_ in rows:
        for f, keys in FAMILIES:
            if any(k in name for k in keys):
                break
        else:
            f = "other: " + name[:40]
        fam[f][0] += float(ms)
        fam[f][1] += float(calls)
    total = sum(v[0] for v in fam.values())
    for f, (ms, calls) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        print(f"{ms:8.3f} ms {calls:7.0f} launches  {f}")
    print(f"{total:8.3f} ms total")


if __name__ == "__main__":
    main()
