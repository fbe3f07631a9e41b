"""Compare two directories of golden fixtures array by array (used when a generator or the oracle changes:
`python tools/diff_golden.py /tmp/golden_old tests/golden`).  Prints, per file, the arrays that differ and by
how much (max abs / max relative-to-max difference); integer and string arrays must be equal."""
import sys
import os
import numpy as np


def main(a, b):
    worst = 0.0
    for f in sorted(os.listdir(a)):
        if not f.endswith(".npz") or not os.path.exists(os.path.join(b, f)):
            continue
        x, y = np.load(os.path.join(a, f), allow_pickle=False), np.load(os.path.join(b, f), allow_pickle=False)
        keys = sorted(set(x.files) | set(y.files))
        ndiff = 0
        for k in keys:
            if k not in x.files or k not in y.files:
                print(f"{f}:{k}: only in {'old' if k in x.files else 'new'}")
                ndiff += 1
                continue
            u, v = x[k], y[k]
            if u.shape != v.shape or u.dtype != v.dtype:
                print(f"{f}:{k}: shape/dtype {u.shape}{u.dtype} -> {v.shape}{v.dtype}")
                ndiff += 1
                continue
            if u.dtype.kind in "fc":
                if not np.array_equal(u, v, equal_nan=True):
                    d = np.abs(u.astype(np.float64) - v.astype(np.float64))
                    fin = np.isfinite(d)
                    mx = float(d[fin].max()) if fin.any() else 0.0
                    scale = float(np.abs(u[np.isfinite(u)]).max()) if np.isfinite(u).any() else 1.0
                    rel = mx / max(scale, 1e-30)
                    worst = max(worst, rel)
                    print(f"{f}:{k}: max|d| {mx:.3e}  (max|x| {scale:.3e}, rel {rel:.2e}), {int((d > 0).sum())}/{d.size} differ")
                    ndiff += 1
            elif not np.array_equal(u, v):
                print(f"{f}:{k}: INTEGER/STRING ARRAY DIFFERS ({int((u != v).sum())} entries)")
                ndiff += 1
        print(f"{f}: {len(keys)} arrays, {ndiff} differ")
    print(f"worst relative difference {worst:.2e}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
