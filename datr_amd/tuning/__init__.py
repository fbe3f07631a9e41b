"""Per-shape GEMM kernel selection for the dense layers of the hot path.

The encoder / decoder linears run through hipBLASLt (torch.addmm / mm).  hipBLASLt's default
heuristic picks kernels that reach ~83 TF/s on the FFN shapes of the 1333x800 step
([88892,256]x[256,2048] and friends); PyTorch's TunableOp picks, per shape, the fastest of the
library's kernels (127-146 TF/s on the same shapes = 81-93 % of the 157 TF/s fp32-MFMA peak).
`gemm_mi355x.csv` is the result of one tuning run of bench.py on a MI355X
(`python -m datr_amd.tuning.retune`, a couple of minutes); `enable()` loads it with tuning
switched OFF, so a training run only looks selections up -- shapes that are not in the file
keep the library default.  Numerics are unchanged: every candidate is a plain fp32 GEMM.
"""
from __future__ import annotations

import os
import warnings

import torch

RESULTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_mi355x.csv")


def enable(path: str = RESULTS, tune: bool = False) -> bool:
    """Turn TunableOp on with the shipped selections.  Returns False (and leaves everything at
    the library defaults) when the file is missing or this build has no TunableOp."""
    if not torch.cuda.is_available() or not hasattr(torch.cuda, "tunable"):
        return False
    t = torch.cuda.tunable
    if not os.path.exists(path) and not tune:
        return False
    t.enable(True)
    t.tuning_enable(bool(tune))
    if os.path.exists(path):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ok = t.read_file(path)
        if not ok and not tune:
            t.enable(False)         # validators (torch / hipBLASLt versions) do not match
            return False
    return True
