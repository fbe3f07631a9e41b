"""Per-shape library-kernel selection for the dense layers and convolutions of the hot path.

GEMMs
-----
The encoder / decoder linears run through hipBLASLt (torch.addmm / mm).  hipBLASLt's default
heuristic picks kernels that reach ~83 TF/s on the FFN shapes of the 1333x800 step
([88892,256]x[256,2048] and friends); PyTorch's TunableOp picks, per shape, the fastest of the
library's kernels (127-146 TF/s on the same shapes = 81-93 % of the 157 TF/s fp32-MFMA peak).
`gemm_mi355x.csv` is the result of one tuning run of bench.py on a MI355X
(`python -m datr_amd.tuning.retune`, a couple of minutes); `enable()` loads it with tuning
switched OFF, so a training run only looks selections up -- shapes that are not in the file
keep the library default.  Numerics are unchanged: every candidate is a plain fp32 GEMM.

Convolutions
------------
(Since round 4 no MIOpen kernel is left in the training step -- every convolution runs on own kernels -- so
the find-db below is only loaded on request, DATR_MIOPEN_DB=1: it matters for the A/B switches that send
convolutions back to the library, DATR_OWN_CONV3X3=0 / DATR_OWN_CONV_S2=0 / --no-channels-last.)
PyTorch calls MIOpen in "immediate" mode (`torch.backends.cudnn.benchmark = False`): for a
problem MIOpen has never measured it falls back to a heuristic solver choice.  `miopen/` holds
MIOpen's own user find-db / perf-db text files after exhaustive Find runs of bench.py's
1333x800 step on a MI355X, once with the backbone in NCHW and once in NHWC
(`tools/probes/bench_cudnn_benchmark.py`, ~8 + 2 min); `enable()` copies them to a writable
directory and points `MIOPEN_USER_DB_PATH` at it before the first convolution, so immediate
mode picks the measured-fastest solver for those shapes (117.7 -> 113.0 ms per step in NCHW;
110.3 ms with the backbone in torch.channels_last, where the NHWC implicit-GEMM solvers need no
layout transposes).  Other shapes keep MIOpen's defaults; every solver is an fp32 convolution.
"""
from __future__ import annotations

import os
import warnings

import torch

RESULTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_mi355x.csv")
MIOPEN_DB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen")


def enable_miopen_db(src: str = MIOPEN_DB) -> bool:
    """Point MIOpen's user find-db at a writable copy of the shipped one.  Must run before the
    process' first convolution (MIOpen reads the variable when it creates its handle); a
    MIOPEN_USER_DB_PATH the user already set is respected."""
    if "MIOPEN_USER_DB_PATH" in os.environ or not os.path.isdir(src):
        return False
    import shutil
    import tempfile
    dst = os.path.join(tempfile.gettempdir(), f"datr_miopen_userdb_{os.getuid()}_{os.getpid()}")
    try:
        os.makedirs(dst, exist_ok=True)
        for name in os.listdir(src):
            if name.endswith(".txt"):
                shutil.copyfile(os.path.join(src, name), os.path.join(dst, name))
    except OSError:
        return False
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    import atexit
    atexit.register(shutil.rmtree, dst, True)        # the per-process copy does not outlive the process
    return True


def enable(path: str = None, tune: bool = False) -> bool:
    """Turn TunableOp on with the shipped selections.  Returns False when the file is missing, does not
    validate against the installed torch / hipBLASLt, or this build has no TunableOp -- and then (unless
    DATR_GEMM_BACKEND=library) routes the large linear / FFN / 1x1-convolution products to the own GEMM family
    (datr_amd.gemm.set_backend("own")): hipBLASLt's default heuristic would run the FFN shapes at ~83 TF/s
    where the selections and the own kernels both reach ~130."""
    from .. import gemm
    forced = os.environ.get("DATR_GEMM_BACKEND", "auto")
    if path is None:
        path = os.environ.get("DATR_TUNING_FILE") or RESULTS
    ok = _enable(path, tune)
    if forced == "own":
        gemm.set_backend("own", "DATR_GEMM_BACKEND=own")
    elif forced == "library" or ok:
        # row counts the selections cover: large products with another row count go to the own family (gemm.TUNED_ROWS)
        rows = None
        if ok and forced != "library" and not tune and os.environ.get("DATR_GEMM_UNTUNED", "own") != "library":
            rows = tuned_row_counts(path, gemm.BIG_ROWS)
        gemm.set_backend("library", "hipBLASLt, per-shape selections of datr_amd/tuning"
                         + ("; own GEMM family for large products of other row counts" if rows is not None else "") if ok
                         else "DATR_GEMM_BACKEND=library: hipBLASLt default heuristic", rows)
    else:
        gemm.set_backend("own", "tuning file missing or not valid for this torch / hipBLASLt: own GEMM family")
    return ok


def tuned_row_counts(path: str, at_least: int):
    """Every GEMM dimension >= at_least that appears in the selections file (entries `op,tn_m_n_k_ld_...,solution,ms`):
    the token / pixel counts the selections were recorded for."""
    rows = set()
    try:
        for line in open(path):
            p = line.split(",")
            if len(p) < 4 or p[0] == "Validator":
                continue
            dims = p[1].split("_ld_")[0].split("_")[1:4]
            rows.update(int(d) for d in dims if d.isdigit() and int(d) >= at_least)
    except OSError:
        return None
    return rows


def _enable(path: str, tune: bool) -> bool:
    if torch.cuda.is_available() and os.environ.get("DATR_MIOPEN_DB", "0") == "1":
        enable_miopen_db()
    if not torch.cuda.is_available() or not hasattr(torch.cuda, "tunable"):
        return False
    t = torch.cuda.tunable
    if not os.path.exists(path) and not tune:
        return False
    t.enable(True)
    t.tuning_enable(bool(tune))
    if not tune and hasattr(t, "write_file_on_exit"):
        t.write_file_on_exit(False)      # lookup only: nothing to record, and N ranks must not race on a file
    if os.path.exists(path):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ok = t.read_file(path)
        if not ok and not tune:
            t.enable(False)         # validators (torch / hipBLASLt versions) do not match
            return False
    return True
