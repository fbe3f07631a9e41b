"""One exhaustive MIOpen Find run of bench.py's step (torch.backends.cudnn.benchmark = True): fills
MIOpen's user find-db (~/.config/miopen or $MIOPEN_USER_DB_PATH), which datr_amd/tuning ships.
    python tools/probes/bench_cudnn_benchmark.py [extra bench.py flags, e.g. --channels-last]
"""
import os
import runpy
import sys

import torch

torch.backends.cudnn.benchmark = True
sys.argv = ["bench.py", "--steps", "20", "--warmup", "8", "--no-cpu-baseline"] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "bench.py"),
               run_name="__main__")
