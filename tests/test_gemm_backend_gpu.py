"""The training step must not depend on the TunableOp selections file: when
`datr_amd/tuning/gemm_mi355x.csv` does not validate against the installed torch / hipBLASLt, `tuning.enable()`
returns False and the large linear / FFN / 1x1-convolution products go to the own fp32-MFMA GEMM family instead
of hipBLASLt's default heuristic (~83 TF/s on the 88 892-row FFN shapes, ~1.1 ms per launch).  Run in a
subprocess (TunableOp state is process-global): one full-size step under the kernel profiler with a selections
file whose validators cannot match; no library GEMM (`Cijk_*`) kernel may take more than 1 ms.
Mirrors /root/reference/models/dino/deformable_transformer.py:784-787,801-805 (the FFN those GEMMs are)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import json, sys, time, torch
sys.path.insert(0, %(root)r)
from datr_amd import gemm
from datr_amd.training import build_training, run_steps, synthetic_batch
dev = torch.device("cuda:0")
state = build_training(device=dev)
batch = synthetic_batch(2, 800, 1333, 10, dev, seed=1)
run_steps(state, [batch] * 3)
torch.cuda.synchronize()
t0 = time.perf_counter()
run_steps(state, [batch] * 4)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 4 * 1e3
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    stats = run_steps(state, [batch])
    torch.cuda.synchronize()
lib = {}
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA and e.name.startswith("Cijk_"):
        lib[e.name] = max(lib.get(e.name, 0.0), e.device_time)
print("RESULT " + json.dumps({"backend": gemm.BACKEND, "reason": gemm.BACKEND_REASON, "ms_per_step": ms,
                              "loss": stats["loss"], "tunable": torch.cuda.tunable.is_enabled(),
                              "library_gemm_kernels": len(lib), "slowest_library_gemm_us": max(lib.values(), default=0.0)}))
'''


def _run(env_extra):
    env = dict(os.environ, **env_extra)
    p = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], env=env, capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[7:])


def test_step_without_valid_selections_runs_the_big_gemms_on_the_own_family(tmp_path):
    bad = tmp_path / "stale.csv"
    src = open(os.path.join(ROOT, "datr_amd", "tuning", "gemm_mi355x.csv")).read().splitlines()
    bad.write_text("\n".join("Validator,PT_VERSION,0.0.0" if ln.startswith("Validator,PT_VERSION") else ln
                             for ln in src) + "\n")
    r = _run({"DATR_TUNING_FILE": str(bad)})
    assert r["backend"] == "own" and not r["tunable"], r
    assert r["loss"] == r["loss"] and r["loss"] > 0
    assert r["slowest_library_gemm_us"] < 1000.0, r           # no default-heuristic FFN GEMM in the step
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        json.dump(r, open(os.path.join(out, "gemm_backend_fallback.json"), "w"), indent=1)
    except OSError:
        pass
    print(r)
