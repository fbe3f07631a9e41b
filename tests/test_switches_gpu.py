"""Every A/B switch of the package (environment variables read at import) still runs on its
NON-default side: two training steps of the real detector at 384 x 512 through `engine.train_one_epoch` in a
fresh process per group of switches, finite loss each time, and the loss of the first step within 2 % of the
default configuration's (every switch selects another implementation of the same arithmetic).  The groups
put switches together that do not shadow one another."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import json, sys, torch
sys.path.insert(0, %(root)r)
from datr_amd.training import build_training, run_steps, synthetic_batch
dev = torch.device("cuda:0")
state = build_training(device=dev, channels_last=%(channels_last)s)
batch = synthetic_batch(2, 384, 512, 6, dev, seed=3, channels_last=%(channels_last)s)
losses = []
for _ in range(2):
    torch.manual_seed(7)          # the same de-noising draws in every process
    losses.append(run_steps(state, [batch])["loss"])
torch.cuda.synchronize()
print("RESULT " + json.dumps(losses))
'''

GROUPS = {
    "default": {},
    "fused-transformer-nodes-off": {"DATR_FUSED_FFN": "0", "DATR_FUSED_ADD_NORM": "0", "DATR_FUSED_PROLOGUE": "0",
                                    "DATR_FUSED_SELF_ATTN": "0", "DATR_FUSED_CLASS_SCORES": "0", "DATR_FUSED_CONTRAST": "0",
                                    "DATR_OWN_PROTOTYPES": "0", "DATR_FAN_OUT": "0"},
    "ffn-block-pieces": {"DATR_FUSED_FFN_BLOCK": "0", "DATR_FFN_FUSED_DZ": "0", "DATR_FFN_OWN_HIDDEN": "1",
                         "DATR_OWN_LINEAR_WGRAD": "0", "DATR_SELECTED_ROWS_BWD": "0"},
    "no-next-query-node": {"DATR_FUSED_NEXT_QUERY": "0", "DATR_NOGRAD_LINEAR_OWN_MIN_ROWS": "1000000000"},
    "decoder-layout-and-projections": {"DATR_BATCH_FIRST_DECODER": "0", "DATR_MERGE_QPROJ": "0", "DATR_VALUE_PROJ_BATCH": "0",
                                       "DATR_OWN_ATTENTION": "0"},
    "msda-routes": {"DATR_MSDA_PYR_FWD": "0", "DATR_MSDA_PYR_BWD": "0", "DATR_MSDA_ADAPTIVE": "0"},
    "msda-one-kernel-backward": {"DATR_MSDA_BWD_SPLIT": "0", "DATR_MSDA_PYR2": "0", "DATR_MSDA_PYRB_WIDEN": "0"},
    "msda-two-node-module": {"DATR_MSDA_QUERY_GRAD": "0"},
    "library-convolutions": {"DATR_OWN_BOTTLENECK": "0", "DATR_OWN_CONV3X3": "0", "DATR_OWN_CONV_S2": "0",
                             "DATR_OWN_D_IMG": "0", "DATR_OWN_CLASSIFIER": "0", "DATR_CONV1X1_GEMM": "0", "DATR_MIOPEN_DB": "1"},
    "per-op-backbone": {"DATR_OWN_BOTTLENECK": "0", "DATR_OWN_D_IMG_WGRAD": "0", "DATR_OVERLAP_D_IMG": "0",
                        "DATR_OWN_CONV3X3_MAX_CH": "64"},
    "own-gemm-backend": {"DATR_GEMM_BACKEND": "own", "DATR_OWN_BOTTLENECK_MIN_PIXELS": "1", "DATR_FREEZE_GC": "0"},
    "library-gemm-default-heuristic": {"DATR_GEMM_BACKEND": "library", "DATR_TUNING_FILE": "/nonexistent.csv"},
    "untuned-rows-on-the-library": {"DATR_GEMM_UNTUNED": "library"},
    "split-bf16-experiment": {"DATR_GEMM_SPLIT_BF16": "1"},
    "nchw-backbone": {"__channels_last": "False"},
    "one-rank-collectives": {"DATR_DIST_FORCE_COLLECTIVES": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29541"},
}
_default = {}


def _run(env_extra):
    env = dict(os.environ)
    cl = env_extra.get("__channels_last", "True")
    env.update({k: v for k, v in env_extra.items() if not k.startswith("__")})
    p = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT, "channels_last": cl}], env=env,
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])


@pytest.mark.parametrize("group", list(GROUPS))
def test_non_default_side_of_every_switch_still_runs(group):
    losses = _run(GROUPS[group])
    assert all(v == v and 0 < v < 1e4 for v in losses), losses
    if group == "default":
        _default["loss"] = losses
        return
    if "loss" not in _default:
        _default["loss"] = _run({})
    ref = _default["loss"][0]
    tol = 0.05 if group == "split-bf16-experiment" else 0.02
    assert abs(losses[0] - ref) <= tol * abs(ref), (group, losses, _default["loss"])
