"""The multi-rank leg of bench.py, executed before the driver's SCALE run does: `python bench.py --gpus 2`
started as a plain process re-launches itself under torch.distributed.run (the driver's command line), two
gloo ranks share the one GPU of the box (`--allow-gloo`, DATR_DIST_BACKEND=gloo), every rank trains on its
own batches through the flat-bucket reducer, the time is the max over the ranks, and rank 0 alone prints
the JSON line LAST on stdout (/root/reference/main.py:156, scripts/DINO_train_dist.sh:1 is the launch it
stands for)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra, timeout=900):
    env = dict(os.environ)
    env.update(env_extra)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
           "--padded-steps", "0", "--trained-like-steps", "0", "--height", "384", "--width", "512"] + extra
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + "\n" + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    return lines, p.stderr


@pytest.mark.gpu
def test_bench_two_gloo_ranks_share_the_gpu_and_rank0_prints_the_line_last():
    lines, err = _run(["--gpus", "2", "--allow-gloo"], {"DATR_DIST_BACKEND": "gloo"})
    line = json.loads(lines[-1])                          # the LAST stdout line is the JSON line
    assert sum(1 for ln in lines if ln.lstrip().startswith('{"metric"')) == 1, "exactly one rank prints"
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    cfg = line["config"]
    assert cfg["rccl_world"] == 2 and cfg["dist_backend"] == "gloo" and cfg["parallelism"] == "dp2"
    assert cfg["grad_reducer"] is True and cfg["global_batch_pairs"] == 4 and cfg["images_per_gpu"] == 4
    assert line["scaling"] == "weak" and line["higher_is_better"] is True and line["unit"] == "images/s"
    # whole-job aggregate: images of BOTH ranks over the max-over-ranks time
    assert line["value"] > 0 and line["ms_per_step"] > 0
    assert abs(line["value"] - 8 * 2 / (line["ms_per_step"] * 2 / 1e3)) / line["value"] < 0.02
    # the multi-rank diagnosis (one record per rank, gathered on rank 0): the rank's own step time, the host's
    # enqueue time and loss wait per step, the gradient exchange left exposed after backward, the core pinning
    ranks = line["ranks"]
    assert len(ranks) == 2
    for r in ranks:
        assert r["ms_per_step"] > 0 and r["ms_per_step"] <= line["ms_per_step"] * 1.05
        assert r["host_enqueue_ms"] > 0 and r["loss_wait_ms"] >= 0 and r["host_cpu_ms"] > 0
        assert r["finish_host_ms"] >= 0 and r["buckets"] >= 1 and r["allreduce_bytes"] > 190e6
        assert "exposed_allreduce_ms" in r and r["exposed_allreduce_ms"] >= 0
    aff = cfg["cpu_affinity"]
    assert aff is not None and ("pinned" in aff)


@pytest.mark.gpu
def test_bench_refuses_a_gloo_world_without_the_test_flag():
    env = dict(os.environ, DATR_DIST_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--no-cpu-baseline", "--height", "384", "--width", "512"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode != 0
    assert "RCCL" in p.stderr
