"""Host-side cost of the training step with R processes running concurrently on one host, as on the 8-GPU
node (one process per GPU): the GPU time of a shared single GPU is meaningless here, what is measured is
the CPU time a rank needs to ENQUEUE one step (time.process_time: all threads of the process) and the wall
time of the enqueue loop with the device left to run behind -- the numbers that decide whether R x launch
work fits beside an ~77 ms GPU step on the real node.

    python tools/probes/host_ranks.py --ranks 8 [--pin] [--steps 6]
Each rank is a separate process (`--worker`); --pin gives rank r the cores [r * c, (r + 1) * c),
c = cpu_count // ranks (taskset).  Prints one JSON line."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(steps):
    sys.path.insert(0, ROOT)
    import torch
    from datr_amd.training import Stepper, synthetic_batch
    dev = torch.device("cuda:0")
    tr = Stepper(dev)
    h, w = (int(v) for v in os.environ.get("HOST_RANKS_SIZE", "800x1333").split("x"))
    samples, targets = synthetic_batch(2, h, w, 10, dev, seed=1)
    for _ in range(3):
        tr.step(samples, targets)
    torch.cuda.synchronize()
    # barrier file: start the measured loop when every rank is warm
    flag = os.environ["HOST_RANKS_DIR"]
    open(os.path.join(flag, f"ready{os.environ['HOST_RANK']}"), "w").close()
    n = int(os.environ["HOST_RANKS"])
    while len([f for f in os.listdir(flag) if f.startswith("ready")]) < n:
        time.sleep(0.01)
    # time the host spends BLOCKED on the device (the loss fetch and the offset monitors wait on events):
    # enqueue wall time minus this = the host's own work per step, whatever the speed of the shared GPU
    blocked = [0.0]
    real_sync = torch.cuda.Event.synchronize

    def timed_sync(self):
        t = time.perf_counter()
        real_sync(self)
        blocked[0] += time.perf_counter() - t
    torch.cuda.Event.synchronize = timed_sync
    c0, w0 = time.process_time(), time.perf_counter()
    for _ in range(steps):
        tr.step(samples, targets)
    c1, w1 = time.process_time(), time.perf_counter()
    torch.cuda.synchronize()
    w2 = time.perf_counter()
    print(json.dumps({"rank": int(os.environ["HOST_RANK"]), "cpu_ms_per_step": (c1 - c0) / steps * 1e3,
                      "enqueue_wall_ms_per_step": (w1 - w0) / steps * 1e3,
                      "host_work_ms_per_step": (w1 - w0 - blocked[0]) / steps * 1e3,
                      "wall_ms_per_step_incl_gpu": (w2 - w0) / steps * 1e3}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--pin", action="store_true")
    ap.add_argument("--worker", action="store_true")
    ap.add_argument("--size", default="800x1333",
                    help="image size HxW; a small one (96x128) makes the step HOST-bound -- same ~2 100 launches, "
                         "GPU time of a few ms -- so that the wall time per step IS the host's enqueue work even "
                         "with every rank on one shared GPU")
    a = ap.parse_args()
    if a.worker:
        return worker(a.steps)
    import tempfile
    d = tempfile.mkdtemp()
    ncpu = os.cpu_count()
    per = max(1, ncpu // a.ranks)
    procs = []
    for r in range(a.ranks):
        env = dict(os.environ, HOST_RANK=str(r), HOST_RANKS=str(a.ranks), HOST_RANKS_DIR=d, HOST_RANKS_SIZE=a.size)
        cmd = [sys.executable, os.path.abspath(__file__), "--worker", "--steps", str(a.steps)]
        if a.pin:
            cmd = ["taskset", "-c", f"{r * per}-{(r + 1) * per - 1}"] + cmd
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    rows = []
    for p in procs:
        out, err = p.communicate(timeout=1200)
        rows += [json.loads(l) for l in out.splitlines() if l.startswith("{")]
        if p.returncode != 0 and not rows:
            print("worker failed:", err.strip().splitlines()[-3:], file=sys.stderr)
    if not rows:
        raise SystemExit("no rank reported")
    cpu = sorted(r["cpu_ms_per_step"] for r in rows)
    enq = sorted(r["enqueue_wall_ms_per_step"] for r in rows)
    hw = sorted(r["host_work_ms_per_step"] for r in rows)
    print(json.dumps({"ranks": a.ranks, "pinned": a.pin, "image_size": a.size, "host_cpus": ncpu, "cores_per_rank_when_pinned": per,
                      "steps": a.steps, "ranks_reporting": len(rows),
                      "cpu_ms_per_step": {"min": round(cpu[0], 1), "median": round(cpu[len(cpu) // 2], 1), "max": round(cpu[-1], 1)},
                      "host_work_ms_per_step": {"min": round(hw[0], 1), "median": round(hw[len(hw) // 2], 1), "max": round(hw[-1], 1)},
                      "enqueue_wall_ms_per_step": {"min": round(enq[0], 1), "median": round(enq[len(enq) // 2], 1), "max": round(enq[-1], 1)},
                      "wall_ms_per_step_incl_shared_gpu": round(max(r["wall_ms_per_step_incl_gpu"] for r in rows), 1)}))


if __name__ == "__main__":
    main()
