"""bench.py -- headline benchmark of the DATR hot path on MI355X.

    python bench.py --gpus 1 --steps 8 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one iteration of `datr_amd.engine.train_one_epoch` -- the counterpart of the
reference's training loop (/root/reference/engine.py:54-111) -- on one synthetic batch:
model(samples, targets) -> criterion -> sum(loss*weight) -> reduce_dict for logging -> zero_grad ->
backward -> gradient all-reduce -> clip_grad_norm_(0.1) -> AdamW step -> loss dict to the host +
non-finite guard, for DINO-4scale R50 with
DATR's domain-adaptation branch (the reference has no switch that turns it off while training,
SURVEY.md 8d), batch_size 2 per GPU = 2 source + 2 target images of 1333x800, fp32.
Inputs are resident in HBM before the timed region.  Weak scaling: every rank runs the same
per-GPU batch; value = images of all ranks / max-over-ranks time.  `--gpus N` with N > 1 started
as a plain `python bench.py` re-launches itself under torch.distributed.run with N ranks.

Rank 0 prints ONE JSON line: the driver's contract fields plus
  roofline     -- MSDA forward (encoder call, the dominant HIP kernel family of this repo):
                  algorithmic bytes / mean launch duration measured with HIP
                  events on the launch stream inside the timed region, vs 8 TB/s HBM
  cpu_baseline -- the same training step on the host cores (PyTorch-CPU + oracle/msda_ref.c,
                  the counterpart of the reference's CPU fallback) on a bounded sample:
                  ONE step at 640x640, B=1 (BASELINE config 1's shape), N=1 runs only
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# multi-process GPU work on this platform needs dmabuf IPC (the host driver has no legacy IPC: without it RCCL fails
# with `hipIpcGetMemHandle: invalid argument`); the launcher's environment normally carries it already -- set it
# before the HIP runtime starts in case it does not
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 MFMA peak
PMC_RECORD = "r06_msda_pmc.json"  # raw FETCH_SIZE / WRITE_SIZE rows of the encoder forward launch
STEP_MFMA_RECORD = "r06_step_mfma.txt"   # in-step matrix-pipe busy per kernel family (committed PMC pass)


from datr_amd.training import build_training, run_steps, synthetic_batch  # noqa: E402


class MsdaTimer:
    """Brackets every MSDA forward launch whose Lq equals S (the encoder calls) with HIP
    events on the stream the kernel is launched on (torch's current stream)."""

    def __init__(self):
        from datr_amd import msda
        self.msda = msda
        self.orig = msda.ms_deform_attn_forward
        self.orig_bwd = msda.ms_deform_attn_backward
        self.events = []
        self.bwd_events = []
        self.shape = None
        self.phased = None
        self.enabled = False

    def install(self):
        def timed(value, shapes, lsi, loc, attn, step, **kw):
            if self.enabled and loc.shape[1] == value.shape[1]:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                out = self.orig(value, shapes, lsi, loc, attn, step, **kw)
                b.record()
                self.events.append((a, b))
                self.shape = (value.shape[0], value.shape[1], value.shape[2], value.shape[3],
                              loc.shape[3] * loc.shape[4], loc.shape[1])
                if self.phased is None:
                    env = kw.get("envelope")
                    self.phased = bool(kw.get("route", 0) == 0 and self.msda.pyramid_plan(
                        shapes, lsi, value.shape[0], value.shape[2], value.shape[3], loc.shape[4], env)["phased"])
                return out
            return self.orig(value, shapes, lsi, loc, attn, step, **kw)
        self.msda.ms_deform_attn_forward = timed

        def timed_bwd(value, shapes, lsi, loc, attn, grad_out, step, **kw):
            if self.enabled and loc.shape[1] == value.shape[1]:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                out = self.orig_bwd(value, shapes, lsi, loc, attn, grad_out, step, **kw)
                b.record()
                self.bwd_events.append((a, b))
                return out
            return self.orig_bwd(value, shapes, lsi, loc, attn, grad_out, step, **kw)
        self.msda.ms_deform_attn_backward = timed_bwd

        # the module's own entry to the same two kernels (grad_loc / grad_attn leave as the query projection's
        # gradient rows: the same bytes)
        self.orig_bwd_q = self.msda.ms_deform_attn_backward_query_grad

        def timed_bwd_q(value, shapes, lsi, loc, attn, grad_out, step, **kw):
            if not self.enabled:
                return self.orig_bwd_q(value, shapes, lsi, loc, attn, grad_out, step, **kw)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = self.orig_bwd_q(value, shapes, lsi, loc, attn, grad_out, step, **kw)
            b.record()
            if out is not None:
                self.bwd_events.append((a, b))
            return out
        self.msda.ms_deform_attn_backward_query_grad = timed_bwd_q

    def backward_result(self):
        """The encoder calls' backward (two kernels: LDS-window dots + value-free sorted scatter, plus the
        zero-fill of grad_value): algorithmic bytes of SURVEY.md 8d over the HIP-event time of the call."""
        if not self.bwd_events or self.shape is None:
            return None
        us = [a.elapsed_time(b) * 1e3 for a, b in self.bwd_events]
        N, S, M, D, K, Lq = self.shape
        algo = 4 * N * (2 * S * M * D + Lq * (M * D + 2 * M * K * 2 + 2 * M * K))
        mean_us = sum(us) / len(us)
        return {"bound": "hbm", "achieved": round(algo / mean_us / 1e3, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(algo / mean_us / 1e3 / HBM_PEAK_GBPS, 4), "traffic": None,
                "kernel": "msda_bwd_dots_pyr2_d32 + msda_bwd_pyr_d32 (+ grad_value zero-fill): the encoder "
                          "call's backward, csrc/msda_fwd_pyr2.hip + csrc/msda_bwd_pyr.hip",
                "launches": len(us), "mean_us": round(mean_us, 2), "algorithmic_bytes": algo}

    def result(self):
        if not self.events:
            return None
        us = [a.elapsed_time(b) * 1e3 for a, b in self.events]
        N, S, M, D, K, Lq = self.shape
        algo_bytes = 4 * N * (S * M * D + Lq * M * K * 3 + Lq * M * D)
        mean_us = sum(us) / len(us)
        # HBM traffic per launch from the PMC passes committed under profiles/ (same kernel, same
        # N = 4 launch; FETCH_SIZE corrected as the micro-architecture guide prescribes).  PMC
        # counters cannot be collected inside this process: null when the shape differs.
        traffic, kernel = None, "msda forward (encoder call)"
        traffic_src = None
        pmc = os.path.join(ROOT, "profiles", PMC_RECORD)
        if os.path.exists(pmc):
            # raw per-launch counter rows of the committed rocprofv3 PMC passes; the formula is
            # applied HERE so that the number reproduces from the committed rows:
            # bytes = (2 x mean FETCH_SIZE [KB] + mean WRITE_SIZE [KB]) x 1024
            rec = json.load(open(pmc))
            sh = rec["shape"]
            if (sh["N"], sh["S"], sh["M"], sh["D"], sh["Lq"]) == (N, S, M, D, Lq) and sh["L"] * sh["P"] == K \
                    and self.msda.PYR_FORWARD and rec.get("kernel") and \
                    (("pyr2" in rec["kernel"]) == bool(self.phased)):
                fetch, write = rec["FETCH_SIZE_KB_per_launch"], rec["WRITE_SIZE_KB_per_launch"]
                traffic = int(round((2.0 * sum(fetch) / len(fetch) + sum(write) / len(write)) * 1024))
                traffic_src = f"profiles/{PMC_RECORD}: (2 x mean FETCH_SIZE + mean WRITE_SIZE) x 1024, " \
                              f"{len(fetch)} + {len(write)} launches"
                pipes = rec.get("pipe_counters_mean_per_launch") or {}
                if all(k in pipes for k in ("SQ_ACTIVE_INST_VALU", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE")):
                    # What this instruction stream could reach if its busiest pipe never waited: the launch's time x the
                    # busy fraction of the vector ALUs / the LDS data path (committed PMC passes of the same launch:
                    # SQ_ACTIVE_INST_VALU x 4 cycles over 1024 SIMDs, SQ_LDS_IDX_ACTIVE over 256 CUs, against
                    # GRBM_GUI_ACTIVE / 8 elapsed cycles), and the PMC traffic at the HBM peak.
                    elapsed = pipes["GRBM_GUI_ACTIVE"] / 8.0
                    valu = 4.0 * pipes["SQ_ACTIVE_INST_VALU"] / 1024.0 / elapsed
                    lds = pipes["SQ_LDS_IDX_ACTIVE"] / 256.0 / elapsed
                    hbm_us = traffic / (HBM_PEAK_GBPS * 1e3)
                    ceiling_us = max(valu * mean_us, lds * mean_us, hbm_us)
                    self._ceiling = {
                        "valu_busy": round(valu, 3), "lds_busy": round(lds, 3),
                        "valu_us": round(valu * mean_us, 1), "lds_us": round(lds * mean_us, 1), "hbm_us": round(hbm_us, 1),
                        "ceiling_us": round(ceiling_us, 1), "achieved_over_ceiling": round(ceiling_us / mean_us, 3),
                        "bound_by": "valu" if ceiling_us == valu * mean_us else "lds" if ceiling_us == lds * mean_us else "hbm",
                        "source": f"profiles/{PMC_RECORD} pipe_counters_mean_per_launch (rocprofv3 PMC passes of the same "
                                  "launch shape; busy fractions applied to this run's mean_us)"}
        if self.msda.PYR_FORWARD and D == 32 and K == 16 and self.phased:
            kernel = "msda_fwd_pyr2_d32 (encoder call, csrc/msda_fwd_pyr2.hip: all levels out of LDS windows)"
        elif self.msda.PYR_FORWARD and D == 32 and K == 16:
            kernel = "msda_fwd_pyr_d32 (encoder call, csrc/msda_fwd_pyr.hip)"
        else:
            kernel = "msda_fwd_rows<8,16> (encoder call)"
        achieved = algo_bytes / mean_us / 1e3
        return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                "traffic_source": traffic_src, "kernel": kernel, "launches": len(us),
                "mean_us": round(mean_us, 2), "algorithmic_bytes": algo_bytes,
                "empty_event_pair_us": empty_event_pair_us(),
                "ceiling": getattr(self, "_ceiling", None)}


def empty_event_pair_us(n: int = 64):
    """What a HIP event pair reads with NOTHING between the two records, on the stream the kernels run on: the part
    of `mean_us` above that is bracketing, not kernel (reported beside it, never subtracted; the rocprofv3 tables
    under profiles/ hold the kernel's own duration)."""
    torch.cuda.synchronize()
    pairs = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        b.record()
        pairs.append((a, b))
    torch.cuda.synchronize()
    us = sorted(a.elapsed_time(b) * 1e3 for a, b in pairs)
    return round(us[len(us) // 2], 2)


def msda_rand_roofline(device, shape):
    """SURVEY.md 8d's micro-bench inputs for the same launch shape: value = rand * 0.01, loc ~ U[0, 1)
    over the whole image, attn = softmax(randn) -- the reference op test's recipe
    (/root/reference/models/dino/ops/test.py:28-37), the worst case for locality: no window plan helps,
    the call runs whatever kernel the library picks for it.  HIP events, mean of 10 launches."""
    from datr_amd import msda
    N, S, M, D, K, Lq = shape
    if (N, M, D, K, Lq) != (N, 8, 32, 16, S):
        return None
    g = torch.Generator(device="cpu").manual_seed(3)
    shapes = torch.tensor([(100, 167), (50, 84), (25, 42), (13, 21)], dtype=torch.int64)
    if int(shapes.prod(1).sum()) != S:
        return None
    lsi = torch.cat([shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]])
    value = (torch.rand(N, S, M, D, generator=g) * 0.01).to(device)
    loc = torch.rand(N, Lq, M, 4, 4, 2, generator=g).to(device)
    attn = torch.softmax(torch.randn(N, Lq, M, 16, generator=g), -1).view(N, Lq, M, 4, 4).to(device)
    shapes, lsi = shapes.to(device), lsi.to(device)
    f = lambda: msda.ms_deform_attn_forward(value, shapes, lsi, loc, attn, 64, route=1)
    for _ in range(3):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        f()
    b.record()
    b.synchronize()
    us = a.elapsed_time(b) * 1e3 / 10
    algo = 4 * N * (S * M * D + Lq * M * K * 3 + Lq * M * D)
    return {"bound": "hbm", "achieved": round(algo / us / 1e3, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(algo / us / 1e3 / HBM_PEAK_GBPS, 4), "traffic": None,
            "kernel": "msda_fwd_rows<8,16> (row kernel, corner rows through the vector-memory path)",
            "inputs": "loc ~ U[0,1) over the image (ops/test.py recipe), stand-alone launches after the timed region",
            "mean_us": round(us, 2), "algorithmic_bytes": algo}


def msda_trained_like_roofline(device, shape, sigma=2.5):
    """The same launch shape with offsets a TRAINED encoder plausibly has: every sample at its query's pixel
    centre + N(0, sigma px) in the sampled level, independently per (query, head, level, point) -- no two
    queries agree on an offset, unlike the ring initialisation the timed step runs with (where a head's
    offsets are one constant vector).  Envelope measured from the locations, as OffsetMonitor does in the
    model.  Forward and backward, HIP events, mean of 10 launches after 3 warm-up launches."""
    from datr_amd import msda
    N, S, M, D, K, Lq = shape
    shapes_l = [(100, 167), (50, 84), (25, 42), (13, 21)]
    if (M, D, K, Lq) != (8, 32, 16, S) or sum(h * w for h, w in shapes_l) != S:
        return None
    g = torch.Generator(device="cpu").manual_seed(5)
    shapes = torch.tensor(shapes_l, dtype=torch.int64)
    lsi = torch.cat([shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]])
    refs = []
    for h, w in shapes_l:
        ys, xs = torch.meshgrid(torch.linspace(0.5, h - 0.5, h) / h, torch.linspace(0.5, w - 0.5, w) / w, indexing="ij")
        refs.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
    ref = torch.cat(refs, 0).view(1, S, 1, 1, 1, 2)
    wh = torch.tensor([[w, h] for h, w in shapes_l], dtype=torch.float32).view(1, 1, 1, 4, 1, 2)
    loc = (ref + torch.randn(N, Lq, M, 4, 4, 2, generator=g) * sigma / wh).contiguous().to(device)
    value = (torch.rand(N, S, M, D, generator=g) * 0.01).to(device)
    attn = torch.softmax(torch.randn(N, Lq, M, 16, generator=g), -1).view(N, Lq, M, 4, 4).to(device)
    go = torch.randn(N, Lq, M * D, generator=g).to(device)
    shapes, lsi = shapes.to(device), lsi.to(device)
    env = msda.measure_envelope(loc, shapes)

    def timed(fn):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        b.record()
        b.synchronize()
        return a.elapsed_time(b) * 1e3 / 10
    fus = timed(lambda: msda.ms_deform_attn_forward(value, shapes, lsi, loc, attn, 64, envelope=env))
    bus = timed(lambda: msda.ms_deform_attn_backward(value, shapes, lsi, loc, attn, go, 64, envelope=env))
    algo = 4 * N * (S * M * D + Lq * M * K * 3 + Lq * M * D)
    algo_b = 4 * N * (2 * S * M * D + Lq * (M * D + 2 * M * K * 2 + 2 * M * K))
    return {"bound": "hbm", "achieved": round(algo / fus / 1e3, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(algo / fus / 1e3 / HBM_PEAK_GBPS, 4), "traffic": None,
            "kernel": "the library's pick for the measured envelope (pyramid kernels with wider windows / more phases)",
            "inputs": f"pixel-centre reference points + N(0, {sigma} px) offsets per (query, head, level, point); "
                      "stand-alone launches after the timed region",
            "mean_us": round(fus, 2), "algorithmic_bytes": algo,
            "backward": {"mean_us": round(bus, 2), "achieved": round(algo_b / bus / 1e3, 1),
                         "frac": round(algo_b / bus / 1e3 / HBM_PEAK_GBPS, 4), "algorithmic_bytes": algo_b}}


def msda_n2_launch_us(device):
    """The encoder forward launch at the CPU leg's shape -- N = 2, S = Lq = 22 223, M = 8, D = 32, L = P = 4 --
    MEASURED as its own launches (round 5 quoted the merged N = 4 launch / 2): (a) on the CPU leg's own inputs
    (`oracle.msda_oracle.random_inputs(seed=3)`: loc ~ U[0,1), the reference op test's recipe, moved to the
    device), (b) with the sampling locations a freshly initialised encoder produces (pixel-centre reference
    points + the ring offsets of ms_deform_attn.py:59-68: head m looks along angle 2 pi m / 8, point p at
    p + 1 pixels), the regime of the timed step.  HIP events, mean of 10 launches after 3 warm-up launches."""
    import math
    from datr_amd import msda
    from oracle import msda_oracle as O      # input generator of the CPU leg only; nothing of it is timed here
    shapes_l = [(100, 167), (50, 84), (25, 42), (13, 21)]
    S = sum(h * w for h, w in shapes_l)
    value, sh, lsi, loc, attn = O.random_inputs(2, S, 8, 32, shapes_l, 4, seed=3)
    value, loc, attn, sh, lsi = (x.to(device) for x in (value, loc, attn, sh, lsi))

    def timed(fn):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        b.record()
        b.synchronize()
        return round(a.elapsed_time(b) * 1e3 / 10, 2)
    out = {"same_inputs_as_cpu_leg_us": timed(lambda: msda.ms_deform_attn_forward(value, sh, lsi, loc, attn, 64))}
    refs = []
    for h, w in shapes_l:
        ys, xs = torch.meshgrid(torch.linspace(0.5, h - 0.5, h) / h, torch.linspace(0.5, w - 0.5, w) / w, indexing="ij")
        refs.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
    ref = torch.cat(refs, 0).view(1, S, 1, 1, 1, 2)
    th = torch.arange(8, dtype=torch.float32) * (2.0 * math.pi / 8)
    ring = torch.stack([th.cos(), th.sin()], -1)
    ring = (ring / ring.abs().max(-1, keepdim=True)[0]).view(1, 1, 8, 1, 1, 2) * \
        torch.arange(1, 5, dtype=torch.float32).view(1, 1, 1, 1, 4, 1)
    wh = torch.tensor([[w, h] for h, w in shapes_l], dtype=torch.float32).view(1, 1, 1, 4, 1, 2)
    loc_r = (ref + ring / wh).expand(2, S, 8, 4, 4, 2).contiguous().to(device)
    env = msda.measure_envelope(loc_r, sh)
    out["ring_offsets_us"] = timed(lambda: msda.ms_deform_attn_forward(value, sh, lsi, loc_r, attn, 64, envelope=env))
    return out


def trained_like_step_ms(state, pool, steps, sigma=2.5):
    """Step time with encoder offsets of a trained-like spread: every encoder layer's `sampling_offsets`
    gets weights ~ N(0, (sigma / 26)^2) and a zero bias, so that its offsets are ~ N(0, sigma px) and
    differ from query to query (the encoder's queries src + pos have a norm of ~26 in this model: with the
    divisor 26 the monitors report ~14 % of the samples beyond 4.5 px, the share N(0, 2.5 px) gives).  The layers' OffsetMonitors are
    reset, so each measures the new offsets at its first call and routes / sizes windows from its third call
    on (in a training run the monitors re-measure every 50th call); 12 warm-up steps.  Parameters and
    monitors are restored / reset afterwards."""
    from datr_amd import msda
    enc = [m for n, m in state.model.named_modules()
           if n.startswith("transformer.encoder.layers.") and n.endswith(".self_attn")]
    if not enc:
        return None
    saved = [(m.sampling_offsets.weight.detach().clone(), m.sampling_offsets.bias.detach().clone()) for m in enc]
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for m in enc:
            w = m.sampling_offsets.weight
            w.copy_((torch.randn(w.shape, generator=g) * (sigma / 26.0)).to(w.device))
            m.sampling_offsets.bias.zero_()
    try:
        msda._MONITORS.clear()
        run_steps(state, [pool[i % len(pool)] for i in range(12)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(state, [pool[i % len(pool)] for i in range(steps)])
        torch.cuda.synchronize()
        ms = round((time.perf_counter() - t0) / steps * 1e3, 2)
        routes = sorted({(mon.route, round(mon.fraction, 3)) for mon in msda._MONITORS.values()})
        return {"ms_per_step": ms, "steps": steps, "monitor_routes_and_far_fractions": routes}
    finally:
        msda._MONITORS.clear()
        with torch.no_grad():
            for m, (w, b) in zip(enc, saved):
                m.sampling_offsets.weight.copy_(w)
                m.sampling_offsets.bias.copy_(b)


def split_bf16_step_ms(state, pool, steps):
    """Step time with the own GEMM family's EXPERIMENTAL split-bf16 inner product switched on
    (DATR_GEMM_SPLIT_BF16=1, read per launch; see csrc/gemm_f32.hip `Split3`): 6 warm-up + `steps` timed steps
    after the timed region.  Reported beside the headline, never as `value`."""
    os.environ["DATR_GEMM_SPLIT_BF16"] = "1"
    try:
        # (the trained-like measurement before this one reset the OffsetMonitors: they re-measure in the first
        # step and switch plans in the third)
        run_steps(state, [pool[i % len(pool)] for i in range(6)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(state, [pool[i % len(pool)] for i in range(steps)])
        torch.cuda.synchronize()
        return {"ms_per_step": round((time.perf_counter() - t0) / steps * 1e3, 2), "steps": steps, "default": False}
    finally:
        del os.environ["DATR_GEMM_SPLIT_BF16"]


def config1_device_step(state, steps=6):
    """BASELINE configs[0]'s shape on the HIP path, beside the CPU leg's number at the same shape: ONE source
    image 640 x 640, DA branch off (S = 8 500), `steps` training steps through engine.train_one_epoch after 4
    warm-up steps, with the timed run's model / optimizer (run after every other measurement)."""
    was = state.model.domain_adaptation
    state.model.domain_adaptation = False
    try:
        b = synthetic_batch(1, 640, 640, 5, state.device, seed=1, source_only=True)
        run_steps(state, [b] * 4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(state, [b] * steps)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
    finally:
        state.model.domain_adaptation = was
    return {"value": round(1e3 / ms, 2), "unit": "images/s", "ms_per_step": round(ms, 2), "steps": steps,
            "sample": "the same step (1 source image 640x640, DA branch off) on the HIP path, one MI355X"}


def teacher_student_stage(args, device):
    """BASELINE config 5 on one GPU: datr_amd.engine.train_one_epoch_with_self_training (the
    reference's epoch function, engine.py:146-342) on synthetic batches -- EMA teacher forward
    on the target images, pseudo labels (threshold + class-wise NMS), student step on source +
    strongly augmented target, EMA update of the teacher.  Reference semantics kept: one
    `.item()` per step for the logged loss."""
    from datr_amd import tuning
    from datr_amd.config import c2f_args, get_param_dict
    from datr_amd.detector import build_dino
    from datr_amd.ema import ModelEMA
    from datr_amd.engine import train_one_epoch_with_self_training
    from datr_amd.nested import NestedTensor
    tuning.enable()
    cfg = c2f_args(device=str(device))
    # random-initialised weights score every box ~0.01-0.05: the C2F threshold (0.3) would leave
    # the target branch idle, so the synthetic run keeps boxes above 0.02 (stated in the line)
    cfg.pseudo_label_threshold = 0.02
    torch.manual_seed(0)
    model, criterion, _ = build_dino(cfg)
    model.to(device)
    model.backbone.to(memory_format=torch.channels_last)
    teacher = ModelEMA(model, decay=cfg.ema_decay_teacher)
    optimizer = torch.optim.AdamW(get_param_dict(cfg, model), lr=cfg.lr, weight_decay=cfg.weight_decay,
                                  fused=True)
    samples, targets = synthetic_batch(args.batch, args.height, args.width, args.num_gt, device, seed=1)
    g = torch.Generator().manual_seed(7)
    strong = NestedTensor((samples.tensors + 0.3 * torch.randn(samples.tensors.shape, generator=g).to(device))
                          .contiguous(memory_format=torch.channels_last), samples.mask, padded=False)
    meta = [{"image_id": torch.tensor([i]), "orig_size": torch.tensor([args.height, args.width]),
             "size": torch.tensor([args.height, args.width])} for i in range(args.batch)]
    batch = (samples, tuple(targets), tuple(meta), strong)

    def run(n):
        stats = train_one_epoch_with_self_training(model, teacher, criterion, [batch] * n, [batch] * n,
                                                   optimizer, device, 0, cfg.clip_max_norm, args=cfg)
        teacher.update(model)
        return stats
    run(args.warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats = run(args.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({
        "metric": "images/sec, teacher-student mutual-learning step (EMA teacher + student), bs=2/GPU, 1333x800",
        "value": round(2 * args.batch * args.steps / dt, 3), "unit": "images/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp32",
        "data": "synthetic",
        "config": {"workload": "BASELINE config 5 on one GPU: teacher forward + pseudo labels + student "
                               "step incl. target-domain criterion", "pseudo_label_threshold": 0.02,
                   "pseudo_labelled_images_per_step": stats.get("num_pseudo_images")}}), flush=True)


def mfma_utilisation(device, rows, experiments=True):
    """Matrix-core figures, against the dense fp32 MFMA peak (exact-fp32 `v_mfma_f32_32x32x2_f32`).
    Headline = the builder's OWN largest MFMA kernel of the step, measured live with HIP events after the
    timed region: the FFN hidden gradient dz = (dy W2) * [h > 0] + bias sums, [rows, 256] x [256, 2048],
    one launch of csrc/gemm_f32.hip per encoder layer (6 per step).  Secondary: the library GEMMs of the
    same FFN (hipBLASLt / rocBLAS) measured the same way, and the in-step matrix-pipe busy figures per
    kernel family from the committed PMC pass (not measurable inside this process)."""
    from datr_amd import gemm
    x = torch.randn(rows, 256, device=device)
    w1 = torch.randn(2048, 256, device=device) * 0.05
    w2 = torch.randn(256, 2048, device=device) * 0.05
    b1 = torch.zeros(2048, device=device)
    h = torch._addmm_activation(b1, x, w1.t(), use_gelu=False)
    dy = torch.randn(rows, 256, device=device)
    dh = torch.randn_like(h)
    flops = 2.0 * rows * 256 * 2048

    def tflops(fn):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        b.record()
        b.synchronize()
        return flops / (a.elapsed_time(b) / 10 * 1e-3) / 1e12
    own = tflops(lambda: gemm.gemm_nn(dy, w2, gate=h, colsum=True))
    own_plain = tflops(lambda: gemm.gemm_nn(dy, w2))
    lib = {"linear1 fwd (bias+ReLU epilogue)": tflops(lambda: torch._addmm_activation(b1, x, w1.t(), use_gelu=False)),
           "linear1 dgrad": tflops(lambda: dh.mm(w1)), "linear1 wgrad": tflops(lambda: dh.t().mm(x)),
           "linear2 dgrad (what the own kernel replaces, without the mask / bias-sum pass)": tflops(lambda: dy.mm(w2))}
    lib_mean = 3.0 / sum(1.0 / lib[k] for k in list(lib)[:3])
    # EXPERIMENT, off by default (DATR_GEMM_SPLIT_BF16=1; csrc/gemm_f32.hip `Split3`): the same launches with the
    # operands split exactly into three bf16 pieces and six bf16-MFMA products accumulated in fp32 -- speed, and
    # the error of both inner products against float64 on sampled rows, scaled by sum |a||b|
    split = split_plain = err_fp32 = err_split = None
    if experiments:                       # (off in the profiling runs: their per-family counters cover the product path only)
        idx = torch.randint(0, rows, (128,), device=device)
        ref = dy[idx].double() @ w2.double()
        err_scale = dy[idx].double().abs() @ w2.double().abs()
        err_fp32 = ((gemm.gemm_nn(dy, w2)[idx].double() - ref).abs() / err_scale).max().item()
        os.environ["DATR_GEMM_SPLIT_BF16"] = "1"
        try:
            split = tflops(lambda: gemm.gemm_nn(dy, w2, gate=h, colsum=True))
            split_plain = tflops(lambda: gemm.gemm_nn(dy, w2))
            err_split = ((gemm.gemm_nn(dy, w2)[idx].double() - ref).abs() / err_scale).max().item()
        finally:
            del os.environ["DATR_GEMM_SPLIT_BF16"]
    out = {"bound": "mfma", "achieved": round(own, 1), "peak": MFMA_FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": round(own / MFMA_FP32_PEAK_TFLOPS, 4),
           "kernel": f"gemm_f32_kernel (own, csrc/gemm_f32.hip): FFN hidden gradient with ReLU mask + bias-sum "
                     f"epilogue, M={rows} N=2048 K=256, measured live",
           "own_gemm_without_epilogue_tflops": round(own_plain, 1),
           # rounds 1-3 reported the harmonic mean of the three library FFN GEMMs under `achieved`; kept under its
           # own name so that round-to-round numbers stay comparable
           "r01_r03_headline_mean_of_three_library_gemms_tflops": round(lib_mean, 1),
           "library_gemm": {"achieved": round(lib_mean, 1), "frac": round(lib_mean / MFMA_FP32_PEAK_TFLOPS, 4),
                            "kernel": f"encoder FFN GEMMs, M={rows} N=2048 K=256 (hipBLASLt / rocBLAS, fp32), measured live",
                            "per_gemm_tflops": {k: round(v, 1) for k, v in lib.items()}},
           "experimental_split_bf16": None if split is None else {
               "default": False, "switch": "DATR_GEMM_SPLIT_BF16=1",
               "what": "the own GEMM family's inner product with both fp32 operands split exactly into three bf16 pieces "
                       "(24 mantissa bits = 3 x 8) and the six largest piece products accumulated in fp32 by "
                       "v_mfma_f32_32x32x16_bf16; NOT used for `value`",
               "achieved": round(split, 1), "without_epilogue_tflops": round(split_plain, 1),
               "max_error_over_sum_abs_products_vs_float64": {"fp32_mfma": float(f"{err_fp32:.3g}"),
                                                              "split_bf16_x6": float(f"{err_split:.3g}")}}}
    return out


def msda_cpu_ops(gpu_us=None):
    """SURVEY.md 8d 'CPU baseline beside it': the MSDA op at the encoder shape (N=2, 1333x800:
    S = Lq = 22 223, M=8, D=32, L=P=4) on the host cores -- the C restatement (oracle/msda_ref.c,
    OpenMP) and the torch `grid_sample` formulation the reference's own CPU path uses
    (ms_deform_attn_func.py:41-61) -- min of 3 after one warm-up call."""
    from oracle import msda_oracle as O
    shapes = [(100, 167), (50, 84), (25, 42), (13, 21)]
    S = sum(h * w for h, w in shapes)
    value, sh, lsi, loc, attn = O.random_inputs(2, S, 8, 32, shapes, 4, seed=3)
    go = torch.randn(2, S, 256, generator=torch.Generator().manual_seed(1))

    def best(fn, reps=3):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return round(min(ts) * 1e3, 1)

    def gs_bwd():
        v, l_, a = (x.clone().requires_grad_(True) for x in (value, loc, attn))
        O.msda_grid_sample(v, sh, l_, a).backward(go)
    out = {"shape": "N=2 S=Lq=22223 M=8 D=32 L=P=4 (encoder call)",
           "oracle_c_fwd_ms": best(lambda: O.msda_forward(value, sh, lsi, loc, attn)),
           "oracle_c_bwd_ms": best(lambda: O.msda_backward(value, sh, lsi, loc, attn, go)),
           "grid_sample_fwd_ms": best(lambda: O.msda_grid_sample(value, sh, loc, attn)),
           "grid_sample_fwd_bwd_ms": best(gs_bwd, reps=2),
           "threads": torch.get_num_threads()}
    if gpu_us is not None:
        out["hip_fwd_ms_same_shape"] = round(gpu_us / 1e3, 4)
    return out


def cpu_baseline(msda_gpu_us=None):
    """Reference-equivalent CPU step (torch CPU kernels + the oracle for MSDA / focal loss) on the host
    cores: ONE timed training step at 640x640, B=1 (2 images) after one un-timed warm-up step --
    BASELINE config 1's shape; like every training step of the reference it includes the
    domain-adaptation branch (there is no switch that turns it off).  Plus the MSDA
    op alone at the encoder shape (`msda_cpu_ops`)."""
    from datr_amd import msda
    from datr_amd.config import c2f_args, get_param_dict
    from datr_amd.detector import build_dino
    from oracle import msda_oracle as O

    def fwd(value, shapes, lsi, loc, attn, step):
        return O.msda_forward(value, shapes, lsi, loc, attn)

    def bwd(value, shapes, lsi, loc, attn, go, step):
        return list(O.msda_backward(value, shapes, lsi, loc, attn, go))

    from datr_amd import criterion as crit_mod
    from oracle import focal_oracle
    saved = (msda.ms_deform_attn_forward, msda.ms_deform_attn_backward, crit_mod.focal_loss_sums)
    msda.ms_deform_attn_forward, msda.ms_deform_attn_backward = fwd, bwd
    crit_mod.focal_loss_sums = focal_oracle.focal_sums_torch
    try:
        cfg = c2f_args(device="cpu")
        torch.manual_seed(0)
        model, criterion, _ = build_dino(cfg)
        model.train()
        criterion.train()
        opt = torch.optim.AdamW(get_param_dict(cfg, model), lr=cfg.lr, weight_decay=cfg.weight_decay)
        samples, targets = synthetic_batch(1, 640, 640, 5, torch.device("cpu"), seed=1)
        targets = list(targets)

        def step():
            out = model(samples, targets)
            loss_dict = criterion(out, targets)
            wd = criterion.weight_dict
            loss = sum(loss_dict[k] * wd[k] for k in loss_dict if k in wd)
            opt.zero_grad()
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.clip_max_norm)
            opt.step()
        step()                                  # warm-up (allocator, oneDNN primitive caches)
        t0 = time.perf_counter()
        step()
        dt = time.perf_counter() - t0
        # BASELINE config 1 as stated: ONE image 640x640, source-only (DA branch off)
        model.domain_adaptation = False
        samples, targets = synthetic_batch(1, 640, 640, 5, torch.device("cpu"), seed=1, source_only=True)
        targets = list(targets)
        step()
        t0 = time.perf_counter()
        step()
        dt1 = time.perf_counter() - t0
    finally:
        msda.ms_deform_attn_forward, msda.ms_deform_attn_backward, crit_mod.focal_loss_sums = saved
    return {"value": round(2.0 / dt, 4), "unit": "images/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"1 training step (after 1 warm-up step), 1 source + 1 target image 640x640 "
                      f"(BASELINE config 1 shape, DA branch included), {dt:.1f} s on "
                      f"{os.cpu_count()} host cpus",
            "config1_source_only": {"value": round(1.0 / dt1, 4), "unit": "images/s",
                                    "sample": f"1 training step (after 1 warm-up step), 1 source image "
                                              f"640x640, DA branch off (BASELINE config 1 as stated), "
                                              f"{dt1:.1f} s"},
            "msda_op": msda_cpu_ops(msda_gpu_us)}


def _gemm_backend(args):
    """Who ran the large linear / FFN / 1x1-convolution GEMMs of the timed step (datr_amd.gemm.BACKEND)."""
    from datr_amd import gemm
    if not args.tuned_gemm:
        return "hipBLASLt default heuristic (--no-tuned-gemm)"
    return {"library": "hipBLASLt", "own": "own fp32-MFMA GEMM family (csrc/gemm_f32.hip)"}[gemm.BACKEND] + \
        f" [{gemm.BACKEND_REASON}]"


def pin_to_local_cores(local_rank: int, local_world: int):
    """One process per GPU on one host: give every rank its own slice of the host cores, on the NUMA node
    its GPU hangs off when sysfs says which (the reference leaves placement to the launcher, main.py:138,156
    run under torch.distributed.launch; eight un-pinned ranks migrate and share caches).  Returns a
    description for the bench line; None when affinity cannot be set here."""
    try:
        avail = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return None
    if local_world <= 1 or len(avail) < local_world:
        return {"cpus": len(avail), "pinned": False}
    node, node_cpus = None, None
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node >= 0:
            cpus = []
            for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
                a, _, b = part.partition("-")
                cpus += list(range(int(a), int(b or a) + 1))
            node_cpus = [c for c in cpus if c in set(avail)]
    except (OSError, ValueError, AttributeError, AssertionError, RuntimeError):
        node, node_cpus = None, None
    per = max(1, len(avail) // local_world)
    mine = avail[local_rank * per:(local_rank + 1) * per]
    how = "contiguous slice of the allowed cores"
    if node_cpus and len(node_cpus) >= per:
        # ranks whose GPUs share the node split it: position among them by local rank order
        sharers = max(1, local_world * len(node_cpus) // len(avail))
        k = local_rank % sharers
        per_n = max(1, len(node_cpus) // sharers)
        mine = node_cpus[k * per_n:(k + 1) * per_n]
        how = f"NUMA node {node} of the rank's GPU"
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return {"cpus": len(avail), "pinned": False}
    torch.set_num_threads(max(1, min(len(mine), 8)))
    return {"cpus": len(mine), "first_cpu": mine[0], "pinned": True, "how": how, "numa_node": node}


def rank_diagnostics(state, step_diag, elapsed_local, steps, world, device):
    """What `bench.py --gpus N` says about WHERE a multi-rank step spends its time, per rank (gathered on
    rank 0): the rank's own time per step, the host's enqueue wall time per step (includes launch-queue
    back-pressure) and the process's CPU time per step, all threads (the loop is host-bound when THAT approaches the
    step time), how long the host then waits for the losses, and the span on the compute
    stream between "last bucket launched" and "every all-reduce done" in reducer.finish() -- the part of the
    gradient exchange NOT overlapped with backward."""
    mine = {"rank": dist.get_rank() if dist.is_initialized() else 0, "ms_per_step": round(elapsed_local / steps * 1e3, 2)}
    if step_diag:
        d = step_diag[-steps:]
        mine["host_enqueue_ms"] = round(sum(x["host_enqueue_ms"] for x in d) / len(d), 2)
        mine["loss_wait_ms"] = round(sum(x["loss_wait_ms"] for x in d) / len(d), 2)
        mine["host_cpu_ms"] = round(sum(x["host_cpu_ms"] for x in d) / len(d), 2)
        mine["host_cpu_main_thread_ms"] = round(sum(x["host_cpu_main_thread_ms"] for x in d) / len(d), 2)
    red = state.reducer
    if red is not None and red.diag:
        d = red.diag[-steps:]
        spans = [a.elapsed_time(b) for a, b in (x["events"] for x in d) if a is not None and b is not None]
        if spans:
            mine["exposed_allreduce_ms"] = round(sum(spans) / len(spans), 3)
        mine["finish_host_ms"] = round(sum(x["host_ms"] for x in d) / len(d), 3)
        mine["buckets_launched_in_finish"] = round(sum(x["buckets_launched_in_finish"] for x in d) / len(d), 2)
        mine["buckets"] = len(red.buckets)
        mine["allreduce_bytes"] = red.total_bytes
    if world > 1:
        # every rank reaches this point with the same control flow; a failure of the object gather is symmetric
        # (all ranks raise) and only costs the per-rank table
        try:
            out = [None] * world
            dist.all_gather_object(out, mine)
            return out
        except Exception as exc:  # noqa: BLE001
            mine["gather_error"] = repr(exc)[:200]
    return [mine]


def relaunch_with_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script under
    torch.distributed.run on this node (the command line the driver uses) and pass its output
    through."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=2, help="reference batch_size per GPU (image pairs)")
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=1333)
    ap.add_argument("--num-gt", type=int, default=10)
    ap.add_argument("--flat-grads", action="store_true", help="use the flat-bucket reducer at N=1 too")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--padded-steps", type=int, default=4,
                    help="after the timed region, also time this many steps of a batch that "
                         "needs padding (general path: masks, no shape-keyed caches); 0 = skip")
    ap.add_argument("--trained-like-steps", type=int, default=6,
                    help="after the timed region (N = 1), also run this many steps with encoder sampling offsets "
                         "~ N(0, 2.5 px) per query (what a trained encoder plausibly has) and report their "
                         "time beside the MSDA figures at those offsets; 0 = skip")
    ap.add_argument("--no-channels-last", dest="channels_last", action="store_false",
                    help="keep the backbone in NCHW (default: NHWC / torch.channels_last, which "
                         "MIOpen's measured-fastest fp32 solvers want; same arithmetic)")
    ap.add_argument("--no-tuned-gemm", dest="tuned_gemm", action="store_false",
                    help="leave hipBLASLt on its default heuristic (datr_amd/tuning)")
    ap.add_argument("--allow-gloo", action="store_true",
                    help="TEST flag: accept DATR_DIST_BACKEND=gloo for --gpus N > 1, so that the multi-rank leg "
                         "(self-relaunch, rank-0-only line, max-over-ranks time, per-rank batches) can be "
                         "executed by ranks SHARING one GPU (tests/test_bench_gpu.py); the line then says "
                         "config.dist_backend = gloo and is not a scaling measurement")
    ap.add_argument("--stage", choices=["burn-in", "source-only", "teacher"], default="burn-in",
                    help="teacher: the teacher-student stage (BASELINE config 5) on one GPU; "
                         "source-only: BASELINE config 2 read literally -- the burn-in step with the "
                         "DA branch off (model.domain_adaptation = False: B source images, no "
                         "target pass, no D_img / prototypes); the default is the headline burn-in "
                         "step, which is the reference's real one (DA branch on, config 3)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_with_ranks(args.gpus)
    if args.stage == "teacher":
        device = torch.device("cuda", 0)
        torch.cuda.set_device(device)
        return teacher_student_stage(args, device)

    from datr_amd.dist import FORCE_COLLECTIVES, init_distributed
    rank, local_rank, world = init_distributed()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if world > 1:
        assert dist.is_initialized() and dist.get_world_size() == world, "process group not up"
        assert dist.get_backend() == "nccl" or (args.allow_gloo and dist.get_backend() == "gloo"), \
            "the multi-GPU bench runs over RCCL (torch.distributed backend 'nccl'); --allow-gloo is for tests"
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    affinity = None
    if world > 1:
        try:            # placement is a courtesy: never let it end a scaling run
            affinity = pin_to_local_cores(int(os.environ.get("LOCAL_RANK", "0")),
                                          int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
        except Exception as exc:  # noqa: BLE001
            affinity = {"pinned": False, "error": repr(exc)[:200]}

    state = build_training(device=device, rank=rank, channels_last=args.channels_last,
                           tuned_gemm=args.tuned_gemm,
                           reducer=True if (world > 1 or FORCE_COLLECTIVES or args.flat_grads) else False)
    source_only = args.stage == "source-only"
    if source_only:
        state.model.domain_adaptation = False
    # four different batches of the workload's shape, cycled (a data loader's batches differ)
    pool = [synthetic_batch(args.batch, args.height, args.width, args.num_gt, device,
                            seed=1 + 7 * rank + 1000 * i, channels_last=args.channels_last,
                            source_only=source_only)
            for i in range(4)]
    timer = MsdaTimer()
    timer.install()

    def batches(n, offset=0):
        return [pool[(offset + i) % len(pool)] for i in range(n)]

    step_marks = []

    def marked(bs):
        """The batches, with a HIP event recorded on the launch stream at the start of every step and after
        the last one: consecutive marks are one step of GPU time apart in steady state."""
        for b_ in bs:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            step_marks.append(ev)
            yield b_
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        step_marks.append(ev)

    run_steps(state, batches(args.warmup))

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    timer.enabled = True
    # per-step host / reducer diagnostics: two clock reads per step and, with a reducer, one event pair
    from datr_amd import engine as engine_mod
    step_diag = engine_mod.STEP_DIAG = []
    if state.reducer is not None:
        state.reducer.diag = []
    t0 = time.perf_counter()
    run_steps(state, marked(batches(args.steps, args.warmup)))
    torch.cuda.synchronize()
    elapsed_local = time.perf_counter() - t0
    fence()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    engine_mod.STEP_DIAG = None
    ranks = rank_diagnostics(state, step_diag, elapsed_local, args.steps, world, device)
    if state.reducer is not None:
        state.reducer.diag = None
    per_step_raw = [a.elapsed_time(b) for a, b in zip(step_marks[:-1], step_marks[1:])]
    per_step = sorted(per_step_raw)

    def pct(q):
        return round(per_step[min(len(per_step) - 1, int(q * len(per_step)))], 2) if per_step else None

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    padded_ms = None
    if args.padded_steps > 0 and world == 1 and not source_only:
        # the same workload as a batch of different-sized images would arrive: images of
        # (H-32) x W inside the H x W batch tensor, mask True on the padding
        pb = synthetic_batch(args.batch, args.height - 32, args.width, args.num_gt, device, seed=99,
                             channels_last=args.channels_last, pad_to=(args.height, args.width))
        run_steps(state, [pb] * 2)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_steps(state, [pb] * args.padded_steps)
        torch.cuda.synchronize()
        padded_ms = round((time.perf_counter() - t1) / args.padded_steps * 1e3, 2)

    if rank == 0:
        per_gpu = args.batch if source_only else 2 * args.batch
        images = per_gpu * world * args.steps
        roof = timer.result()
        line = {
            "metric": ("images/sec, DINO-4scale R50 source-only training step (DA branch off), bs=2/GPU, "
                       "1333x800" if source_only else
                       "images/sec, DINO-4scale R50 + DATR DA branch training step, bs=2/GPU, 1333x800"),
            "value": round(images / elapsed, 3), "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2),
            "ms_per_step_percentiles": {"median": pct(0.5), "p10": pct(0.1), "p90": pct(0.9),
                                        "max": pct(0.9999),
                                        "source": "HIP events at every step boundary of the timed region (rank 0)"},
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "fp32", "data": "synthetic",
            "config": {"workload": ("BASELINE config 2 read literally: DINO-4scale R50 bs=2 1333x800 burn-in "
                                    "step with the DA branch off (2 source images, no target pass) "
                                    "through engine.train_one_epoch" if source_only else
                                    "DINO-4scale R50 bs=2 1333x800 burn-in step through "
                                    "engine.train_one_epoch (fwd + SetCriterion + reduce_dict + bwd + "
                                    "grad all-reduce + clip + AdamW + loss fetch/guard; source+target "
                                    "pass, D_img, prototypes)"),
                       "images_per_gpu": per_gpu, "global_batch_pairs": args.batch * world,
                       "parallelism": f"dp{world}", "num_gt_per_image": args.num_gt,
                       "grad_reducer": state.reducer is not None,
                       "gemm_backend": _gemm_backend(args),
                       "rccl_world": dist.get_world_size() if dist.is_initialized() else 1,
                       "dist_backend": dist.get_backend() if dist.is_initialized() else None,
                       "batch_seeds_rank0": [1 + 1000 * i for i in range(len(pool))],
                       "batch_seed_rule": "1 + 7 * rank + 1000 * i (every rank trains on different batches)"},
            "pairs_per_sec": round(images / elapsed / (1 if source_only else 2), 3),
            "ms_per_step_each": [round(v, 1) for v in per_step_raw] if os.environ.get("DATR_BENCH_PER_STEP") else None,
            "padded_batch_ms_per_step": padded_ms,
            "roofline": roof,
            # 7.17 TFLOP of convolution / linear work per step at B = 2 (fwd 1253.5 + bwd 2331.0
            # GFLOP per image pair, counted on the reference) against the dense fp32-MFMA peak: the figure that
            # describes the ~93 % of the step's GPU time that is matrix work
            "step_roofline": None if source_only else {
                "bound": "mfma", "flops": 3584.5e9 * args.batch, "unit": "TFLOP/s", "peak": MFMA_FP32_PEAK_TFLOPS,
                "achieved": round(3584.5e9 * args.batch / (elapsed / args.steps) / 1e12, 2),
                "frac": round(3584.5e9 * args.batch / (elapsed / args.steps) / 1e12 / MFMA_FP32_PEAK_TFLOPS, 4),
                "note": "valid at 1333x800 only (the FLOP count is for that size); per GPU"} if
            (args.height, args.width) == (800, 1333) else None,
            "ranks": ranks,
        }
        line["config"]["cpu_affinity"] = affinity
        line["roofline_backward"] = timer.backward_result()
        if timer.shape is not None:
            line["roofline_rand_locations"] = msda_rand_roofline(device, timer.shape)
            if world == 1 and not source_only and args.trained_like_steps > 0:
                line["roofline_trained_like"] = msda_trained_like_roofline(device, timer.shape)
                timer.enabled = False
                line["trained_like_offsets_ms_per_step"] = trained_like_step_ms(state, pool, args.trained_like_steps)
                line["experimental_split_bf16_ms_per_step"] = split_bf16_step_ms(state, pool, args.trained_like_steps)
            line["mfma"] = mfma_utilisation(device, timer.shape[0] * timer.shape[1], experiments=args.trained_like_steps > 0)
            # the whole step's matrix-pipe utilisation from the committed PMC pass (every launch of
            # five steps), beside the isolated figure above
            mf = os.path.join(ROOT, "profiles", STEP_MFMA_RECORD)
            if os.path.exists(mf):
                fam = {}
                for ln in open(mf):
                    if ln.startswith("#") or ln.startswith("family") or not ln.strip():
                        continue
                    name = ln[:58].strip()
                    busy = float(ln.split()[-1].rstrip("%")) / 100.0
                    if name.startswith("all kernels of the run"):
                        line["mfma"]["in_step_matrix_pipe_busy"] = busy
                    else:
                        fam[name] = busy
                line["mfma"]["in_step_matrix_pipe_busy_by_family"] = fam
                line["mfma"]["in_step_source"] = (f"COMMITTED record profiles/{STEP_MFMA_RECORD} (rocprofv3 PMC "
                                                  "SQ_VALU_MFMA_BUSY_CYCLES of every launch of five steps, "
                                                  "tools/pmc_step_mfma.sh) -- not measured in this run")
        if world == 1 and not args.no_cpu_baseline and not source_only:
            timer.enabled = False
            hip_c1 = config1_device_step(state)
            if dist.is_initialized():       # one-rank RCCL mode: the CPU leg must not see a NCCL group
                dist.destroy_process_group()
            # GPU time of the op at the CPU leg's shape (N = 2): measured launches of that shape
            n2 = msda_n2_launch_us(device)
            line["cpu_baseline"] = cpu_baseline(n2["ring_offsets_us"])
            line["cpu_baseline"]["msda_op"]["hip_fwd_ms_same_inputs"] = round(n2["same_inputs_as_cpu_leg_us"] / 1e3, 4)
            line["cpu_baseline"]["msda_op"]["hip_fwd_source"] = (
                "measured N = 2 launches (HIP events, mean of 10): `hip_fwd_ms_same_shape` with the ring-initialised "
                "sampling offsets of the timed step, `hip_fwd_ms_same_inputs` on the CPU leg's own tensors "
                "(loc ~ U[0,1): the row kernel)")
            line["cpu_baseline"]["config1_source_only"]["hip_same_shape"] = hip_c1
        # RCCL writes its version banner through C stdio, which sits in a buffer when stdout is a pipe
        # and would come out AFTER this line at exit: push it out first, the JSON line stays the last one
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
