"""Kernel-level timing of the MSDA HIP kernels at BASELINE's call shapes.

    python tools/bench_msda.py [--iters 50] [--dist uniform|model]

`uniform`: loc ~ U[0,1) over the whole image (the reference op test's recipe; worst case for
cache locality).  `model`: reference points on the pixel grid + the module's initial ring
offsets (what the encoder produces at initialisation).  Prints one JSON line per shape with
algorithmic GB/s (bytes defined in DESIGN.md / SURVEY.md 8d).
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datr_amd import msda  # noqa: E402

SHAPES = [(100, 167), (50, 84), (25, 42), (13, 21)]


def fwd_bytes(N, S, Lq, M=8, D=32, K=16):
    return 4 * N * (S * M * D + Lq * M * K * 3 + Lq * M * D)


def bwd_bytes(N, S, Lq, M=8, D=32, K=16):
    return 4 * N * (2 * S * M * D + Lq * (M * D + 2 * M * K * 2 + 2 * M * K))


def make_inputs(dev, Lq, dist, seed=3, N=2):
    g = torch.Generator(device="cpu").manual_seed(seed)
    shapes = torch.tensor(SHAPES, dtype=torch.int64)
    S = int(shapes.prod(1).sum())
    lsi = torch.cat([shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]])
    M, D, L, P = 8, 32, 4, 4
    value = torch.rand(N, S, M, D, generator=g) * 0.01
    attn = torch.softmax(torch.randn(N, Lq, M, L * P, generator=g), -1).view(N, Lq, M, L, P)
    if dist == "uniform":
        loc = torch.rand(N, Lq, M, L, P, 2, generator=g)
    elif dist.startswith("gauss") and Lq == S:
        # pixel-centre reference points + offsets ~ N(0, sigma px) in the target level; "gaussclip<sigma>": the same,
        # clamped to +-4.6 px (every sample inside the windows of the clipped envelope: the plan's cost without misses)
        clip = dist.startswith("gaussclip")
        sigma = float(dist[9 if clip else 5:] or 2.0)
        refs = []
        for h, w in SHAPES:
            ys, xs = torch.meshgrid(torch.linspace(0.5, h - 0.5, h) / h,
                                    torch.linspace(0.5, w - 0.5, w) / w, indexing="ij")
            refs.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
        ref = torch.cat(refs, 0).view(1, S, 1, 1, 1, 2)
        wh = torch.tensor([[w, h] for h, w in SHAPES], dtype=torch.float32).view(1, 1, 1, L, 1, 2)
        off = torch.randn(N, Lq, M, L, P, 2, generator=g) * sigma
        if clip:
            off = off.clamp(-4.6, 4.6)
        loc = (ref + off / wh).contiguous()
    else:
        # reference points: pixel centres of the pyramid (encoder) or uniform boxes (decoder)
        if Lq == S:
            refs = []
            for h, w in SHAPES:
                ys, xs = torch.meshgrid(torch.linspace(0.5, h - 0.5, h) / h,
                                        torch.linspace(0.5, w - 0.5, w) / w, indexing="ij")
                refs.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
            ref = torch.cat(refs, 0)[None].expand(N, -1, -1)
        else:
            ref = torch.rand(N, Lq, 2, generator=g)
        m = msda.MSDeformAttn(256, 4, 8, 4)
        off = m.sampling_offsets.bias.detach().view(1, 1, M, L, P, 2)
        wh = torch.tensor([[w, h] for h, w in SHAPES], dtype=torch.float32).view(1, 1, 1, L, 1, 2)
        loc = ref[:, :, None, None, None, :] + off / wh
        loc = loc.expand(N, Lq, M, L, P, 2).contiguous()
    return [t.to(dev).contiguous() for t in (value, shapes, lsi, loc, attn)]


def time_fn(fn, iters, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)   # us
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--dist", default="both")
    ap.add_argument("--n", type=int, default=2, help="batch items per call (the training step merges source + target: 4)")
    ap.add_argument("--fwd-only", action="store_true")
    ap.add_argument("--encoder-only", action="store_true", help="only the Lq = S call")
    ap.add_argument("--envelope", default="measured", choices=["none", "measured"],
                    help="window envelope of the phased pyramid forward: measured from the locations "
                         "(what datr_amd.msda.OffsetMonitor hands over in the model) or none (symmetric 4.5 px)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    dists = ["uniform", "model"] if args.dist == "both" else [args.dist]
    for dist in dists:
        for Lq in ((22223,) if args.encoder_only else (22223, 1100, 900)):
            value, sh, lsi, loc, attn = make_inputs(dev, Lq, dist, N=args.n)
            N, S = value.shape[0], value.shape[1]
            go = torch.randn(N, Lq, 256, device=dev)
            env, plan = None, None
            if Lq == S:
                if args.envelope == "measured":
                    env = msda.measure_envelope(loc, sh)
                plan = msda.pyramid_plan(sh, lsi, N, 8, 32, 4, env)
            kw = {} if env is None else {"envelope": env}
            f = lambda: msda.ms_deform_attn_forward(value, sh, lsi, loc, attn, 64, **kw)
            b = lambda: msda.ms_deform_attn_backward(value, sh, lsi, loc, attn, go, 64, **kw)
            fm, fmin = time_fn(f, args.iters)
            bm, bmin = (0.0, 0.0) if args.fwd_only else time_fn(b, args.iters)
            if os.environ.get("DATR_HIP_LIB", "").endswith("probe.so"):
                import ctypes
                from datr_amd import _native
                buf = (ctypes.c_ulonglong * 8)()
                _native.lib.datr_probe_phase_cycles(buf, 1)
                b()
                torch.cuda.synchronize()
                _native.lib.datr_probe_phase_cycles(buf, 1)
                tot = sum(buf) or 1
                print("phase cycles (thread 0 of every block, one bwd call): " + ", ".join(
                    f"{n}={100 * v / tot:.1f}%" for n, v in zip(
                        ["maxgo", "A:geom", "win-zero", "B:gather+add", "B-barrier", "C:flush"], buf)),
                    f"total={tot / 1e6:.1f} Mcycles")
            if os.environ.get("PYR_PROBE") and Lq == 22223:
                import ctypes
                from datr_amd import _native
                buf = (ctypes.c_ulonglong * 8)()
                _native.lib.datr_probe_pyr_phase_cycles(buf, 1)
                f()
                torch.cuda.synchronize()
                _native.lib.datr_probe_pyr_phase_cycles(buf, 1)
                tot = sum(buf) or 1
                print("pyr phase cycles (lane 0 of every wave, one call): " + ", ".join(
                    f"{n}={100 * v / tot:.1f}%" for n, v in zip(
                        ["fill-issue", "fill-land", "barrier", "locwait+geom", "gathers", "slow+store", "tail-wait"], buf)),
                    f"total={tot / 1e6:.1f} Mcycles")
            if os.environ.get("PYR2_PROBE") and Lq == 22223:
                import ctypes
                from datr_amd import _native
                buf = (ctypes.c_ulonglong * 8)()
                _native.lib.datr_probe_pyr2_phase_cycles(buf, 1)
                f()
                torch.cuda.synchronize()
                _native.lib.datr_probe_pyr2_phase_cycles(buf, 1)
                tot = sum(buf) or 1
                print("pyr2 phase cycles (lane 0 of every wave, one call): " + ", ".join(
                    f"{n}={100 * v / tot:.1f}%" for n, v in zip(
                        ["prologue", "refill-barrier", "fill-issue", "fill-land", "fill-barrier", "loc-land",
                         "geom+gather+blend", "stores"], buf)),
                    f"total={tot / 1e6:.1f} Mcycles")
                import numpy as np
                spans = np.zeros((8192, 4), dtype=np.uint64)
                _native.lib.datr_probe_pyr2_wg_spans(spans.ctypes.data_as(ctypes.c_void_p))
                sp = spans.astype(np.int64)
                sp = sp[sp[:, 0] > 0]                                  # workgroups that recorded a span
                t0, t1 = sp[:, 0].min(), sp[:, 1].max()
                dur = (sp[:, 1] - sp[:, 0]) / 100.0                      # us (100 MHz)
                # concurrency over time
                ev = np.concatenate([np.stack([sp[:, 0], np.ones(len(sp), np.int64)], 1),
                                     np.stack([sp[:, 1], -np.ones(len(sp), np.int64)], 1)])
                ev = ev[np.argsort(ev[:, 0], kind="stable")]
                conc = np.cumsum(ev[:, 1])
                dt = np.diff(ev[:, 0])
                avg_conc = float((conc[:-1] * dt).sum() / max(1, dt.sum()))
                clk = sp[:, 2] / np.maximum(dur, 1e-9) / 1e3                # GHz
                cu = (sp[:, 3] >> 32) * 1000 + ((sp[:, 3] >> 8) & 0xf) * 16 + ((sp[:, 3] >> 13) & 0x7) * 100
                print(f"wg spans: kernel {(t1 - t0) / 100.0:.1f} us, wg duration mean {dur.mean():.1f} us "
                      f"(min {dur.min():.1f}, max {dur.max():.1f}), mean concurrency {avg_conc:.0f} workgroups "
                      f"(peak {conc.max()}), shader clock {np.median(clk):.2f} GHz, distinct CU ids {len(np.unique(cu))}")
            if os.environ.get("PYR_PROBE_BWD") and Lq == 22223:
                import ctypes
                from datr_amd import _native
                buf = (ctypes.c_ulonglong * 8)()
                _native.lib.datr_probe_pyr_bwd_phase_cycles(buf, 1)
                b()
                torch.cuda.synchronize()
                _native.lib.datr_probe_pyr_bwd_phase_cycles(buf, 1)
                tot = sum(buf) or 1
                print("pyr bwd phase cycles: " + ", ".join(
                    f"{n}={100 * v / tot:.1f}%" for n, v in zip(
                        ["setup", "passA", "barrier+scan", "passB", "reduce+flush"], buf)),
                    f"total={tot / 1e6:.1f} Mcycles")
            print(json.dumps({
                "dist": dist, "Lq": Lq, "N": N, "pyr_fwd": msda.PYR_FORWARD, "plan": plan,
                "fwd_us_median": round(fm, 2), "fwd_us_min": round(fmin, 2),
                "fwd_GBps": round(fwd_bytes(N, S, Lq) / fm / 1e3, 1),
                "bwd_us_median": round(bm, 2), "bwd_us_min": round(bmin, 2),
                "bwd_GBps": round(bwd_bytes(N, S, Lq) / bm / 1e3, 1) if bm else None}), flush=True)


if __name__ == "__main__":
    main()
