"""Round 6 probe: the phased pyramid forward fed with PACKED PER-SAMPLE RECORDS {pixel, lh, lw, a} (16 B, produced here by
torch ops -- in a product they would come out of the prologue kernel) instead of locations + weights (12 B): what dropping
`locate` from the gather kernel is worth.  Needs two builds: the shipped library and one of msda_fwd_pyr2.hip with
-DPYR2_RECORDS=1 (tools/probes/r06_fwd_variants.sh build with PYR2_VARIANTS="records:-DPYR2_RECORDS=1").
    python tools/probes/r06_fwd_records.py [--dist model|gauss2.5]"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from datr_amd import msda  # noqa: E402
from bench_msda import make_inputs, time_fn, fwd_bytes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dist", default="model")
ap.add_argument("--iters", type=int, default=40)
a = ap.parse_args()
dev = torch.device("cuda:0")
value, sh, lsi, loc, attn = make_inputs(dev, 22223, a.dist, N=4)
N, S, M, D = value.shape
env = msda.measure_envelope(loc, sh)
sh_h = sh.cpu().numpy().astype(np.int64).copy()
lsi_h = lsi.cpu().numpy().astype(np.int64).copy()
env_h = np.ascontiguousarray(env, dtype=np.float32)

# records: the kernel's own arithmetic (csrc/msda_fwd_pyr2.hip::locate) in torch
wh = sh.flip(1).to(torch.float32).view(1, 1, 1, 4, 1, 2)              # (W, H) per level
im = loc * wh - 0.5                                                   # (w_im, h_im)
w_im, h_im = im[..., 0], im[..., 1]
Wf, Hf = wh[..., 0], wh[..., 1]
inside = (h_im > -1) & (w_im > -1) & (h_im < Hf) & (w_im < Wf)
hf, wf = torch.floor(h_im), torch.floor(w_im)
pix = (hf.to(torch.int32) << 16) | (wf.to(torch.int32) & 0xffff)
pix = torch.where(inside, pix, torch.full_like(pix, 0x7fffffff))
rec = torch.stack([pix.view(torch.float32), h_im - hf, w_im - wf, attn], -1).contiguous()   # [N, Lq, M, L, P, 4]
assert rec.shape[-1] == 4 and rec.dtype == torch.float32


def entry(path):
    lib = ctypes.CDLL(path)
    f = lib.datr_internal_msda_fwd_pyr2_d32
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int64] * 7 + [ctypes.c_void_p, ctypes.c_void_p]
    return f


def run(f, locp, attnp, out):
    st = torch.cuda.current_stream().cuda_stream
    rc = f(value.data_ptr(), locp, attnp, sh_h.ctypes.data, lsi_h.ctypes.data, env_h.ctypes.data, N, S, M, D, 4, S, 4,
           out.data_ptr(), st)
    assert rc == 0, rc


base = entry(os.path.join(ROOT, "datr_amd", "lib", "libdatr_hip.so"))
recs = entry(os.path.join(ROOT, "datr_amd", "lib", "libdatr_hip_f_records.so"))
o0, o1 = torch.empty(N, S, M * D, device=dev), torch.empty(N, S, M * D, device=dev)
run(base, loc.data_ptr(), attn.data_ptr(), o0)
run(recs, rec.data_ptr(), attn.data_ptr(), o1)
torch.cuda.synchronize()
err = (o0 - o1).abs()
print(f"records vs locations: max |diff| {float(err.max()):.3e} (outputs up to {float(o0.abs().max()):.3e}); "
      f"elements beyond 1e-6: {int((err > 1e-6).sum())} of {err.numel()}")
for rnd in range(3):
    t0 = time_fn(lambda: run(base, loc.data_ptr(), attn.data_ptr(), o0), a.iters)
    t1 = time_fn(lambda: run(recs, rec.data_ptr(), attn.data_ptr(), o1), a.iters)
    print(f"{a.dist} round {rnd}: locations + weights (12 B / sample) median {t0[0]:.1f} us (min {t0[1]:.1f});  "
          f"records (16 B / sample) median {t1[0]:.1f} us (min {t1[1]:.1f});  algorithmic {fwd_bytes(N, S, S) / 1e6:.1f} MB")
