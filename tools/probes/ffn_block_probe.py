"""Does row-blocking the FFN backward keep the hidden-gradient block in the 256 MB memory-side cache?
The encoder FFN backward at the step's size (88 892 tokens x 2048 hidden): dH = dS W2; dH *= [H > 0]
(+ bias sums); dW1 += dH^T X; dX = dS + dH W1 -- whole, and in 2 / 4 / 8 row blocks.
    python tools/probes/ffn_block_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from datr_amd import _native, tuning  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    tuning.enable()
    rows, C, F = 88892, 256, 2048
    g = torch.Generator(device=dev).manual_seed(0)
    x2 = torch.randn(rows, C, device=dev, generator=g)
    dsum = torch.randn(rows, C, device=dev, generator=g)
    h = torch.randn(rows, F, device=dev, generator=g).relu_()
    w1 = torch.randn(F, C, device=dev, generator=g) * 0.05
    w2 = torch.randn(C, F, device=dev, generator=g) * 0.05
    lib = _native.lib
    st = _native.current_stream_ptr(dev)

    def run(nblocks):
        dw1 = torch.zeros(F, C, device=dev)
        db1 = torch.zeros(F, device=dev)
        dx = torch.empty(rows, C, device=dev)
        step = (rows + nblocks - 1) // nblocks
        for r0 in range(0, rows, step):
            r1 = min(rows, r0 + step)
            ds, hb, xb = dsum[r0:r1], h[r0:r1], x2[r0:r1]
            dh = ds.mm(w2)
            n = r1 - r0
            nblk = int(lib.datr_relu_bwd_bias_partial_rows(n))
            part = torch.empty(nblk * F, device=dev)
            dbb = torch.empty(F, device=dev)
            rc = lib.datr_relu_bwd_bias_f32(dh.data_ptr(), hb.data_ptr(), n, F, part.data_ptr(), dbb.data_ptr(), st)
            assert rc == 0
            if nblocks == 1:
                dw1 = dh.t().mm(xb)
                db1 = dbb
            else:
                dw1.addmm_(dh.t(), xb)
                db1 += dbb
            torch.addmm(ds, dh, w1, out=dx[r0:r1])
        return dw1, db1, dx

    ref = run(1)
    for nb in (1, 2, 4, 8, 16):
        out = run(nb)
        err = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(out, ref))
        for _ in range(2):
            run(nb)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run(nb)
        e1.record()
        torch.cuda.synchronize()
        print(f"blocks {nb:2d}: {e0.elapsed_time(e1) / 5:.3f} ms per FFN backward (max rel diff vs whole {err:.1e})", flush=True)


if __name__ == "__main__":
    main()
