"""A/B of the GEMM family's epilogue access width (DATR_GEMM_WIDE_EPI=0/1, read once per process): the step's
epilogue-carrying launches, HIP events, median of 30 after 5 warm-up launches.
    for w in 0 1; do DATR_GEMM_WIDE_EPI=$w python tools/probes/gemm_epilogue_ab.py; done"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from datr_amd import gemm
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
R = lambda *s: torch.randn(*s, device=dev, generator=g)


def med(fn, n=30):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


rows = 88892
dy, w2, h = R(rows, 256), R(256, 2048) * 0.05, R(rows, 2048)
x, w1, b1 = R(rows, 256), R(2048, 256) * 0.05, R(2048)
cases = [("FFN dz: NN gate + colsum  [88892,256]x[256,2048]", 2.0 * rows * 256 * 2048, lambda: gemm.gemm_nn(dy, w2, gate=h, colsum=True)),
         ("NN plain                  [88892,256]x[256,2048]", 2.0 * rows * 256 * 2048, lambda: gemm.gemm_nn(dy, w2)),
         ("NT shift + relu           [88892,256]x[2048,256]", 2.0 * rows * 256 * 2048, lambda: gemm.gemm_nt(x, w1, shift=b1, relu=True)),
         ("NT shift (value proj)     [88892,256]x[256,256] ", 2.0 * rows * 256 * 256, lambda: gemm.gemm_nt(x, w1[:256], shift=b1[:256]))]
# backbone 1x1 convolutions at 4 x 200 x 334 / 100 x 167 / 50 x 84 pixels
for pix, cin, cmid in ((267200, 256, 64), (66800, 512, 128), (16800, 1024, 256)):
    xi, wa, sa = R(pix, cin), R(cmid, cin) * 0.05, R(cmid)
    y2, wc, sc, idn = R(pix, cmid), R(cin, cmid) * 0.05, R(cin), R(pix, cin)
    dz3, y2g, s2 = R(pix, cin), R(pix, cmid), torch.rand(cmid, device=dev, generator=g) + 0.5
    cases += [(f"conv1 NT shift+relu       [{pix},{cin}]x[{cmid},{cin}]", 2.0 * pix * cin * cmid, lambda xi=xi, wa=wa, sa=sa: gemm.gemm_nt(xi, wa, shift=sa, relu=True)),
              (f"conv3 NT shift+res+relu   [{pix},{cmid}]x[{cin},{cmid}]", 2.0 * pix * cin * cmid, lambda y2=y2, wc=wc, sc=sc, idn=idn: gemm.gemm_nt(y2, wc, shift=sc, residual=idn, relu=True)),
              (f"dz2 NN scale+gate         [{pix},{cin}]x[{cin},{cmid}]", 2.0 * pix * cin * cmid, lambda dz3=dz3, wc=wc, s2=s2, y2g=y2g: gemm.gemm_nn(dz3, wc, scale=s2, gate=y2g)),
              (f"dx NN res+gate            [{pix},{cmid}]x[{cmid},{cin}]", 2.0 * pix * cin * cmid, lambda y2=y2, wa=wa, idn=idn, xi=xi: gemm.gemm_nn(y2, wa, residual=idn, gate=xi))]
print("DATR_GEMM_WIDE_EPI =", os.environ.get("DATR_GEMM_WIDE_EPI", "1"))
for name, flops, fn in cases:
    us = med(fn)
    print(f"{name:58s} {us:8.1f} us  {flops / us / 1e6:6.1f} TF/s")
