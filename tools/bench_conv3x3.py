"""Hand-written MFMA conv3x3 NHWC (libdatr_hip.so) vs torch/MIOpen (channels_last, shipped find-db)
on the discriminator's shapes."""
import os, sys, json
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datr_amd import _native, tuning
tuning.enable()

def mine(x, wt, b, Cout, slope):
    N, Cin, H, W = x.shape                      # channels_last storage
    y = torch.empty(N, Cout, H, W, device=x.device).contiguous(memory_format=torch.channels_last)
    rc = _native.lib.datr_conv3x3_nhwc_forward_f32(x.data_ptr(), wt.data_ptr(), 0 if b is None else b.data_ptr(),
                                                   N, H, W, Cin, Cout, slope, 1.0, y.data_ptr(),
                                                   _native.current_stream_ptr(x.device))
    _native.check(rc, "conv3x3_nhwc")
    return y

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3

dev = torch.device("cuda:0")
torch.manual_seed(0)
for (Cin, Cout) in ((256, 256), (256, 128), (128, 128)):
    for (H, W) in ((100, 167), (50, 84), (25, 42), (13, 21)):
        x = torch.randn(4, Cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02).contiguous(memory_format=torch.channels_last)
        b = torch.randn(Cout, device=dev)
        wt = w.permute(2, 3, 1, 0).contiguous()
        ref = F.leaky_relu(F.conv2d(x, w, b, padding=1), 0.2)
        out = mine(x, wt, b, Cout, 0.2)
        err = (out - ref).abs().max().item()
        t_ref = timeit(lambda: F.leaky_relu(F.conv2d(x, w, b, padding=1), 0.2))
        t_conv = timeit(lambda: F.conv2d(x, w, b, padding=1))
        t_mine = timeit(lambda: mine(x, wt, b, Cout, 0.2))
        gf = 2 * 4 * H * W * Cout * Cin * 9 / 1e9
        print(json.dumps({"Cin": Cin, "Cout": Cout, "HW": [H, W], "max_err": round(err, 7),
                          "miopen_conv_us": round(t_conv, 1), "miopen_conv_lrelu_us": round(t_ref, 1),
                          "mine_us": round(t_mine, 1), "miopen_TF": round(gf / t_conv, 1),
                          "mine_TF": round(gf / t_mine, 1)}), flush=True)
