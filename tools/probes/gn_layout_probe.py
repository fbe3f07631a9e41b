"""Memory format of conv1x1 + GroupNorm(32, 256) outputs for a channels_last input, and what ATen's
group norm costs in either layout (the input_proj of dino.py:111-126 at the 100x167 level)."""
import torch

dev = torch.device("cuda:0")
x = torch.randn(4, 512, 100, 167, device=dev).contiguous(memory_format=torch.channels_last)
conv = torch.nn.Conv2d(512, 256, 1).to(dev)
gn = torch.nn.GroupNorm(32, 256).to(dev)
y = conv(x)
z = gn(y)
print("conv out channels_last:", y.is_contiguous(memory_format=torch.channels_last), "strides", y.stride())
print("gn out channels_last:", z.is_contiguous(memory_format=torch.channels_last), "contiguous:", z.is_contiguous(), z.stride())


def timed(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


yc = y.detach().contiguous()
yl = y.detach().contiguous(memory_format=torch.channels_last)
print("gn fwd NCHW %.0f us, channels_last %.0f us" % (timed(lambda: gn(yc)), timed(lambda: gn(yl))))
for t, name in ((yc, "NCHW"), (yl, "channels_last")):
    t = t.clone().requires_grad_(True)
    out = gn(t)
    go = torch.randn_like(out)
    print(name, "gn bwd %.0f us" % timed(lambda: torch.autograd.grad(out, (t, gn.weight, gn.bias), go, retain_graph=True)),
          "grad channels_last:", torch.autograd.grad(out, t, go, retain_graph=True)[0].is_contiguous(memory_format=torch.channels_last))
