import sys, torch
sys.path.insert(0, "/root/repo")
from datr_amd import backbone, wino, tuning
tuning.enable()
dev = torch.device("cuda:0")
torch.manual_seed(1)
blk = backbone.Bottleneck(1024, 256, 1, backbone.FrozenBatchNorm2d, downsample=False).to(dev).to(memory_format=torch.channels_last)
x = torch.randn(4, 1024, 50, 84, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
go = torch.randn(4, 1024, 50, 84, device=dev).contiguous(memory_format=torch.channels_last)
params = [blk.conv1.weight, blk.conv2.weight, blk.conv3.weight]
from torch.profiler import profile, ProfilerActivity
for own in (True, False):
    wino.OWN_BACKBONE_3X3 = own
    for _ in range(3):
        torch.autograd.grad(blk(x), [x] + params, go)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        torch.autograd.grad(blk(x), [x] + params, go)
        torch.cuda.synchronize()
    print("==== own" if own else "==== lib")
    for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:14]:
        if e.device_time_total > 0:
            print(f"{e.count:3d} {e.device_time_total:9.1f} us  {e.key[:90]}")
