import copy, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from test_bottleneck_gpu import _make_stage, _fold, _reference_stage
from datr_amd import bottleneck

dev = torch.device("cuda:0")
for (inpl, pl, nb, st, shape) in [(256, 128, 1, 2, (2, 67, 90)), (512, 128, 1, 1, (2, 40, 44)), (256, 128, 2, 2, (2, 67, 90)), (256, 128, 3, 2, (2, 67, 90))]:
    for own in (True, False):
        bottleneck.MIN_PIXELS = 1 if own else 1 << 30
        stage = _make_stage(inpl, pl, nb, st, dev, seed=5).train()
        if st == 1 and nb == 1:
            # identity block alone
            from datr_amd.backbone import Bottleneck, BottleneckStage, FrozenBatchNorm2d
            torch.manual_seed(5)
            stage = BottleneckStage(Bottleneck(inpl, pl, 1, FrozenBatchNorm2d, downsample=False)).to(dev).train()
        N, H, W = shape
        g = torch.Generator().manual_seed(7)
        x = torch.relu(torch.randn(N, inpl, H, W, generator=g)).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        ref_stage = copy.deepcopy(stage).double()
        xr = x.detach().double().requires_grad_(True)
        yr = _reference_stage(list(ref_stage), xr)
        go = torch.randn(yr.shape, generator=g).to(dev)
        yr.backward(go.double())
        _fold(stage, x)
        y = stage(x)
        y.backward(go.contiguous(memory_format=torch.channels_last))
        def err(a, b):
            return float((a.detach().double() - b).abs().max()) / max(float(b.abs().max()), 1e-30)
        out = {"y": err(y, yr), "x.grad": err(x.grad, xr.grad)}
        for (n, p), (_, pr) in zip(stage.named_parameters(), ref_stage.named_parameters()):
            out[n] = err(p.grad, pr.grad)
        print(f"cfg {(inpl, pl, nb, st)} own={own}: " + " ".join(f"{k}={v:.1e}" for k, v in out.items()), flush=True)
