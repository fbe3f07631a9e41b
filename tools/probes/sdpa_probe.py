"""Decoder self-attention shape (merged source+target pass: N=4, 8 heads, 1100 queries, d=32, fp32,
boolean DN mask): which SDPA backend is fastest on MI355X, forward + backward."""
import torch
import torch.nn.functional as F
from torch.nn.attention import SDPBackend, sdpa_kernel

dev = torch.device("cuda:0")
torch.manual_seed(0)
N, H, L, D = 4, 8, 1100, 32
q, k, v = (torch.randn(N, H, L, D, device=dev, requires_grad=True) for _ in range(3))
mask = torch.zeros(L, L, dtype=torch.bool, device=dev)
mask[200:, :200] = True                      # matching queries do not see the DN part
for g in range(10):
    mask[g * 20:(g + 1) * 20, :g * 20] = True
    mask[g * 20:(g + 1) * 20, (g + 1) * 20:200] = True
fmask = torch.zeros(L, L, device=dev).masked_fill(mask, float("-inf"))
go = torch.randn(N, H, L, D, device=dev)


def bench(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def fwd_bwd(backend):
    def f():
        with sdpa_kernel(backend):
            o = F.scaled_dot_product_attention(q, k, v, attn_mask=fmask)
        o.backward(go)
        q.grad = k.grad = v.grad = None
    return f


def manual():
    s = (q @ k.transpose(-1, -2)) * (D ** -0.5) + fmask
    o = torch.softmax(s, -1) @ v
    o.backward(go)
    q.grad = k.grad = v.grad = None


for name, b in (("flash", SDPBackend.FLASH_ATTENTION), ("efficient", SDPBackend.EFFICIENT_ATTENTION),
                ("math", SDPBackend.MATH)):
    try:
        print(name, "fwd+bwd us:", round(bench(fwd_bwd(b)), 1))
    except Exception as ex:          # noqa: BLE001
        print(name, "unavailable:", str(ex)[:100])
print("manual bmm+softmax fwd+bwd us:", round(bench(manual), 1))
