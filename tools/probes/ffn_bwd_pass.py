import sys, torch
sys.path.insert(0, "/root/repo")
from datr_amd import fused, _native
dev = torch.device("cuda:0")
rows, cols = 88892, 2048
h = torch.relu(torch.randn(rows, cols, device=dev))
dh = torch.randn(rows, cols, device=dev)
db = torch.empty(cols, device=dev)
nblk = int(_native.lib.datr_relu_bwd_bias_partial_rows(rows))
partial = torch.empty(nblk * cols, device=dev)
def run():
    rc = _native.lib.datr_relu_bwd_bias_f32(dh.data_ptr(), h.data_ptr(), rows, cols, partial.data_ptr(), db.data_ptr(), _native.current_stream_ptr(dev))
    assert rc == 0
for _ in range(3): run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): run()
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) / 20 * 1e3
print(f"relu_bwd_bias [{rows}x{cols}]: {us:.0f} us, {3 * rows * cols * 4 / us / 1e6:.2f} TB/s")
x = torch.randn(rows, 256, device=dev)
for _ in range(3): fused.column_sums(x)
torch.cuda.synchronize(); a.record()
for _ in range(20): fused.column_sums(x)
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) / 20 * 1e3
print(f"column_sums [{rows}x256]: {us:.0f} us, {rows * 256 * 4 / us / 1e6:.2f} TB/s")
# correctness vs torch
dh2 = torch.randn(4000, 2048, device=dev); h2 = torch.relu(torch.randn(4000, 2048, device=dev))
want = dh2 * (h2 > 0); wb = want.sum(0)
p2 = torch.empty(int(_native.lib.datr_relu_bwd_bias_partial_rows(4000)) * 2048, device=dev); db2 = torch.empty(2048, device=dev)
_native.lib.datr_relu_bwd_bias_f32(dh2.data_ptr(), h2.data_ptr(), 4000, 2048, p2.data_ptr(), db2.data_ptr(), _native.current_stream_ptr(dev))
print("exact dz:", torch.equal(dh2, want), "db err", float((db2 - wb).abs().max()))
