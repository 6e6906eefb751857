import sys, torch
sys.path.insert(0, '.')
from chexpert_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16; B = 256
x4 = torch.randn(B, 320, 320, 4, device=dev).to(bf)
w = torch.randn(64, 3, 7, 7, device=dev)
wp = ops.pack_weights(w, stem=True)
y = torch.empty(B, 160, 160, 64, device=dev, dtype=bf)
st = torch.zeros(2, 64, device=dev)
def f(): ops.conv_gemm(x4, wp, y, N=64, mode=ops.MODE_STEM, stat_sum=st[0], stat_sq=st[1])
f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): f()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 5 * 1e3
print("stem fwd %.1f us  %.2f TB/s" % (us, (x4.numel() * 2 + y.numel() * 2) / us / 1e6))
