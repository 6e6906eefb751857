"""Stem backward pair at the BASELINE batch: bnrelu_maxpool_bwd (2x2 pixel blocks) and the stem weight gradient (row strips),
us per launch and HBM bytes / time.  python scratch/bench_stembwd.py [B]"""
import os, sys, torch
sys.path.insert(0, '.')
from chexpert_amd import ops, _lib
dev = torch.device('cuda:0'); bf = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = W = 160
x = torch.randn(B, H, W, 64, device=dev).to(bf)
g = torch.randn(B, H // 2, W // 2, 256, device=dev).to(bf)
gx = torch.randn(B, H // 2, W // 2, 256, device=dev).to(bf)
amax = torch.randint(0, 9, (B, H // 2, W // 2, 64), device=dev, dtype=torch.uint8)
v = lambda: torch.rand(64, device=dev) + 0.5
sc, sh, mu, r, ga, gb, gc = v(), v() - 1, v() - 1, v(), v(), v() - 1, v() - 1
dz = torch.empty(B, H, W, 64, device=dev, dtype=bf)
rows = 2048
S = torch.zeros(2, rows, 64, device=dev)
x4 = torch.randn(B, 320, 320, 4, device=dev).to(bf)
dw = torch.zeros(64, 3, 7, 7, device=dev)
def t(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
f1 = lambda: ops.bnrelu_maxpool_bwd(x, sc, sh, mu, r, amax, g[..., :64], gx[..., :64], ga, gb, gc, dz, S[0], S[1], stat_rows=rows)
us = t(f1)
by = x.numel() * 2 * 2 + B * 80 * 80 * 64 * (2 + 2 + 1)
print("bnrelu_maxpool_bwd  B=%d  %.1f us  %.2f TB/s (%.2f GB)" % (B, us, by / us / 1e6, by / 1e9))
f2 = lambda: ops.conv_wgrad(dz, x4, dw, mode=ops.MODE_STEM, g_prologue=ops.PRO_AFFINE2, g2=x, ga=ga, gb=gb, gc=gc)
us = t(f2)
by = x.numel() * 2 * 2 + x4.numel() * 2
print("stem_wgrad (%s)  %.1f us  %.2f TB/s (%.2f GB)" % (_lib.lib().cx_last_kernel().decode(), us, by / us / 1e6, by / 1e9))
