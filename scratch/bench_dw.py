"""Micro-benchmark of cx_dwconv_fwd / _dgrad / _wgrad on the EfficientNet-B4 @380 bs=64 depthwise layer shapes.
usage: python scratch/bench_dw.py [which ...]   (which in fwd dgrad wgrad)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chexpert_amd._lib import lib, ptr, check, stream_ptr

# (k, stride, H_in, C, count)  -- expanded widths of B4 (width 1.4, depth 1.8), 380x380 input
SHAPES = [(3, 1, 190, 48, 1), (3, 1, 190, 24, 1), (3, 2, 190, 144, 1), (3, 1, 95, 192, 3), (5, 2, 95, 192, 1), (5, 1, 48, 336, 3),
          (3, 2, 48, 336, 1), (3, 1, 24, 672, 5), (5, 1, 24, 672, 1), (5, 1, 24, 960, 5), (5, 2, 24, 960, 1), (5, 1, 12, 1632, 7),
          (3, 1, 12, 1632, 1), (3, 1, 12, 2688, 1)]
B = 64
if os.environ.get("DW_ONLY"):
    SHAPES = [SHAPES[int(i)] for i in os.environ["DW_ONLY"].split(",")]
dev = torch.device("cuda:0")
which = sys.argv[1:] or ["fwd", "dgrad", "wgrad"]
tot = {w: 0.0 for w in which}
for k, s, H, C, cnt in SHAPES:
    pad = k // 2
    Ho = (H + 2 * pad - k) // s + 1
    x = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
    g = torch.randn(B, Ho, Ho, C, device=dev).to(torch.bfloat16)
    g2 = torch.randn(B, Ho, Ho, C, device=dev).to(torch.bfloat16)
    y = torch.empty(B, Ho, Ho, C, device=dev, dtype=torch.bfloat16)
    dz = torch.empty(B, H, H, C, device=dev, dtype=torch.bfloat16)
    w = torch.randn(C, 1, k, k, device=dev)
    dw = torch.zeros(C, 1, k, k, device=dev)
    v = [torch.rand(C, device=dev) + 0.5 for _ in range(8)]
    st = torch.zeros(2, C, device=dev)
    sp = stream_ptr()
    calls = {
        "fwd": lambda: check(lib().cx_dwconv_fwd(ptr(x), ptr(w), ptr(v[0]), ptr(v[1]), ptr(y), ptr(st[0]), ptr(st[1]), B, H, H, C, k, s, pad, sp), "f"),
        "dgrad": lambda: check(lib().cx_dwconv_dgrad(ptr(g), ptr(g2), ptr(v[0]), ptr(v[1]), ptr(v[2]), ptr(w), ptr(x), ptr(v[3]), ptr(v[4]),
                                                     ptr(v[5]), ptr(v[6]), ptr(dz), ptr(st[0]), ptr(st[1]), B, H, H, C, k, s, pad, 0, sp), "d"),
        "wgrad": lambda: check(lib().cx_dwconv_wgrad(ptr(g), ptr(g2), ptr(v[0]), ptr(v[1]), ptr(v[2]), ptr(x), ptr(v[3]), ptr(v[4]), ptr(dw),
                                                     B, H, H, C, k, s, pad, sp), "w"),
    }
    nb = {"fwd": x.numel() * 2 + y.numel() * 2, "dgrad": 2 * g.numel() * 2 + 2 * x.numel() * 2, "wgrad": 2 * g.numel() * 2 + x.numel() * 2}
    line = "k%d s%d %3dx%-3d C%-4d x%d" % (k, s, H, H, C, cnt)
    for wname in which:
        f = calls[wname]
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        tot[wname] += us * cnt
        line += "  %s %7.1f us %5.2f TB/s" % (wname, us, nb[wname] / us / 1e6)
    print(line, flush=True)
print("per step (ms): " + "  ".join("%s %.2f" % (k_, v_ / 1e3) for k_, v_ in tot.items()))
