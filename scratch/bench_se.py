"""Per-shape times of the squeeze-excite FC kernels on EfficientNet-B4's (C, R) pairs at B = 64.  python scratch/bench_se.py"""
import sys, torch
sys.path.insert(0, '.')
from chexpert_amd import _lib
from chexpert_amd._lib import ptr, check
lb = _lib.lib()
dev = torch.device('cuda:0')
B = 64
shapes = [(48, 12), (144, 6), (192, 8), (336, 14), (672, 28), (960, 40), (1632, 68), (2688, 112)]
sp = torch.cuda.current_stream().cuda_stream
ws = torch.empty(256 << 20, dtype=torch.float32, device=dev)
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for C, R in shapes:
    g = lambda *s: torch.randn(*s, device=dev)
    pooled, w1, b1, w2, b2 = g(B, C), g(R, C) * 0.1, g(R), g(C, R) * 0.1, g(C)
    h1, s = torch.empty(B, R, device=dev), torch.empty(B, C, device=dev)
    ds, dpl = g(B, C), torch.empty(B, C, device=dev)
    dw1, db1, dw2, db2 = torch.zeros(R, C, device=dev), torch.zeros(R, device=dev), torch.zeros(C, R, device=dev), torch.zeros(C, device=dev)
    f = lambda: check(lb.cx_se_fwd(ptr(pooled), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(h1), ptr(s), B, C, R, sp), "fwd")
    b = lambda: check(lb.cx_se_bwd(ptr(ds), ptr(s), ptr(h1), ptr(pooled), ptr(w1), ptr(w2), ptr(dw1), ptr(db1), ptr(dw2), ptr(db2), ptr(dpl),
                                   B, C, R, ptr(ws), ws.numel(), sp), "bwd")
    print("C=%5d R=%4d  se_fwd %7.1f us   se_bwd (+4 slab sums) %7.1f us" % (C, R, timeit(f), timeit(b)), flush=True)
