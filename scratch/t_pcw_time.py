import os, sys, torch
sys.path.insert(0, '.')
from chexpert_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16
OLD = 0
NEW = ops.kernel_hint(-1, 8)
ops.set_det_wgrad(True)
def timeit(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
B = 256
out = []
for hw in (80, 40):
    y1 = (torch.randn(B, hw, hw, 128, device=dev) * 0.5).to(bf)
    gs = (torch.randn(B, hw, hw, 32, device=dev) * 0.5).to(bf)
    one, zero = torch.ones(128, device=dev), torch.zeros(128, device=dev)
    dw = torch.zeros(32, 128, 3, 3, device=dev)
    r = []
    for hint in (NEW, OLD):
        f = lambda: ops.conv_wgrad(gs, y1, dw, kh=3, kw=3, pad=1, x_prologue=ops.PRO_AFFINE_RELU, pa=one, pb=zero, hint=hint)
        r.append(min(timeit(f), timeit(f)))
    out.append("%dx%d pc %.1f old %.1f" % (hw, hw, r[0], r[1]))
print(" | ".join(out))
