"""conv3x3_pc_wgrad_kernel against the ring / strip weight-gradient kernels (kernel_hint form 7) and a torch fp32 reference; timings."""
import os, sys, torch, ctypes
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chexpert_amd import ops, _lib
dev = torch.device('cuda:0'); bf = torch.bfloat16
raw = ctypes.CDLL(_lib.LIB_PATH); raw.cx_last_kernel.restype = ctypes.c_char_p
OLD = 0
NEW = ops.kernel_hint(-1, 8)
ops.set_det_wgrad(True)

def run(B, H, W, hint, a2, seed=0):
    g_ = torch.Generator(device='cpu').manual_seed(seed)
    y1 = (torch.randn(B, H, W, 128, generator=g_) * 0.7).to(bf).to(dev)
    gbuf = (torch.randn(B, H, W, 96, generator=g_) * 0.5).to(bf).to(dev)
    xbuf = (torch.randn(B, H, W, 96, generator=g_) * 0.5).to(bf).to(dev)
    gs, xs = gbuf[..., 64:96], xbuf[..., 32:64]
    sc = (torch.rand(128, generator=g_) + 0.5).to(dev); sh = (torch.randn(128, generator=g_) * 0.3).to(dev)
    qa = (torch.rand(32, generator=g_) + 0.5).to(dev); qb = (torch.randn(32, generator=g_) * 0.2).to(dev); qc = (torch.randn(32, generator=g_) * 0.1).to(dev)
    dw = torch.zeros(32, 128, 3, 3, device=dev)
    kw = dict(g_prologue=ops.PRO_AFFINE2, g2=xs, ga=qa, gb=qb, gc=qc) if a2 else {}
    ops.conv_wgrad(gs, y1, dw, kh=3, kw=3, pad=1, x_prologue=ops.PRO_AFFINE_RELU, pa=sc, pb=sh, hint=hint, **kw)
    name = raw.cx_last_kernel().decode()
    # fp32 reference on the bf16-rounded operands
    a = F.relu(y1.float() * sc + sh).to(bf).float().permute(0, 3, 1, 2)
    gy = (gs.float() * qa + xs.float() * qb + qc).to(bf).float() if a2 else gs.float()
    gy = gy.permute(0, 3, 1, 2)
    ref = torch.nn.grad.conv2d_weight(a, (32, 128, 3, 3), gy, padding=1)
    return dw, ref, name

ok = True
for (B, H, W, a2) in [(2, 80, 80, 0), (3, 40, 40, 1), (5, 20, 20, 0), (7, 10, 10, 1), (2, 16, 24, 0), (1, 7, 9, 1), (9, 5, 4, 0), (130, 20, 20, 0), (17, 33, 66, 1),
                      (1, 1, 8, 0), (64, 64, 64, 0), (3, 70, 130, 0), (300, 10, 10, 0)]:
    d1, ref, n1 = run(B, H, W, NEW, a2)
    d2, _, n2 = run(B, H, W, OLD, a2)
    torch.cuda.synchronize()
    sc_ = ref.abs().max().item()
    e1, e2 = (d1 - ref).abs().max().item() / sc_, (d2 - ref).abs().max().item() / sc_
    good = e1 < 2e-3 and "pc_wgrad" in n1
    print("B%d %dx%d a2=%d: %s err %.2e | %s err %.2e  %s" % (B, H, W, a2, n1, e1, n2, e2, "" if good else "<-- BAD"), flush=True)
    ok &= good
print("ALL OK" if ok else "MISMATCH")
if not ok or (len(sys.argv) > 1 and sys.argv[1] == "notime"):
    sys.exit(0 if ok else 1)

def timeit(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for B in (256, 128):
    for hw in (80, 40, 20, 10):
        M = B * hw * hw
        y1 = (torch.randn(B, hw, hw, 128, device=dev) * 0.5).to(bf)
        gs = (torch.randn(B, hw, hw, 32, device=dev) * 0.5).to(bf)
        one, zero = torch.ones(128, device=dev), torch.zeros(128, device=dev)
        dw = torch.zeros(32, 128, 3, 3, device=dev)
        res = []
        for hint in (NEW, OLD, NEW, OLD):
            f = lambda: ops.conv_wgrad(gs, y1, dw, kh=3, kw=3, pad=1, x_prologue=ops.PRO_AFFINE_RELU, pa=one, pb=zero, hint=hint)
            res.append(timeit(f))
        print("B%d %2dx%-2d wgrad pc %6.1f / %6.1f us   old %6.1f / %6.1f us   (bytes at 5 TB/s %5.1f us)" % (
            B, hw, hw, res[0], res[2], res[1], res[3], 2.0 * M * (128 + 32) / 5e6), flush=True)
