"""Per-map-size timing of the DenseNet 3x3 kernels (K = 128 -> N = 32) at bs=256: forward, input gradient, weight gradient."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chexpert_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256

def timeit(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for hw, ctot in ((80, 256), (40, 512), (20, 1024), (10, 1024)):
    M = B * hw * hw
    z1 = (torch.randn(B, hw, hw, 128, device=dev) * 0.5).to(bf)
    buf = (torch.randn(B, hw, hw, ctot, device=dev) * 0.5).to(bf)       # block buffer: the layer's 32 new channels are a slice
    gbuf = (torch.randn(B, hw, hw, ctot, device=dev) * 0.5).to(bf)
    dz = torch.empty(B, hw, hw, 128, device=dev, dtype=bf)
    w = (torch.randn(9 * 32 * 128, device=dev) * 0.05).to(bf)
    one, zero = torch.ones(128, device=dev), torch.zeros(128, device=dev)
    cap = 4096
    st = torch.zeros(2, cap * 128, device=dev)
    dw = torch.zeros(32, 128, 3, 3, device=dev)
    ys, gs = buf[..., 64:96], gbuf[..., 64:96]
    f = lambda: ops.conv_gemm(z1, w, ys, N=32, kh=3, kw=3, pad=1, prologue=ops.PRO_AFFINE_RELU, pa=one, pb=zero, stat_sum=st[0], stat_sq=st[1],
                              stat_det=True, stat_replicas=cap, stat_rstride=32)
    d = lambda: ops.conv_gemm(gs, w, dz, N=128, kh=3, kw=3, pad=1, prologue=ops.PRO_AFFINE2, x2=ys, pa=one[:32], pb=zero[:32], pc=zero[:32],
                              epilogue=ops.EPI_MASK, ex=z1, e_sc=one, e_sh=zero, e_mu=zero, e_r=one, e_scale=one, stat_sum=st[0], stat_sq=st[1],
                              stat_det=True, stat_replicas=cap, stat_rstride=128)
    g = lambda: ops.conv_wgrad(gs, z1, dw, kh=3, kw=3, stride=1, pad=1, g_prologue=ops.PRO_AFFINE2, g2=ys, ga=one[:32], gb=zero[:32], gc=zero[:32],
                               x_prologue=ops.PRO_AFFINE_RELU, pa=one, pb=zero)
    fl_f, fl_d = 2.0 * M * (128 + 32), 2.0 * M * (32 + 32 + 128 + 128)
    tf, td, tg = timeit(f), timeit(d), timeit(g)
    print("%2dx%-2d fwd %6.1f us (floor %5.1f)  dgrad %6.1f us (floor %5.1f)  wgrad %6.1f us (floor %5.1f)" % (
        hw, hw, tf, fl_f / 4.5e6, td, fl_d / 4.5e6, tg, 2.0 * M * (32 + 32 + 128) / 4.5e6), flush=True)
