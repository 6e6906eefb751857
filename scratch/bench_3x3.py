"""Micro-benchmark of the dense-layer 3x3 kernels at DenseNet121 bs=256 shapes: python scratch/bench_3x3.py [fwd,dgrad,wgrad]"""
import sys, torch
sys.path.insert(0, '.')
from chexpert_amd import ops
dev = torch.device('cuda:0')
kinds = sys.argv[1].split(',') if len(sys.argv) > 1 else ['fwd', 'dgrad', 'wgrad']
B, bf, R = 256, torch.bfloat16, 16

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for hw, ct in [(80, 256), (40, 512), (20, 1024), (10, 1024)]:
    M = B * hw * hw
    y1 = torch.randn(B, hw, hw, 128, device=dev).to(bf)
    buf = (torch.randn(B, hw, hw, ct, device=dev) * 0.5).to(bf)
    gbuf = (torch.randn(B, hw, hw, ct, device=dev) * 0.5).to(bf)
    dz2 = torch.zeros(B, hw, hw, 128, device=dev, dtype=bf)
    ones = torch.ones(1024, device=dev); zeros = torch.zeros(1024, device=dev)
    st = torch.zeros(2, R * 1024, device=dev)
    wf = torch.randn(9 * 32 * 128, device=dev).to(bf)
    dw = torch.zeros(32, 128, 3, 3, device=dev)
    c0 = ct - 32
    if 'fwd' in kinds:
        us = timeit(lambda: ops.conv_gemm(y1, wf, buf[..., c0:], N=32, kh=3, kw=3, pad=1, prologue=ops.PRO_AFFINE_RELU, pa=ones, pb=zeros,
                                          stat_sum=st[0], stat_sq=st[1], stat_replicas=R, stat_rstride=1024))
        print("3x3 fwd   hw=%2d %8.1f us  %5.2f TB/s  %6.1f TFLOP/s" % (hw, us, M * 320 / us / 1e6, M * 73728 / us / 1e6), flush=True)
    if 'dgrad' in kinds:
        us = timeit(lambda: ops.conv_gemm(gbuf[..., c0:], wf, dz2, N=128, kh=3, kw=3, pad=1, prologue=ops.PRO_AFFINE2, x2=buf[..., c0:],
                                          pa=ones, pb=zeros, pc=zeros, epilogue=ops.EPI_MASK, ex=y1, e_sc=ones, e_sh=zeros, e_mu=zeros,
                                          e_r=ones, e_scale=ones, stat_sum=st[0], stat_sq=st[1], stat_replicas=R, stat_rstride=1024))
        print("3x3 dgrad hw=%2d %8.1f us  %5.2f TB/s  %6.1f TFLOP/s" % (hw, us, M * (128 + 512) / us / 1e6, M * 73728 / us / 1e6), flush=True)
    if 'wgrad' in kinds:
        us = timeit(lambda: ops.conv_wgrad(gbuf[..., c0:], y1, dw, kh=3, kw=3, pad=1, g_prologue=ops.PRO_AFFINE2, g2=buf[..., c0:], ga=ones,
                                           gb=zeros, gc=zeros, x_prologue=ops.PRO_AFFINE_RELU, pa=ones, pb=zeros))
        print("3x3 wgrad hw=%2d %8.1f us  %5.2f TB/s  %6.1f TFLOP/s" % (hw, us, M * (128 + 256) / us / 1e6, M * 73728 / us / 1e6), flush=True)
    del y1, buf, gbuf, dz2
