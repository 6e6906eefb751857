"""Micro-benchmark of the 1x1 kernel family at DenseNet121 bs=256 shapes: python scratch/bench_pw.py [fwd,dgrad,wgrad]"""
import sys, torch
sys.path.insert(0, '.')
from chexpert_amd import ops
dev = torch.device('cuda:0')
kinds = sys.argv[1].split(',') if len(sys.argv) > 1 else ['fwd', 'dgrad', 'wgrad']
B = 256
bf = torch.bfloat16
shapes = [(80, 256, [64, 128, 224]), (40, 512, [128, 256, 480]), (20, 1024, [256, 512, 992]), (10, 1024, [512, 992])]
R = 16

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for hw, ct, cins in shapes:
    M = B * hw * hw
    buf = (torch.randn(B, hw, hw, ct, device=dev) * 0.5).to(bf)
    gbuf = torch.zeros(B, hw, hw, ct, device=dev, dtype=bf)
    y1 = torch.randn(B, hw, hw, 128, device=dev).to(bf)
    dz2 = torch.randn(B, hw, hw, 128, device=dev).to(bf)
    ones = torch.ones(1024, device=dev); zeros = torch.zeros(1024, device=dev)
    st = torch.zeros(2, R * 1024, device=dev)
    for cin in cins:
        wf = torch.randn(128 * cin, device=dev).to(bf)
        dw = torch.zeros(128, cin, 1, 1, device=dev)
        if 'fwd' in kinds:
            us = timeit(lambda: ops.conv_gemm(buf[..., :cin], wf, y1, N=128, prologue=ops.PRO_AFFINE_RELU, pa=ones, pb=zeros,
                                              stat_sum=st[0], stat_sq=st[1], stat_replicas=R, stat_rstride=128))
            by = M * (cin * 2 + 256)
            print("fwd   hw=%2d cin=%4d %8.1f us  %5.2f TB/s" % (hw, cin, us, by / us / 1e6), flush=True)
        if 'dgrad' in kinds:
            us = timeit(lambda: ops.conv_gemm(dz2, wf, gbuf[..., :cin], N=cin, prologue=ops.PRO_AFFINE2, x2=y1, pa=ones, pb=zeros, pc=zeros,
                                              epilogue=ops.EPI_MASK, ex=buf[..., :cin], e_sc=ones, e_sh=zeros, e_mu=zeros, e_r=ones,
                                              e_scale=ones, stat_sum=st[0], stat_sq=st[1], accumulate=True, stat_replicas=R,
                                              stat_rstride=1024))
            by = M * (512 + cin * 6)
            print("dgrad hw=%2d cin=%4d %8.1f us  %5.2f TB/s" % (hw, cin, us, by / us / 1e6), flush=True)
        if 'fused' in kinds:
            us = timeit(lambda: ops.conv_gemm(dz2, wf, gbuf[..., :cin], N=cin, prologue=ops.PRO_AFFINE2, x2=y1, pa=ones, pb=zeros, pc=zeros,
                                              epilogue=ops.EPI_MASK, ex=buf[..., :cin], e_sc=ones, e_sh=zeros, e_mu=zeros, e_r=ones,
                                              e_scale=ones, stat_sum=st[0], stat_sq=st[1], accumulate=True, stat_replicas=R,
                                              stat_rstride=1024, fused_dw=dw))
            by = M * (512 + cin * 6)
            print("fused hw=%2d cin=%4d %8.1f us  %5.2f TB/s" % (hw, cin, us, by / us / 1e6), flush=True)
        if 'wgrad' in kinds:
            us = timeit(lambda: ops.conv_wgrad(dz2, buf[..., :cin], dw, g_prologue=ops.PRO_AFFINE2, g2=y1, ga=ones, gb=zeros, gc=zeros,
                                               x_prologue=ops.PRO_AFFINE_RELU, pa=ones, pb=zeros))
            by = M * (512 + cin * 2)
            print("wgrad hw=%2d cin=%4d %8.1f us  %5.2f TB/s" % (hw, cin, us, by / us / 1e6), flush=True)
    del buf, gbuf, y1, dz2
