import sys, torch
sys.path.insert(0, '.')
from chexpert_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16; B = 256
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for hw, ct, cins in [(80, 256, [64, 224]), (40, 512, [256, 480]), (20, 1024, [512, 992])]:
    M = B * hw * hw
    buf = (torch.randn(B, hw, hw, ct, device=dev) * 0.5).to(bf)
    y1 = torch.randn(B, hw, hw, 128, device=dev).to(bf); dz2 = torch.randn(B, hw, hw, 128, device=dev).to(bf)
    ones = torch.ones(1024, device=dev); zeros = torch.zeros(1024, device=dev)
    for cin in cins:
        dw = torch.zeros(128, cin, 1, 1, device=dev)
        ctl = (cin + 63) // 64
        for target in (2048, 1024, 512, 256):
            sp = max(1, target // ctl)
            us = timeit(lambda: ops.conv_wgrad(dz2, buf[..., :cin], dw, g_prologue=ops.PRO_AFFINE2, g2=y1, ga=ones, gb=zeros, gc=zeros,
                                               x_prologue=ops.PRO_AFFINE_RELU, pa=ones, pb=zeros, splits=sp))
            print("wgrad hw=%2d cin=%4d wgs=%5d splits=%4d %8.1f us  %5.2f TB/s" % (hw, cin, sp * ctl, sp, us, M * (512 + cin * 2) / us / 1e6), flush=True)
