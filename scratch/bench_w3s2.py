# stride-2 3x3 weight gradients: wgrad3_kernel<.., S2> (hint 1, 3) against conv_wgrad.hip's generic kernel (hint 0) at 128 images
import sys, torch
sys.path.insert(0, '.')
from chexpert_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16
ops.WGRAD_SCRATCH_FLOATS = 64 << 20
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1000
for name, B, H, K, N, gpro in [("aa T1", 128, 80, 256, 120, 0), ("aa T2", 128, 40, 512, 248, 0), ("aa T3", 128, 20, 1024, 504, 0),
                               ("resnet layer2.0", 128, 80, 128, 128, 2), ("resnet layer3.0", 128, 40, 256, 256, 2), ("resnet layer4.0", 128, 20, 512, 512, 2)]:
    Ho = H // 2
    g = torch.randn(B, Ho, Ho, N, device=dev).to(bf); g2 = torch.randn(B, Ho, Ho, N, device=dev).to(bf)
    x = torch.randn(B, H, H, K, device=dev).to(bf)
    ga, gb_, gc = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev) * 0.1, torch.randn(N, device=dev) * 0.1
    pa, pb = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3
    dw = torch.zeros(N, K, 3, 3, device=dev)
    res = []
    for on, form in ((1, 3), (0, 0)):
        ops.KERNEL_HINT = ops.kernel_hint(on, form)
        f = lambda: ops.conv_wgrad(g, x, dw, kh=3, kw=3, stride=2, pad=1, g_prologue=gpro, g2=g2 if gpro else None, ga=ga, gb=gb_, gc=gc,
                                   x_prologue=ops.PRO_AFFINE_RELU, pa=pa, pb=pb)
        t = timeit(f)
        res.append("%s %.1f us" % (ops.last_kernel()[:34], t))
    gf = 2.0 * B * Ho * Ho * N * K * 9 / 1e9
    print("%-16s B=%d %dx%d K=%d N=%d (%.0f GFLOP): %s" % (name, B, H, H, K, N, gf, " | ".join(res)), flush=True)
ops.KERNEL_HINT = 0
