# the ResNet bottleneck expansion shapes: activation-stationary kernel (form 5) against the tiled kernel (forms 1, 3)
import sys, torch
sys.path.insert(0, '.')
from chexpert_amd import ops, synth
dev = torch.device('cuda:0'); bf = torch.bfloat16
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1000
for (B, H, K, N, pro) in [(128, 20, 256, 1024, 1), (128, 20, 256, 1024, 2), (128, 40, 128, 512, 1), (128, 80, 64, 256, 1), (128, 20, 256, 1024, 0)]:
    x = (torch.randn(B, H, H, K, device=dev)).to(bf); x2 = (torch.randn(B, H, H, K, device=dev)).to(bf)
    w = ops.pack_weights((torch.randn(N, K, 1, 1, device=dev) * 0.05))
    pa, pb, pc = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3, torch.randn(K, device=dev) * 0.1
    y = torch.empty(B, H, H, N, dtype=bf, device=dev)
    cap = 8192
    s1, s2 = torch.zeros(cap, N, device=dev), torch.zeros(cap, N, device=dev)
    kw = {} if pro == 0 else dict(prologue=ops.PRO_AFFINE_RELU, pa=pa, pb=pb) if pro == 1 else dict(prologue=ops.PRO_AFFINE2, x2=x2, pa=pa, pb=pb, pc=pc)
    res = []
    for form in (5, 1, 3):
        ops.KERNEL_HINT = ops.kernel_hint(1, form)
        f = lambda: ops.conv_gemm(x, w, y, N=N, stat_sum=s1, stat_sq=s2, stat_det=True, stat_replicas=cap, stat_rstride=N, **kw)
        t = timeit(f)
        res.append("%s %.1f us" % (ops.last_kernel()[:28], t))
    mb = (B * H * H * (K * (2 if pro == 2 else 1) + N) * 2) / 1e6
    print("B=%d %dx%d K=%d N=%d pro=%d (%.0f MB, %.0f us at 5 TB/s): " % (B, H, H, K, N, pro, mb, mb / 5) + " | ".join(res), flush=True)
ops.KERNEL_HINT = 0
# the conv1 input gradient with the join epilogue (K = 256 -> N = 1024)
for (B, H, K, N) in [(128, 20, 256, 1024), (128, 40, 128, 512), (128, 80, 64, 256)]:
    g = torch.randn(B, H, H, K, device=dev).to(bf); g2 = torch.randn(B, H, H, K, device=dev).to(bf)
    w = ops.pack_weights((torch.randn(N, K, 1, 1, device=dev) * 0.05))
    pa, pb, pc = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3, torch.randn(K, device=dev) * 0.1
    y = torch.randn(B, H, H, N, device=dev).to(bf); y3 = torch.randn(B, H, H, N, device=dev).to(bf)
    mask = torch.randint(0, 256, (B * H * H * N // 8,), dtype=torch.uint8, device=dev)
    mu, r = torch.randn(N, device=dev) * 0.1, torch.rand(N, device=dev) + 0.5
    cap = 8192
    s1, s2 = torch.zeros(cap, N, device=dev), torch.zeros(cap, N, device=dev)
    res = []
    for form in (5, 1, 3):
        ops.KERNEL_HINT = ops.kernel_hint(1, form)
        f = lambda: ops.conv_gemm(g, w, y, N=N, prologue=ops.PRO_AFFINE2, x2=g2, pa=pa, pb=pb, pc=pc, accumulate=True, epilogue=ops.EPI_JOIN, ex=y3, e_mu=mu, e_r=r,
                                  emask=mask, stat_sum=s1, stat_sq=s2, stat_det=True, stat_replicas=cap, stat_rstride=N)
        t = timeit(f)
        res.append("%s %.1f us" % (ops.last_kernel()[:28], t))
    mb = (B * H * H * (2 * K + 3 * N) * 2 + B * H * H * N / 8) / 1e6
    print("join B=%d %dx%d K=%d N=%d (%.0f MB, %.0f us at 5 TB/s): " % (B, H, H, K, N, mb, mb / 5) + " | ".join(res), flush=True)
ops.KERNEL_HINT = 0
