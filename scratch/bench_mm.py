"""Micro-benchmark of the wide-channel implicit GEMM (csrc/conv_mm.hip) against the generic kernel at the ResNet152 bs=128 shapes:
python scratch/bench_mm.py [layers e.g. 2,3,4] [kinds e.g. f1,f2,f3,d3,d2,d1]"""
import ctypes, sys, torch
sys.path.insert(0, '.')
from chexpert_amd import ops, _lib
dev = torch.device('cuda:0')
raw = ctypes.CDLL(_lib.LIB_PATH)
layers = [int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else [1, 2, 3, 4]
kinds = sys.argv[2].split(',') if len(sys.argv) > 2 else ['f1', 'f2', 'f3', 'd3', 'd2', 'd1']
B, bf = int(sys.argv[3]) if len(sys.argv) > 3 else 128, torch.bfloat16

def timeit(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

def t(*shape):
    return (torch.randn(*shape, device=dev) * 0.5).to(bf)

for L in layers:
    hw, C = {1: (80, 64), 2: (40, 128), 3: (20, 256), 4: (10, 512)}[L]
    M = B * hw * hw
    x4, x4b, y1, y1b, y4 = t(B, hw, hw, 4 * C), t(B, hw, hw, 4 * C), t(B, hw, hw, C), t(B, hw, hw, C), t(B, hw, hw, 4 * C)
    w1, w2 = t(4 * C * C), t(9 * C * C)
    one, zero = torch.ones(4 * C, device=dev), torch.zeros(4 * C, device=dev)
    cap = (M + 127) // 128
    st = torch.zeros(2, cap * 4 * C, device=dev)
    def stats(N):
        return dict(stat_sum=st[0], stat_sq=st[1], stat_det=True, stat_replicas=cap, stat_rstride=N)
    def mask(N, ex):
        return dict(epilogue=ops.EPI_MASK, ex=ex, e_sc=one[:N], e_sh=zero[:N], e_mu=zero[:N], e_r=one[:N], e_scale=one[:N])
    cases = {
        'f1': (lambda: ops.conv_gemm(x4, w1, y1, N=C, **stats(C)), 4 * C, C, 1),
        'f2': (lambda: ops.conv_gemm(y1, w2, y1b, N=C, kh=3, kw=3, pad=1, prologue=ops.PRO_AFFINE_RELU, pa=one[:C], pb=zero[:C], **stats(C)), C, C, 9),
        'f3': (lambda: ops.conv_gemm(y1, w1, y4, N=4 * C, prologue=ops.PRO_AFFINE_RELU, pa=one[:C], pb=zero[:C], **stats(4 * C)), C, 4 * C, 1),
        'd3': (lambda: ops.conv_gemm(x4, w1, y1b, N=C, prologue=ops.PRO_AFFINE2, x2=x4b, pa=one, pb=zero, pc=zero, **mask(C, y1), **stats(C)), 4 * C, C, 1),
        'd2': (lambda: ops.conv_gemm(y1, w2, y1b, N=C, kh=3, kw=3, pad=1, prologue=ops.PRO_AFFINE2, x2=y1, pa=one[:C], pb=zero[:C], pc=zero[:C], **mask(C, y1), **stats(C)), C, C, 9),
        'd1': (lambda: ops.conv_gemm(y1, w1, y4, N=4 * C, prologue=ops.PRO_AFFINE2, x2=y1b, pa=one[:C], pb=zero[:C], pc=zero[:C]), C, 4 * C, 1),
    }
    for k in kinds:
        fn, K, N, taps = cases[k]
        fl = 2.0 * M * K * N * taps
        res = []
        for on, form in [(0, 0), (1, 1), (1, 3), (-1, -1)]:
            ops.KERNEL_HINT = ops.kernel_hint(on, form)
            us = timeit(fn)
            res.append("%6.1f us %5.0f TF" % (us, fl / us / 1e6))
        print("L%d %s K=%4d N=%4d taps=%d | generic %s | 128x128 %s | 128x256 %s | default %s" % (L, k, K, N, taps, *res), flush=True)
    ops.KERNEL_HINT = ops.kernel_hint(-1, -1)
    del x4, x4b, y1, y1b, y4
