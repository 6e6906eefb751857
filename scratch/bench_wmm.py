"""Micro-benchmark of the 1x1 weight-gradient kernel (csrc/wgrad_mm.hip) against conv_wgrad.hip's at the ResNet152 bs=128 shapes:
python scratch/bench_wmm.py [layers e.g. 2,3,4]"""
import ctypes, sys, torch
sys.path.insert(0, '.')
from chexpert_amd import ops, _lib
dev = torch.device('cuda:0')
raw = ctypes.CDLL(_lib.LIB_PATH)
layers = [int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else [1, 2, 3, 4]
B, bf = 128, torch.bfloat16
ops.WGRAD_SCRATCH_FLOATS = 16 << 20          # the slab workspace (as in the engines' backward)

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
    x4, x4b, y1, y1b = t(B, hw, hw, 4 * C), t(B, hw, hw, 4 * C), t(B, hw, hw, C), t(B, hw, hw, C)
    one, zero = torch.ones(4 * C, device=dev), torch.zeros(4 * C, device=dev)
    dw1, dw3 = torch.zeros(C, 4 * C, 1, 1, device=dev), torch.zeros(4 * C, C, 1, 1, device=dev)
    cases = {
        # conv1: dW[C][4C] from dz1 (two tensors) and the block input (plain)
        'w1': (lambda: ops.conv_wgrad(y1, x4, dw1, g_prologue=ops.PRO_AFFINE2, g2=y1b, ga=one[:C], gb=zero[:C], gc=zero[:C]), 4 * C, C),
        # conv3: dW[4C][C] from the block gradient (two tensors) and relu(bn2(y2))
        'w3': (lambda: ops.conv_wgrad(x4, y1, dw3, g_prologue=ops.PRO_AFFINE2, g2=x4b, ga=one, gb=zero, gc=zero, x_prologue=ops.PRO_AFFINE_RELU,
                                      pa=one[:C], pb=zero[:C]), C, 4 * C),
    }
    dw2 = torch.zeros(C, C, 3, 3, device=dev)
    cases['w2'] = (lambda: ops.conv_wgrad(y1, y1b, dw2, kh=3, kw=3, pad=1, g_prologue=ops.PRO_AFFINE2, g2=y1b, ga=one[:C], gb=zero[:C], gc=zero[:C],
                                          x_prologue=ops.PRO_AFFINE_RELU, pa=one[:C], pb=zero[:C]), 9 * C, C)
    for k, (fn, K, N) in cases.items():
        fl = 2.0 * M * K * N
        res = []
        for on, form in [(0, 0), (1, 0) if k == 'w2' else (1, 1), (1, 2), (1, 3), (-1, -1)]:
            ops.KERNEL_HINT = ops.kernel_hint(on, form)
            us = timeit(fn)
            res.append("%6.1f us %5.0f TF" % (us, fl / us / 1e6))
        print("L%d %s K=%4d N=%4d | old %s | 128x128 (w2: strip) %s | 256x128 %s | 128x256 %s | default %s" % (L, k, K, N, *res), flush=True)
    ops.KERNEL_HINT = ops.kernel_hint(-1, -1)
    del x4, x4b, y1, y1b
