"""conv_mm (partial N tiles) against the generic kernel on the EfficientNet-B4 @380 bs=64 1x1 shapes and the AAConv projections."""
import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chexpert_amd import ops, _lib
dev = torch.device('cuda:0')
raw = ctypes.CDLL(_lib.LIB_PATH)
bf = torch.bfloat16

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

B = 64
# (hw, cin, ce, cout): expand cin->ce, project ce->cout on hw x hw maps
STAGES = [(95, 32, 192, 32), (48, 56, 336, 56), (24, 112, 672, 112), (24, 160, 960, 160), (12, 272, 1632, 272), (12, 448, 2688, 448),
          (24, 112, 672, 160), (12, 160, 960, 272)]
for hw, cin, ce, cout in STAGES:
    M = B * hw * hw
    xin, xe, xe2, yp = t(B, hw, hw, cin), t(B, hw, hw, ce), t(B, hw, hw, ce), t(B, hw, hw, cout)
    we, wp = t(ce * cin), t(cout * ce)
    one, zero = torch.ones(max(ce, cout), device=dev), torch.zeros(max(ce, cout), device=dev)
    st = torch.zeros(2, max(ce, cout), device=dev)
    cases = {
        'expand  fwd': (lambda: ops.conv_gemm(xin, we, xe, N=ce, stat_sum=st[0], stat_sq=st[1]), cin, ce),
        'project fwd': (lambda: ops.conv_gemm(xe, wp, yp, N=cout, stat_sum=st[0], stat_sq=st[1]), ce, cout),
        'project dgr': (lambda: ops.conv_gemm(yp, wp, xe, N=ce, prologue=ops.PRO_AFFINE2, x2=yp, pa=one[:cout], pb=zero[:cout], pc=zero[:cout]), cout, ce),
        'expand  dgr': (lambda: ops.conv_gemm(xe, we, xin, N=cin, prologue=ops.PRO_AFFINE2, x2=xe2, pa=one[:ce], pb=zero[:ce], pc=zero[:ce]), ce, cin),
    }
    for name, (fn, K, N) in cases.items():
        res = []
        for on, form in [(0, 0), (1, 1), (1, 3)]:
            ops.KERNEL_HINT = ops.kernel_hint(on, form)
            try:
                us = timeit(fn)
                res.append("%6.1f us" % us)
            except Exception as e:
                res.append("   fail  ")
        by = 2.0 * M * (K + N) * (2 if 'dgr' in name else 1)
        print("%3dx%-3d %s K=%4d N=%4d | generic %s | 128x128 %s | 128x256 %s | floor %.0f us" % (hw, hw, name, K, N, *res, by / 4.5e6), flush=True)
    ops.KERNEL_HINT = ops.kernel_hint(-1, -1)
