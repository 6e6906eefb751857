"""Phase shares of the conv_mm main loop (diagnostic build with s_memtime stamps): python scratch/stamps_mm.py"""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, '.')
from chexpert_amd import ops, _lib
dev = torch.device('cuda:0')
raw = ctypes.CDLL(_lib.LIB_PATH)
bf = torch.bfloat16
B = 128
def t(*shape):
    return (torch.randn(*shape, device=dev) * 0.5).to(bf)
for L, kind in [(3, 'f2'), (3, 'f1'), (2, 'f2'), (4, 'f2')]:
    hw, C = {2: (40, 128), 3: (20, 256), 4: (10, 512)}[L]
    x4, y1, y1b = t(B, hw, hw, 4 * C), t(B, hw, hw, C), t(B, hw, hw, C)
    w1, w2 = t(4 * C * C), t(9 * C * C)
    one, zero = torch.ones(4 * C, device=dev), torch.zeros(4 * C, device=dev)
    for wm in (1, 3):
        ops.KERNEL_HINT = ops.kernel_hint(1, wm)
        raw.dbg_conv_mm_stamps(None, 1)
        for _ in range(2):
            if kind == 'f2':
                ops.conv_gemm(y1, w2, y1b, N=C, kh=3, kw=3, pad=1, prologue=ops.PRO_AFFINE_RELU, pa=one[:C], pb=zero[:C])
            else:
                ops.conv_gemm(x4, w1, y1, N=C)
        torch.cuda.synchronize()
        raw.dbg_conv_mm_stamps(None, 0)
        host = (ctypes.c_ulonglong * (2048 * 16))()
        raw.dbg_conv_mm_stamps(host, 2048 * 16)
        a = np.frombuffer(host, dtype=np.uint64).reshape(2048, 2, 8).astype(np.float64)
        for wv in (0, 1):
            v = a[:, wv]
            v = v[v[:, 4] > 0]
            per = v[:, :4] / v[:, 4:5]
            med = np.median(per, 0)
            print("L%d %s form=%d wave %s: steps %.0f, cycles/step %.0f: step %.0f barrier %.0f (unused %.0f %.0f)" %
                  (L, kind, wm, "0" if wv == 0 else "last", np.median(v[:, 4]), med.sum(), *med), flush=True)
    ops.KERNEL_HINT = ops.kernel_hint(-1, -1)
