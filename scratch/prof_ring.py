import sys, ctypes, torch
sys.path.insert(0, '.')
from chexpert_amd import ops, _lib
dev = torch.device('cuda:0'); bf = torch.bfloat16
B, hw, ct = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 80, 256
y1 = torch.randn(B, hw, hw, 128, device=dev).to(bf)
buf = torch.zeros(B, hw, hw, ct, device=dev, dtype=bf)
ones = torch.ones(1024, device=dev); zeros = torch.zeros(1024, device=dev)
st = torch.zeros(2, 16 * 1024, device=dev)
wf = torch.randn(9 * 32 * 128, device=dev).to(bf)
ops.conv_gemm(y1, wf, buf[..., ct - 32:], N=32, kh=3, kw=3, pad=1, prologue=ops.PRO_AFFINE_RELU, pa=ones, pb=zeros, stat_sum=st[0], stat_sq=st[1], stat_replicas=16, stat_rstride=1024)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
l = ctypes.CDLL(_lib.LIB_PATH) if hasattr(_lib, 'LIB_PATH') else _lib.lib()
l.cx_debug_prof.argtypes = [ctypes.c_void_p]
print("rc", l.cx_debug_prof(out))
names = ['write_rows', 'barrier1', 'issue', 'mfma', 'epilogue', 'barrier2']
for w in (0, 1):
    print("wave", 0 if w == 0 else 5, {n: out[i + 8 * w] for i, n in enumerate(names)})
