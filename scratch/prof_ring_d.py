import sys, ctypes, torch
sys.path.insert(0, '.')
from chexpert_amd import ops, _lib
dev = torch.device('cuda:0'); bf = torch.bfloat16
B, hw, ct = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 80, 256
y1 = torch.randn(B, hw, hw, 128, device=dev).to(bf)
buf = (torch.randn(B, hw, hw, ct, device=dev) * .5).to(bf); gbuf = (torch.randn(B, hw, hw, ct, device=dev) * .5).to(bf)
dz2 = torch.zeros(B, hw, hw, 128, device=dev, dtype=bf)
ones = torch.ones(1024, device=dev); zeros = torch.zeros(1024, device=dev)
st = torch.zeros(2, 16 * 1024, device=dev)
wf = torch.randn(9 * 32 * 128, device=dev).to(bf)
c0 = ct - 32
ops.conv_gemm(gbuf[..., c0:], wf, dz2, N=128, kh=3, kw=3, pad=1, prologue=ops.PRO_AFFINE2, x2=buf[..., c0:], pa=ones, pb=zeros, pc=zeros,
              epilogue=ops.EPI_MASK, ex=y1, e_sc=ones, e_sh=zeros, e_mu=zeros, e_r=ones, e_scale=ones, stat_sum=st[0], stat_sq=st[1],
              stat_replicas=16, stat_rstride=1024)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
l = _lib.lib(); l.cx_debug_prof.argtypes = [ctypes.c_void_p]
print("rc", l.cx_debug_prof(out))
names = ['write_rows', 'barrier1', 'pre-item(y1 issue)', 'mfma', 'epilogue', 'tail', 'barrier2']
print({n: out[i] for i, n in enumerate(names)})
