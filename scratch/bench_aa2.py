import sys, torch
sys.path.insert(0, '.')
from chexpert_amd._lib import lib, ptr, check, stream_ptr
dev = torch.device('cuda:0')
B, H, W, nh, dk, dv = 64, 40, 40, 8, 160, 8
ldq = 2 * dk + dv
qkv = (torch.randn(B, H * W, ldq, device=dev) * 0.3).to(torch.bfloat16)
rh = torch.randn(20, 2 * H - 1, device=dev) * 0.1; rw = torch.randn(20, 2 * W - 1, device=dev) * 0.1
o = torch.zeros(B, H * W, dv, device=dev); lse = torch.zeros(B * nh, H * W, device=dev)
do = torch.randn(B, H * W, dv, device=dev); dqkv = torch.zeros(B, H * W, ldq, device=dev)
drh = torch.zeros_like(rh); drw = torch.zeros_like(rw)
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("fwd %.2f ms" % t(lambda: check(lib().cx_aa_attention_fwd(ptr(qkv), ptr(rh), ptr(rw), ptr(o), ptr(lse), B, H, W, nh, dk, dv, ldq, stream_ptr()), "f")))
print("bwd %.2f ms" % t(lambda: check(lib().cx_aa_attention_bwd(ptr(qkv), ptr(rh), ptr(rw), ptr(o), ptr(do), ptr(lse), ptr(dqkv), ptr(drh), ptr(drw), B, H, W, nh, dk, dv, ldq, stream_ptr()), "b")))
