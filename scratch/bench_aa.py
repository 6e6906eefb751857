import sys, torch
sys.path.insert(0, '.')
from chexpert_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16
B, nh, dk = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 8, 160
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for H, dv in [(40, 8), (20, 24), (10, 48)]:
    W = H; Cq = 2 * dk + dv
    qkv = (torch.randn(B, H, W, Cq, device=dev) * 0.5).to(bf)
    rh = torch.randn(20, 2 * H - 1, device=dev) * 0.2; rw = torch.randn(20, 2 * W - 1, device=dev) * 0.2
    o = torch.zeros(B, H * W, dv, device=dev); lse = torch.zeros(B * nh, H * W, device=dev)
    d_o = torch.randn(B, H * W, dv, device=dev)
    dqkv = torch.zeros(B, H * W, Cq, device=dev); drh = torch.zeros_like(rh); drw = torch.zeros_like(rw)
    tf = timeit(lambda: ops.aa_attention_fwd(qkv, rh, rw, o, lse, nh, dk, dv))
    tb = timeit(lambda: ops.aa_attention_bwd(qkv, rh, rw, o, d_o, lse, dqkv, drh, drw, nh, dk, dv))
    print("attention B=%d %dx%d dv=%d: fwd %.2f ms  bwd %.2f ms" % (B, H, W, dv, tf, tb), flush=True)
