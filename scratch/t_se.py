import sys, torch
sys.path.insert(0, '.')
from chexpert_amd._lib import lib, ptr, check, stream_ptr
lb = lib()
dev = torch.device('cuda:0'); bf = torch.bfloat16
torch.manual_seed(0)
for (B, HW, C, R) in [(8, 9025, 192, 8), (8, 576, 672, 28), (64, 144, 1632, 68), (8, 36100, 48, 12), (3, 100, 24, 6)]:
    x = (torch.randn(B, HW, C, device=dev) * 0.8).to(bf)
    du = (torch.randn(B, HW, C, device=dev) * 0.1).to(bf)
    sc = torch.rand(C, device=dev) + 0.5; sh = torch.randn(C, device=dev) * 0.2
    w1 = torch.randn(R, C, device=dev) * 0.05; b1 = torch.randn(R, device=dev) * 0.1
    w2 = torch.randn(C, R, device=dev) * 0.2; b2 = torch.randn(C, device=dev) * 0.1
    SLAB = 4 << 20
    slab = torch.zeros(SLAB, device=dev)
    res = []
    for fused in (0, 1):
        pooled = torch.zeros(B, C, device=dev); h1 = torch.zeros(B, R, device=dev); s = torch.zeros(B, C, device=dev)
        if fused:
            check(lb.cx_gap_se_fwd(ptr(x), ptr(sc), ptr(sh), ptr(pooled), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(h1), ptr(s), B, HW, C, R, 2, ptr(slab), SLAB, stream_ptr()), "f")
        else:
            check(lb.cx_gap_affine_act(ptr(x), ptr(sc), ptr(sh), ptr(pooled), B, HW, C, 2, ptr(slab), SLAB, stream_ptr()), "g")
            check(lb.cx_se_fwd(ptr(pooled), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(h1), ptr(s), B, C, R, stream_ptr()), "s")
        res.append((pooled.clone(), h1.clone(), s.clone()))
    ref_p = torch.nn.functional.silu(x.float() * sc + sh).mean(1)
    print("B%d HW%d C%d R%d fwd: pooled fused-vs-unfused %.2e (vs torch %.2e / %.2e), h1 %.2e, s %.2e" % (
        B, HW, C, R, (res[0][0] - res[1][0]).abs().max().item(), (res[0][0] - ref_p).abs().max().item(), (res[1][0] - ref_p).abs().max().item(),
        (res[0][1] - res[1][1]).abs().max().item(), (res[0][2] - res[1][2]).abs().max().item()))
    pooled, h1, s = res[0]
    WS = 8 << 20
    wsb = torch.zeros(WS, device=dev)
    outs = []
    for fused in (0, 1):
        ds = torch.zeros(B, C, device=dev); dpl = torch.zeros(B, C, device=dev)
        dw1 = torch.zeros(R, C, device=dev); db1 = torch.zeros(R, device=dev); dw2 = torch.zeros(C, R, device=dev); db2 = torch.zeros(C, device=dev)
        if fused:
            check(lb.cx_se_bwd_fused(ptr(du), ptr(x), ptr(sc), ptr(sh), ptr(ds), ptr(s), ptr(h1), ptr(pooled), ptr(w1), ptr(w2), ptr(dw1), ptr(db1), ptr(dw2), ptr(db2),
                                     ptr(dpl), B, HW, C, R, ptr(slab), SLAB, ptr(wsb), WS, stream_ptr()), "bf")
        else:
            check(lb.cx_se_bwd_reduce(ptr(du), ptr(x), ptr(sc), ptr(sh), ptr(ds), B, HW, C, ptr(slab), SLAB, stream_ptr()), "r")
            check(lb.cx_se_bwd(ptr(ds), ptr(s), ptr(h1), ptr(pooled), ptr(w1), ptr(w2), ptr(dw1), ptr(db1), ptr(dw2), ptr(db2), ptr(dpl), B, C, R, ptr(wsb), WS, stream_ptr()), "b")
        torch.cuda.synchronize()
        outs.append([t.clone() for t in (dpl, dw1, db1, dw2, db2)])
    print("   bwd fused-vs-unfused rel: " + ", ".join("%s %.2e" % (n, ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()) for n, a, b in zip(("dpooled", "dw1", "db1", "dw2", "db2"), outs[1], outs[0])))
