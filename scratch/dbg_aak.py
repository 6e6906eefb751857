import sys, os, torch
sys.path.insert(0, '.')
from chexpert_amd import ops, synth
dev = torch.device('cuda:0')
B, H, W, dv = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
nh, dk = 8, 160
Cq = 2 * dk + dv
qkv = synth.uniform(1, (B, H, W, Cq), -1.5, 1.5).to(torch.bfloat16).to(dev)
rel_h = (synth.uniform(2, (20, 2 * H - 1), -1, 1) + dk ** -0.5).to(dev)
rel_w = (synth.uniform(3, (20, 2 * W - 1), -1, 1) + dk ** -0.5).to(dev)
d_o = synth.uniform(4, (B, H * W, dv), -1, 1).to(dev)
o = torch.zeros(B, H * W, dv, device=dev); lse = torch.zeros(B * nh, H * W, device=dev)
ops.aa_attention_fwd(qkv, rel_h, rel_w, o, lse, nh, dk, dv)
dqkv = torch.zeros(B, H * W, Cq, device=dev); drh = torch.zeros_like(rel_h); drw = torch.zeros_like(rel_w)
ops.aa_attention_bwd(qkv, rel_h, rel_w, o, d_o, lse, dqkv, drh, drw, nh, dk, dv)
torch.save(dqkv.cpu(), sys.argv[5])
if len(sys.argv) > 6:
    ref = torch.load(sys.argv[6])
    got = dqkv.cpu()
    for name, lo, hi in (("dk", dk, 2 * dk), ("dv", 2 * dk, Cq)):
        a, r = got[..., lo:hi], ref[..., lo:hi]
        e = (a - r).abs()
        print(name, "max err %.3e scale %.3e" % (e.max().item(), r.abs().max().item()))
        bad = (e > 1e-3 * r.abs().max()).nonzero()
        print("  bad entries", bad.shape[0], "of", e.numel())
        if bad.shape[0]:
            import collections
            print("  by batch", collections.Counter(bad[:, 0].tolist()).most_common(4))
            keys = collections.Counter(bad[:, 1].tolist())
            print("  bad keys (first 40 sorted)", sorted(keys)[:40], "... n=", len(keys))
            print("  by channel", sorted(collections.Counter(bad[:, 2].tolist()).items())[:48])
            for t in bad[:6].tolist():
                print("   ", t, "got %.5f want %.5f" % (a[tuple(t)].item(), r[tuple(t)].item()))
