# attention backward: error of each gradient against the fp32 oracle on the real 40x40 / 20x20 shapes (max-abs / max-abs, and rms / rms)
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from chexpert_amd import synth
from chexpert_amd import ops
from oracle import aaconv
dev = torch.device('cuda:0')
def bf(t): return t.to(torch.bfloat16).float()
for (B, H, W, dv) in [(1, 40, 40, 8), (2, 20, 20, 24)]:
    nh, dk = 8, 160; dkh, dvh = dk // nh, dv // nh; Cq = 2 * dk + dv
    qkv = bf(synth.uniform(1, (B, H, W, Cq), -1.5, 1.5))
    rel_h = synth.uniform(2, (dkh, 2 * H - 1), -1, 1) + dk ** -0.5
    rel_w = synth.uniform(3, (dkh, 2 * W - 1), -1, 1) + dk ** -0.5
    d_o = synth.uniform(4, (B, H * W, dv), -1, 1)
    t = qkv.permute(0, 3, 1, 2).clone().requires_grad_(True)
    rh, rw = rel_h.clone().requires_grad_(True), rel_w.clone().requires_grad_(True)
    q = t[:, :dk].reshape(B, nh, dkh, H, W) * dkh ** -0.5
    k = t[:, dk:2 * dk].reshape(B, nh, dkh, H, W)
    v = t[:, 2 * dk:].reshape(B, nh, dvh, H * W)
    P = torch.softmax(aaconv.attention_logits(q, k, rh, rw).reshape(B, nh, H * W, H * W), -1)
    o_ref = torch.einsum("bnqk,bndk->bqnd", P, v).reshape(B, H * W, dv)
    (o_ref * d_o).sum().backward()
    qd = qkv.to(torch.bfloat16).to(dev)
    o = torch.zeros(B, H * W, dv, device=dev); lse = torch.zeros(B * nh, H * W, device=dev)
    ops.aa_attention_fwd(qd, rel_h.to(dev), rel_w.to(dev), o, lse, nh, dk, dv)
    dqkv = torch.zeros(B, H * W, Cq, device=dev)
    drh, drw = torch.zeros_like(rel_h, device=dev), torch.zeros_like(rel_w, device=dev)
    ops.aa_attention_bwd(qd, rel_h.to(dev), rel_w.to(dev), o, d_o.to(dev), lse, dqkv, drh, drw, nh, dk, dv)
    want = t.grad.permute(0, 2, 3, 1).reshape(B, H * W, Cq)
    def e(got, w, what):
        got = got.cpu()
        print("  %dx%d %-12s max %.2e  rms %.2e   (bf16 rounding of the result itself: max %.2e rms %.2e)" % (H, W, what, (got - w).abs().max() / w.abs().max(), (got - w).norm() / w.norm(),
              (bf(w) - w).abs().max() / w.abs().max(), (bf(w) - w).norm() / w.norm()))
    e(dqkv[..., :dk], want[..., :dk], "dq"); e(dqkv[..., dk:2 * dk], want[..., dk:2 * dk], "dk"); e(dqkv[..., 2 * dk:], want[..., 2 * dk:], "dv")
    e(drh, rh.grad, "d key_rel_h"); e(drw, rw.grad, "d key_rel_w")
