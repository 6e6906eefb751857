"""GPU: attention-augmented convolution kernels against the oracle's closed form."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from chexpert_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from chexpert_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def bf(t):
    return t.to(torch.bfloat16).float()


def close(got, want, rel, what=""):
    scale = want.abs().max().item() + 1e-9
    err = (got - want).abs().max().item()
    assert err <= rel * scale, "%s: max err %.3e vs scale %.3e (rel %.2e)" % (what, err, scale, err / scale)


@pytest.mark.parametrize("B,H,W,dv", [(2, 5, 7, 8), (1, 10, 10, 48), (2, 20, 20, 24), (1, 40, 40, 8), (2, 12, 20, 16), (1, 9, 40, 8),
                                      (2, 8, 8, 64),         # dv/nh = 8: the third stage of WRN-28-10 at 8 heads (attn_aug_conv.py:602)
                                      (2, 16, 16, 72), (2, 8, 8, 104)])     # dv/nh = 9 / 13: Densenet-BC transitions of the CIFAR harness at v = 0.7
def test_attention_forward_backward(dev, B, H, W, dv):
    from chexpert_amd import ops
    from oracle import aaconv
    nh, dk = 8, 160
    dkh, dvh = dk // nh, dv // nh
    Cq = 2 * dk + dv
    qkv = bf(synth.uniform(1, (B, H, W, Cq), -1.5, 1.5))
    rel_h = synth.uniform(2, (dkh, 2 * H - 1), -1, 1) + dk ** -0.5
    rel_w = synth.uniform(3, (dkh, 2 * W - 1), -1, 1) + dk ** -0.5
    d_o = synth.uniform(4, (B, H * W, dv), -1, 1)
    # oracle on the same (bf16-rounded) qkv
    t = qkv.permute(0, 3, 1, 2).clone().requires_grad_(True)               # (B,Cq,H,W)
    rh, rw = rel_h.clone().requires_grad_(True), rel_w.clone().requires_grad_(True)
    q = t[:, :dk].reshape(B, nh, dkh, H, W) * dkh ** -0.5
    k = t[:, dk:2 * dk].reshape(B, nh, dkh, H, W)
    v = t[:, 2 * dk:].reshape(B, nh, dvh, H * W)
    P = torch.softmax(aaconv.attention_logits(q, k, rh, rw).reshape(B, nh, H * W, H * W), -1)
    o_ref = torch.einsum("bnqk,bndk->bqnd", P, v).reshape(B, H * W, dv)      # channels head-major n*dvh+d
    (o_ref * d_o).sum().backward()
    qd = qkv.to(torch.bfloat16).to(dev)
    o = torch.zeros(B, H * W, dv, device=dev)
    lse = torch.zeros(B * nh, H * W, device=dev)
    ops.aa_attention_fwd(qd, rel_h.to(dev), rel_w.to(dev), o, lse, nh, dk, dv)
    close(o.cpu(), o_ref.detach(), 2e-4, "o")
    if H * W <= 400:                  # AAConv2d.weights (attn_aug_conv.py:87): the softmax rebuilt from the saved log-sum-exp
        wts = ops.aa_attention_weights(qd, rel_h.to(dev), rel_w.to(dev), lse, nh, dk, dv)
        assert wts.shape == (B, nh, H * W, H * W)
        close(wts.cpu(), P.detach(), 2e-4, "weights")
        assert (wts.sum(-1) - 1).abs().max().item() < 1e-4
    dqkv = torch.full((B, H * W, Cq), 7.0, device=dev)
    drh, drw = torch.zeros_like(rel_h, device=dev), torch.zeros_like(rel_w, device=dev)
    ops.aa_attention_bwd(qd, rel_h.to(dev), rel_w.to(dev), o, d_o.to(dev), lse, dqkv, drh, drw, nh, dk, dv)
    want = t.grad.permute(0, 2, 3, 1).reshape(B, H * W, Cq)
    close(dqkv.cpu()[..., :dk], want[..., :dk], 1e-3, "dq")
    close(dqkv.cpu()[..., dk:2 * dk], want[..., dk:2 * dk], 1e-3, "dk")
    close(dqkv.cpu()[..., 2 * dk:], want[..., 2 * dk:], 1e-3, "dv")
    close(drh.cpu(), rh.grad, 1e-3, "d key_rel_h")
    close(drw.cpu(), rw.grad, 1e-3, "d key_rel_w")


def test_instance_norm_relu_and_backward(dev):
    from chexpert_amd import ops
    B, H, W, C = 3, 6, 10, 64
    x = bf(synth.uniform(5, (B, H, W, C + 32), -2, 3))
    xs = x[..., :C].permute(0, 3, 1, 2).clone().requires_grad_(True)
    a_ref = F.relu(F.instance_norm(xs, eps=1e-5))
    da = bf(synth.uniform(6, (B, H, W, C), -1, 1))
    a_ref.backward(da.permute(0, 3, 1, 2))
    xd = x.to(torch.bfloat16).to(dev)
    s, q = torch.zeros(B * C, device=dev), torch.zeros(B * C, device=dev)
    ops.stats_bc(xd[..., :C], s, q)
    sc, sh = torch.zeros(B * C, device=dev), torch.zeros(B * C, device=dev)
    ops.bn_coef(s, q, H * W, None, None, 1e-5, 0.0, None, None, sc, sh, None, None, B * C)
    a = torch.zeros(B, H, W, C, dtype=torch.bfloat16, device=dev)
    ops.affine_relu_bc(xd[..., :C], sc, sh, a)
    close(a.float().cpu().permute(0, 3, 1, 2), a_ref.detach(), 6e-3, "IN+ReLU")
    g = torch.full((B, H, W, C + 32), 3.0, dtype=torch.bfloat16, device=dev)
    S1, S2 = torch.zeros(B * C, device=dev), torch.zeros(B * C, device=dev)
    ops.in_relu_bwd(da.to(torch.bfloat16).to(dev), xd[..., :C], sc, sh, S1, S2, g[..., :C])
    close(g[..., :C].float().cpu().permute(0, 3, 1, 2), xs.grad, 8e-3, "IN backward")
    assert (g[..., C:].float() == 3.0).all()


def test_conv_partial_channel_tiles(dev):
    """N and K that are multiples of 8 but not of 32 (AAConv branch: 120 / 232 / 464 output channels, 328 qkv)."""
    from chexpert_amd import ops
    B, H, W, K, N = 2, 8, 10, 64, 120
    x = bf(synth.uniform(7, (B, H, W, K), -1, 1))
    w = bf(synth.uniform(8, (N, K, 3, 3), -0.1, 0.1))
    want = F.conv2d(x.permute(0, 3, 1, 2), w, stride=2, padding=1)
    y = torch.zeros(B, H // 2, W // 2, N + 8, dtype=torch.bfloat16, device=dev)
    ops.conv_gemm(x.to(torch.bfloat16).to(dev), ops.pack_weights(w.to(dev)), y[..., :N], N=N, kh=3, kw=3, stride=2, pad=1)
    close(y[..., :N].float().cpu().permute(0, 3, 1, 2), want, 6e-3, "fwd N=120")
    # input gradient of the same strided conv: K = 120 (partial last K step), transposed stride 2
    dy = bf(synth.uniform(9, (B, H // 2, W // 2, N), -1, 1))
    want = torch.nn.grad.conv2d_input((B, K, H, W), w, dy.permute(0, 3, 1, 2), stride=2, padding=1)
    dx = torch.zeros(B, H, W, K, dtype=torch.bfloat16, device=dev)
    ops.conv_gemm(dy.to(torch.bfloat16).to(dev), ops.pack_weights(w.to(dev), transpose=True), dx, N=K, kh=3, kw=3, pad=1, tstride=2)
    close(dx.float().cpu().permute(0, 3, 1, 2), want, 6e-3, "dgrad K=120 tstride=2")
    want = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2), (N, K, 3, 3), dy.permute(0, 3, 1, 2), stride=2, padding=1)
    dw = torch.zeros(N, K, 3, 3, device=dev)
    ops.conv_wgrad(dy.to(torch.bfloat16).to(dev), x.to(torch.bfloat16).to(dev), dw, kh=3, kw=3, stride=2, pad=1)
    close(dw.cpu(), want, 2e-3, "wgrad N=120")


# ---------------------------------------------------------------------------------------------- whole AA-DenseNet
def _rel(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-30)), float(a.norm() / (b.norm() + 1e-30))


def _build_aa(cfg, S, n_cls, seed, dev, smooth):
    from chexpert_amd.models import DenseNet
    from oracle import nets
    spec = nets.densenet_spec(n_cls, block_config=cfg, attn=dict(k=.2, v=.1, nh=8), input_hw=(S, S))
    sd = synth.fill_state_dict_(nets.zeros_state_dict(spec), seed)
    if smooth:
        for k in sd:
            if k.endswith(".bias") and "classifier" not in k:
                sd[k] = torch.full_like(sd[k], 2.5)
            if k.endswith(".weight") and sd[k].dim() == 1:
                sd[k] = synth.uniform(7, sd[k].shape, 0.8, 1.2)
    model = DenseNet(32, cfg, 64, num_classes=n_cls, attn_params={"k": 0.2, "v": 0.1, "nh": 8, "relative": True, "input_dims": (S, S)})
    assert list(model.state_dict().keys()) == list(spec.keys())
    model.load_state_dict(sd, strict=True)
    return model.to(dev), sd


@pytest.mark.parametrize("cfg,B,S", [((6, 4, 2, 2), 4, 64), ((6, 4, 2, 2), 2, 128)])
def test_aa_densenet_matches_oracle(dev, cfg, B, S):
    from oracle import nets, step
    n_cls = 5
    model, sd = _build_aa(cfg, S, n_cls, 21, dev, smooth=True)
    x, t = synth.xray_batch(1234, B, S), synth.targets(99, B, n_cls)
    fwd = lambda s, xx, train=True: nets.densenet_forward(s, xx, cfg, train=train, nh=8)
    sd_o = {k: v.clone() for k, v in sd.items()}
    loss_o, logits_o, grads_o = step.train_step(fwd, sd_o, x, t)
    with torch.no_grad():
        le_o = fwd({k: v.clone() for k, v in sd.items()}, x, train=False)
        model.eval()
        le = model(x.to(dev)).cpu()
    print("aa-densenet%s S=%d eval logits rel %.3e" % (cfg, S, _rel(le, le_o)))
    assert _rel(le, le_o) < 1e-2
    model.train()
    out = model(x.to(dev))
    loss = torch.nn.BCEWithLogitsLoss(reduction="none")(out, t.to(dev)).sum(1).mean(0)
    model.zero_grad()
    loss.backward()
    print("aa-densenet%s S=%d train logits rel %.3e" % (cfg, S, _rel(out.detach().cpu(), logits_o)))
    assert _rel(out.detach().cpu(), logits_o) < 1e-2
    gmax = max(g.norm().item() for g in grads_o.values())
    worst = []
    for k, p in model.named_parameters():
        if grads_o[k].norm().item() < 1e-4 * gmax:
            continue
        c, n = _cos(p.grad.cpu(), grads_o[k])
        worst.append((c, n, k))
    worst.sort()
    print("aa-densenet worst (cos, norm ratio): %s" % worst[:4])
    print("aa params:", [w for w in worst if "transition" in w[2]])
    # bf16 storage + batch statistics at B<=4: norm-parameter gradients are cancellation-heavy sums (see test_model_gpu.py)
    # ... and fp32 atomics make the batch statistics vary run to run: observed spread on norm1.bias 0.88-0.96
    # transition1.conv.out_proj.weight (a dv x dv matrix summed over 512 pixels at B=2): norm ratio 0.90-1.0 from run to run
    # (norm1.bias of block 3 at B=2: 0.82-0.96 from run to run, 1 run in 16 below 0.84)
    lim = lambda k: (0.75, 0.20) if ".norm" in k else ((0.95, 0.13) if "out_proj" in k else (0.95, 0.08))
    bad = [w for w in worst if w[0] < lim(w[2])[0] or abs(w[1] - 1) > lim(w[2])[1]]
    assert not bad, "gradient mismatch (cos, norm-ratio, name): %s" % bad[:8]


def test_aa_densenet_without_relative_position_logits_matches_oracle(dev):
    """attn_params["relative"] = False (attn_aug_conv.py:38, :76-86 skipped: no key_rel_h / key_rel_w parameters, plain q.k logits):
    the attention kernels run with zero position tables; state_dict keys as the reference's, logits and gradients against the oracle."""
    from chexpert_amd.models import DenseNet
    from oracle import nets, step
    cfg, B, S, n_cls = (6, 4, 2, 2), 4, 64, 5
    spec = nets.densenet_spec(n_cls, block_config=cfg, attn=dict(k=.2, v=.1, nh=8), input_hw=(S, S))
    sd = synth.fill_state_dict_(nets.zeros_state_dict(spec), 23)
    sd = type(sd)((k, v) for k, v in sd.items() if "key_rel" not in k)
    for k in sd:
        if k.endswith(".bias") and "classifier" not in k:
            sd[k] = torch.full_like(sd[k], 2.5)
        if k.endswith(".weight") and sd[k].dim() == 1:
            sd[k] = synth.uniform(7, sd[k].shape, 0.8, 1.2)
    model = DenseNet(32, cfg, 64, num_classes=n_cls, attn_params={"k": 0.2, "v": 0.1, "nh": 8, "relative": False, "input_dims": (S, S)})
    assert list(model.state_dict().keys()) == list(sd.keys())
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).train()
    x, t = synth.xray_batch(1235, B, S), synth.targets(98, B, n_cls)
    fwd = lambda s_, xx, train=True: nets.densenet_forward(s_, xx, cfg, train=train, nh=8)
    loss_o, logits_o, grads_o = step.train_step(fwd, {k: v.clone() for k, v in sd.items()}, x, t)
    model.zero_grad()
    loss, logits = model.forward_backward(x.to(dev), t.to(dev))
    assert _rel(logits.cpu(), logits_o) < 1e-2, _rel(logits.cpu(), logits_o)
    for k, p in model.named_parameters():
        if "transition" in k and p.dim() > 1:
            c, n = _cos(p.grad.cpu(), grads_o[k])
            assert c > 0.95 and abs(n - 1) < 0.13, (k, c, n)
    w = None
    model.eval()
    with torch.no_grad():
        model(x.to(dev))
        w = model.features.transition1.conv.weights
    assert (w.sum(-1) - 1).abs().max().item() < 1e-4


@pytest.mark.parametrize("vr,dtype", [(0.32, "fp32"), (0.7, "fp32"), (0.32, "bf16"), (0.7, "bf16")])
def test_aa_densenet_value_head_sizes_outside_the_reference_set(dev, vr, dtype):
    """attn_params["v"] = 0.32 / 0.7 on 128-wide transitions: 5 / 11 value channels per head (attn_aug_conv.py:419: dv =
    int((v * cout // nh) * nh); the reference's own configurations give 1, 2, 3, 4, 6, 8, 9, 13) -- the generic attention kernels."""
    from chexpert_amd.models import DenseNet
    from oracle import nets, step
    cfg, B, S, n_cls = (6, 4, 2, 2), 4, 64, 5
    spec = nets.densenet_spec(n_cls, block_config=cfg, attn=dict(k=.2, v=vr, nh=8), input_hw=(S, S))
    sd = synth.fill_state_dict_(nets.zeros_state_dict(spec), 25)
    for k in sd:
        if k.endswith(".bias") and "classifier" not in k:
            sd[k] = torch.full_like(sd[k], 2.5)
        if k.endswith(".weight") and sd[k].dim() == 1:
            sd[k] = synth.uniform(7, sd[k].shape, 0.8, 1.2)
    model = DenseNet(32, cfg, 64, num_classes=n_cls, attn_params={"k": 0.2, "v": vr, "nh": 8, "relative": True, "input_dims": (S, S)})
    assert model.features.transition1.conv.dv // 8 == (5 if vr == 0.32 else 11)
    assert list(model.state_dict().keys()) == list(spec.keys())
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).storage_dtype(dtype).train()
    x, t = synth.xray_batch(1236, B, S), synth.targets(97, B, n_cls)
    fwd = lambda s_, xx, train=True: nets.densenet_forward(s_, xx, cfg, train=train, nh=8)
    loss_o, logits_o, grads_o = step.train_step(fwd, {k: v.clone() for k, v in sd.items()}, x, t)
    model.zero_grad()
    loss, logits = model.forward_backward(x.to(dev), t.to(dev))
    # fp32 storage mode: the sharp check of the new kernel instances (1e-4); bf16 at B = 4 on a hash-filled fixture is a gross-error
    # check only: storage rounding through train-mode statistics of 4 images measures 1.0e-2 at v = 0.32 and 2.1e-2 at v = 0.7 (where
    # 69 % of a transition's channels come out of the attention), as for the head sizes the reference uses
    tol = 1e-4 if dtype == "fp32" else 4e-2
    assert _rel(logits.cpu(), logits_o) < tol, _rel(logits.cpu(), logits_o)
    for k, p in model.named_parameters():
        if "transition" in k and p.dim() > 1:
            c, n = _cos(p.grad.cpu(), grads_o[k])
            assert (c > 0.9999 and abs(n - 1) < 1e-3) if dtype == "fp32" else (c > 0.95 and abs(n - 1) < 0.13), (k, c, n)


def test_aaconv2d_weights_property_after_forward(dev):
    """`model.features.transitionN.conv.weights` (chexpert.py:365, :383) after an eval forward: (B, nh, HW, HW), rows sum to 1,
    equal to the oracle's softmax of the same layer."""
    from oracle import nets
    cfg, B, S, n_cls = (6, 4, 2, 2), 2, 64, 5
    model, sd = _build_aa(cfg, S, n_cls, 21, dev, smooth=True)
    assert model.features.transition1.conv.weights is None            # no forward yet
    x = synth.xray_batch(77, B, S)
    model.eval()
    with torch.no_grad():
        model(x.to(dev))
    aa = model.features.transition1.conv
    w = aa.weights
    hw = (S // 8) ** 2
    assert w.shape == (B, aa.nh, hw, hw) and w.dtype == torch.float32
    assert (w.sum(-1) - 1).abs().max().item() < 1e-4
    taps = {}
    with torch.no_grad():
        nets.densenet_forward({k: v.clone() for k, v in sd.items()}, x, cfg, train=False, nh=8, taps=taps)
        a = F.relu(F.instance_norm(taps["block1"], eps=1e-5))                       # attn_aug_conv.py:438-440
        _, w_ref = nets._aa(sd, "features.transition1.conv", a, 2, 8, return_weights=True)
    err = (w.cpu() - w_ref).abs().max().item()
    print("attention weights max abs err %.3e (max prob %.3f)" % (err, w_ref.max().item()))
    assert err < 2e-2


def test_aadensenet121_reference_golden_eval(dev):
    """aadensenet121 @320 (chexpert.py:475-476): eval logits recorded from the REAL reference."""
    import json
    rec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "nets.json")))["aadensenet121_320_b1"]
    model, sd = _build_aa((6, 12, 24, 16), 320, rec["n_classes"], rec["sd_seed"], dev, smooth=False)
    assert sum(p.numel() for p in model.parameters()) == rec["n_params"] == 12534381
    x = synth.xray_batch(rec["x_seed"], rec["B"], rec["S"]).to(dev)
    model.eval()
    with torch.no_grad():
        le = model(x).cpu()
    e = _rel(le, torch.tensor(rec["logits_eval"]))
    print("aadensenet121 golden eval logits rel %.3e" % e)
    assert e < 1e-2


def test_aadensenet121_reference_golden_train_step(dev):
    """BASELINE configs[2] at full size: one training step of aadensenet121 @320 (B = 1) against the logits / loss / gradient
    norms recorded from the REAL reference (tests/golden/nets.json: aadensenet121_320_b1)."""
    import json
    rec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "nets.json")))["aadensenet121_320_b1"]
    model, sd = _build_aa((6, 12, 24, 16), 320, rec["n_classes"], rec["sd_seed"], dev, smooth=False)
    x = synth.xray_batch(rec["x_seed"], rec["B"], rec["S"]).to(dev)
    t = synth.targets(rec["t_seed"], rec["B"], rec["n_classes"]).to(dev)
    model.train()
    loss, logits = model.forward_backward(x, t)
    want = torch.tensor(rec["logits_train"])
    e = _rel(logits.cpu(), want)
    print("aadensenet121 golden train logits rel %.3e loss %.5f (ref %.5f)" % (e, loss.item(), rec["loss"]))
    # B = 1: every BatchNorm of block 4 normalises over 100 values per channel, which amplifies bf16 storage rounding
    # (tests/test_model_gpu.py docstring); north_star's 1e-2 is stated for this dtype
    assert e < 2e-2
    assert abs(loss.item() - rec["loss"]) < 1e-2 * rec["loss"]
    named = dict(model.named_parameters())
    rows = []
    for k in ("classifier.weight", "classifier.bias", "features.norm5.weight", "features.transition3.conv.conv.weight",
              "features.transition3.conv.in_proj_qkv.weight", "features.transition1.conv.key_rel_h",
              "features.transition1.conv.key_rel_w", "features.conv0.weight"):
        l2, ref = named[k].grad.double().norm().item(), rec["grads"][k]["l2"]
        rows.append((k, l2 / ref))
    print("aadensenet121 golden grad l2 ratios: %s" % rows)
    for k, r in rows:
        lim = 0.05 if k.startswith("classifier") or "norm5" in k else 0.15
        assert abs(r - 1) < lim, (k, r)


@pytest.mark.parametrize("dv,det", [(8, True), (48, False), (64, True), (72, True), (104, True)])
def test_out_projection_forward_backward(dev, dv, det):
    """out_proj of AAConv2d (attn_aug_conv.py:92: a dv x dv 1x1 convolution on the attention output) and its backward against torch,
    for the widths of chexpert.py's networks (<= 64) and of the CIFAR Densenet-BC at v = 0.7 (72 / 104: the weight-gradient sums of
    the larger kernel live in registers); slice of a wider block buffer, deterministic statistic rows or atomics."""
    from chexpert_amd import ops
    B, H, W = 3, 6, 7
    ops.set_det_wgrad(det)
    o = synth.uniform(1, (B, H * W, dv), -1.0, 1.0)
    w = synth.uniform(2, (dv, dv, 1, 1), -0.3, 0.3)
    buf = torch.full((B, H, W, dv + 24), -3.0, dtype=torch.bfloat16, device=dev)
    ys = buf[..., 16:16 + dv]
    rows_cap = 64
    S = torch.zeros(2, rows_cap, dv, device=dev)
    rows = ops.aa_outproj_fwd(o.to(dev), w.to(dev), ys, S[0], S[1], stat_rows=rows_cap if det else 0, stat_rstride=dv)
    want = torch.einsum("bpd,cd->bpc", o, w.view(dv, dv)).view(B, H, W, dv)
    got = ys.float().cpu()
    assert (got - want).abs().max().item() < 8e-3 * want.abs().max().item()
    assert (buf[..., :16].float() == -3.0).all() and (buf[..., 16 + dv:].float() == -3.0).all()
    ssum = S[0, :rows].sum(0).cpu() if det else S[0, 0].cpu()
    assert (ssum - got.sum((0, 1, 2))).abs().max().item() < 1e-3 * got.abs().sum((0, 1, 2)).max().item()
    # backward: dY = g*ga + gx*gb + gc; dO = dY W; dW += dY^T O
    g = bf(synth.uniform(3, (B, H, W, dv), -1, 1))
    gx = bf(synth.uniform(4, (B, H, W, dv), -1, 1))
    ga, gb, gc = synth.uniform(5, (dv,), 0.5, 1.5), synth.uniform(6, (dv,), -0.5, 0.5), synth.uniform(7, (dv,), -0.2, 0.2)
    dY = (g * ga + gx * gb + gc).view(B, H * W, dv)
    dO_want = torch.einsum("bpc,cd->bpd", dY, w.view(dv, dv))
    dW_want = torch.einsum("bpc,bpd->cd", dY, o)
    dO = torch.zeros(B, H * W, dv, device=dev)
    dw0 = synth.uniform(8, (dv, dv, 1, 1), -1, 1)
    dW = dw0.clone().to(dev)
    ops.aa_outproj_bwd(g.to(torch.bfloat16).to(dev), gx.to(torch.bfloat16).to(dev), ga.to(dev), gb.to(dev), gc.to(dev), o.to(dev), w.to(dev), dO, dW)
    assert (dO.cpu() - dO_want).abs().max().item() < 1e-4 * dO_want.abs().max().item()
    assert ((dW.cpu() - dw0).view(dv, dv) - dW_want).abs().max().item() < 1e-4 * dW_want.abs().max().item()
    ops.set_det_wgrad(False)


@pytest.mark.parametrize("B,H,W,dv", [(1, 40, 40, 8), (2, 20, 20, 24)])
def test_attention_backward_is_invariant_to_the_gradient_magnitude(dev, B, H, W, dv):
    """The matrix products of the two backward kernels take dS = p (dO . v - delta) as a 16-bit operand (fp16: csrc/aaconv_row.hip,
    aa_op): a training step's dO can be 1e-8 as well as 1e+3, so both kernels scale it by a power of two per workgroup and unscale
    their outputs.  With dO multiplied by 2^-24 or 2^+12 every output is EXACTLY the scaled output of the unscaled call."""
    from chexpert_amd import ops
    nh, dk = 8, 160
    Cq = 2 * dk + dv
    qkv = bf(synth.uniform(1, (B, H, W, Cq), -1.5, 1.5)).to(torch.bfloat16).to(dev)
    rel_h = (synth.uniform(2, (dk // nh, 2 * H - 1), -1, 1) + dk ** -0.5).to(dev)
    rel_w = (synth.uniform(3, (dk // nh, 2 * W - 1), -1, 1) + dk ** -0.5).to(dev)
    d_o = synth.uniform(4, (B, H * W, dv), -1, 1).to(dev)
    o = torch.zeros(B, H * W, dv, device=dev)
    lse = torch.zeros(B * nh, H * W, device=dev)
    ops.aa_attention_fwd(qkv, rel_h, rel_w, o, lse, nh, dk, dv)

    def run(g):
        dqkv = torch.zeros(B, H * W, Cq, device=dev)
        drh, drw = torch.zeros_like(rel_h), torch.zeros_like(rel_w)
        ops.aa_attention_bwd(qkv, rel_h, rel_w, o, g, lse, dqkv, drh, drw, nh, dk, dv)
        return dqkv, drh, drw

    base = run(d_o)
    assert all(torch.isfinite(t).all().item() for t in base) and base[0][..., :dk].abs().max().item() > 0
    for s in (2.0 ** -24, 2.0 ** 12):
        got = run(d_o * s)
        assert torch.equal(got[0], base[0] * s), "dqkv at gradient scale %g: max rel diff %.3e" % (
            s, ((got[0] - base[0] * s).abs().max() / (base[0].abs().max() * s)).item())
        for a, b_, what in zip(got[1:], base[1:], ("d key_rel_h", "d key_rel_w")):
            # (this direct call has no slab workspace: the workgroups' table partials meet in fp32 atomics, whose order varies)
            close(a.cpu() / s, b_.cpu(), 1e-5, what)
