"""GPU: deterministic statistics (CxConv.stat_det and the `stat_rows` mode of the element-wise producers) and deterministic
weight gradients (CxWgrad.scratch: per-split partial tiles in slabs, added in split order).

Every statistics producer of the DenseNet path writes per-workgroup rows with plain stores (no fp32 atomics, partial sums
combined in a fixed order inside the workgroup) and the coefficient kernels sum the rows in row order.  Checked here:
  * kernel level: the row sums equal the atomic mode's totals, and two launches give bit-identical rows;
  * cx_bn_coef / cx_bn_bwd_coef over hundreds of rows and cx_bn_coef_moments (fresh slice from rows) against torch;
  * weight-gradient kernels: the slab mode equals the atomic mode to fp32 rounding and is bit-identical between launches;
  * model level: two consecutive training steps of the same DenseNet / ResNet on the same batch give bit-identical logits, loss,
    running statistics and EVERY parameter gradient.
"""
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


@pytest.fixture
def det_wgrad():
    """The reproducible weight-gradient sums at kernel level: slab workspace on for the test (the DenseNet / ResNet engines switch
    it on themselves, ops.set_det_wgrad)."""
    from chexpert_amd import ops
    keep, ops.WGRAD_SCRATCH_FLOATS = ops.WGRAD_SCRATCH_FLOATS, 16 << 20
    yield
    ops.WGRAD_SCRATCH_FLOATS = keep


def bf(t):
    return t.to(torch.bfloat16).float()


def rnd(seed, shape, lo=-1.0, hi=1.0):
    return synth.uniform(seed, shape, lo, hi)


def nhwc(seed, B, H, W, C, dev, lo=-1.5, hi=1.5):
    return bf(rnd(seed, (B, H, W, C), lo, hi)).to(torch.bfloat16).to(dev)


CAP = 1 << 20


def _both_modes(run, N, dev):
    """run(stat kwargs) launches the producer; returns (atomic totals, det rows of two launches, rows)."""
    tot = torch.zeros(2, N, device=dev)
    run(dict(stat_sum=tot[0], stat_sq=tot[1]))
    slabs = []
    for _ in range(2):
        slab = torch.full((2, CAP), float("nan"), device=dev)
        rows = run(dict(stat_sum=slab[0], stat_sq=slab[1], stat_det=True, stat_replicas=CAP // N, stat_rstride=N))
        assert rows is not None and 1 <= rows <= CAP // N
        slabs.append(slab[:, :rows * N].view(2, rows, N).clone())
    return tot, slabs, rows


def _check(tot, slabs, what, rel=2e-4):
    a, b = slabs
    assert torch.isfinite(a).all(), "%s: a statistic row element was not written" % what
    assert torch.equal(a, b), "%s: two launches differ" % what
    got = a.double().sum(1).float()
    for i in range(2):
        scale = tot[i].abs().max().item() + 1e-6
        err = (got[i] - tot[i]).abs().max().item()
        assert err <= rel * scale, "%s: stat %d rows-sum %.3e away (scale %.3e)" % (what, i, err, scale)


@pytest.mark.parametrize("case", ["pw_fwd", "pw_fwdk", "generic_1x1", "generic_3x3s2", "pool2", "ring_fwd", "strip_fwd", "stem"])
def test_forward_producers_rows_equal_atomic_totals(dev, case):
    from chexpert_amd import ops
    if case in ("pw_fwd", "pw_fwdk", "generic_1x1"):
        B, H, W, K = {"pw_fwd": (4, 24, 24, 160), "pw_fwdk": (2, 12, 12, 384), "generic_1x1": (1, 470, 470, 288)}[case]
        x = nhwc(1, B, H, W, K, dev)
        w = ops.pack_weights(bf(rnd(2, (128, K, 1, 1), -0.1, 0.1)).to(dev))
        pa, pb = rnd(3, (K,), 0.5, 1.5).to(dev), rnd(4, (K,), -0.3, 0.3).to(dev)
        y = torch.empty(B, H, W, 128, dtype=torch.bfloat16, device=dev)
        run = lambda st: ops.conv_gemm(x, w, y, N=128, prologue=ops.PRO_AFFINE_RELU, pa=pa, pb=pb, **st)
        N = 128
    elif case == "generic_3x3s2":
        x = nhwc(5, 2, 18, 18, 64, dev)
        w = ops.pack_weights(bf(rnd(6, (120, 64, 3, 3), -0.1, 0.1)).to(dev))
        y = torch.empty(2, 9, 9, 120, dtype=torch.bfloat16, device=dev)
        run = lambda st: ops.conv_gemm(x, w, y, N=120, kh=3, kw=3, stride=2, pad=1, **st)
        N = 120
    elif case == "pool2":
        x = nhwc(7, 3, 20, 20, 256, dev)
        w = ops.pack_weights(bf(rnd(8, (128, 256, 1, 1), -0.1, 0.1)).to(dev))
        pa, pb = rnd(9, (256,), 0.5, 1.5).to(dev), rnd(10, (256,), -0.3, 0.3).to(dev)
        y = torch.empty(3, 10, 10, 128, dtype=torch.bfloat16, device=dev)
        run = lambda st: ops.conv_gemm(x, w, y, N=128, mode=ops.MODE_POOL2, prologue=ops.PRO_AFFINE_RELU, pa=pa, pb=pb, **st)
        N = 128
    elif case in ("ring_fwd", "strip_fwd"):
        B, H, W = (40, 80, 80) if case == "ring_fwd" else (3, 10, 10)
        x = nhwc(11, B, H, W, 128, dev)
        w = ops.pack_weights(bf(rnd(12, (32, 128, 3, 3), -0.1, 0.1)).to(dev))
        pa, pb = rnd(13, (128,), 0.5, 1.5).to(dev), rnd(14, (128,), -0.3, 0.3).to(dev)
        buf = torch.zeros(B, H, W, 96, dtype=torch.bfloat16, device=dev)
        run = lambda st: ops.conv_gemm(x, w, buf[..., 32:64], N=32, kh=3, kw=3, pad=1, prologue=ops.PRO_AFFINE_RELU, pa=pa, pb=pb, **st)
        N = 32
    else:
        x = ops.nchw3_to_nhwc4(synth.xray_batch(15, 3, 96).to(dev))
        w = ops.pack_weights(bf(rnd(16, (64, 3, 7, 7), -0.1, 0.1)).to(dev), stem=True)
        y = torch.empty(3, 48, 48, 64, dtype=torch.bfloat16, device=dev)
        run = lambda st: ops.conv_gemm(x, w, y, N=64, mode=ops.MODE_STEM, **st)
        N = 64
    tot, slabs, rows = _both_modes(run, N, dev)
    print("%s: %d rows" % (case, rows))
    _check(tot, slabs, case)


@pytest.mark.parametrize("case", ["ring_dgrad", "strip_dgrad", "pw_bwd2", "pw_bwd_narrow", "generic_mask"])
def test_backward_producers_rows_equal_atomic_totals(dev, case):
    from chexpert_amd import ops
    if case in ("ring_dgrad", "strip_dgrad"):
        B, H, W = (40, 80, 80) if case == "ring_dgrad" else (3, 10, 10)
        g, gx = nhwc(21, B, H, W, 32, dev), nhwc(22, B, H, W, 32, dev)
        y1 = nhwc(23, B, H, W, 128, dev)
        w = ops.pack_weights(bf(rnd(24, (32, 128, 3, 3), -0.1, 0.1)).to(dev), transpose=True)
        qa, qb, qc = rnd(25, (32,), 0.5, 1.5).to(dev), rnd(26, (32,), -0.3, 0.3).to(dev), rnd(27, (32,), -0.2, 0.2).to(dev)
        e = [rnd(28 + i, (128,), lo, hi).to(dev) for i, (lo, hi) in enumerate([(-0.3, 1.5), (-0.5, 0.5), (-0.5, 0.5), (0.5, 2.0)])]
        ones = torch.ones(128, device=dev)
        dz = torch.empty(B, H, W, 128, dtype=torch.bfloat16, device=dev)
        run = lambda st: ops.conv_gemm(g, w, dz, N=128, kh=3, kw=3, pad=1, prologue=ops.PRO_AFFINE2, x2=gx, pa=qa, pb=qb, pc=qc,
                                       epilogue=ops.EPI_MASK, ex=y1, e_sc=e[0], e_sh=e[1], e_mu=e[2], e_r=e[3], e_scale=ones, **st)
        N = 128
    elif case in ("pw_bwd2", "pw_bwd_narrow"):
        B, H, W, N = (4, 33, 31, 296) if case == "pw_bwd2" else (2, 9, 10, 96)
        u, v_ = nhwc(31, B, H, W, 128, dev), nhwc(32, B, H, W, 128, dev)
        ex = nhwc(33, B, H, W, N + 8, dev)
        w = ops.pack_weights(bf(rnd(34, (128, N, 1, 1), -0.1, 0.1)).to(dev), transpose=True)
        pa, pb, pc = rnd(35, (128,), 0.5, 1.5).to(dev), rnd(36, (128,), -0.3, 0.3).to(dev), rnd(37, (128,), -0.2, 0.2).to(dev)
        e = [rnd(38 + i, (N,), lo, hi).to(dev) for i, (lo, hi) in
             enumerate([(-0.3, 1.5), (-0.5, 0.5), (-0.5, 0.5), (0.5, 2.0), (-0.3, 1.5)])]
        old0 = nhwc(43, B, H, W, N + 8, dev)

        def run(st):
            old = old0.clone()
            dw = torch.zeros(128, N, 1, 1, device=dev)
            return ops.conv_gemm(u, w, old[..., :N], N=N, prologue=ops.PRO_AFFINE2, x2=v_, pa=pa, pb=pb, pc=pc, epilogue=ops.EPI_MASK,
                                 ex=ex[..., :N], e_sc=e[0], e_sh=e[1], e_mu=e[2], e_r=e[3], e_scale=e[4], accumulate=True, fused_dw=dw,
                                 **st)
    else:       # generic kernel with the mask epilogue (K != 128)
        B, H, W, K, N = 2, 14, 14, 64, 96
        g = nhwc(51, B, H, W, K, dev)
        ex = nhwc(52, B, H, W, N, dev)
        w = ops.pack_weights(bf(rnd(53, (K, N, 1, 1), -0.1, 0.1)).to(dev), transpose=True)
        e = [rnd(54 + i, (N,), lo, hi).to(dev) for i, (lo, hi) in
             enumerate([(-0.3, 1.5), (-0.5, 0.5), (-0.5, 0.5), (0.5, 2.0), (-0.3, 1.5)])]
        dz = torch.empty(B, H, W, N, dtype=torch.bfloat16, device=dev)
        run = lambda st: ops.conv_gemm(g, w, dz, N=N, epilogue=ops.EPI_MASK, ex=ex, e_sc=e[0], e_sh=e[1], e_mu=e[2], e_r=e[3],
                                       e_scale=e[4], **st)
    tot, slabs, rows = _both_modes(run, N, dev)
    print("%s: %d rows" % (case, rows))
    _check(tot, slabs, case, rel=5e-4)


def test_elementwise_producers_rows(dev):
    from chexpert_amd import ops
    B, H, W, Cn = 3, 24, 32, 64
    x = nhwc(60, B, H, W, Cn, dev)
    sc, sh = rnd(61, (Cn,), -0.3, 1.5).to(dev), rnd(62, (Cn,), -0.5, 0.5).to(dev)
    mu, r = rnd(63, (Cn,), -0.5, 0.5).to(dev), rnd(64, (Cn,), 0.5, 2.0).to(dev)
    y = torch.zeros(B, H // 2, W // 2, Cn, dtype=torch.bfloat16, device=dev)
    amax = torch.zeros(B, H // 2, W // 2, Cn, dtype=torch.uint8, device=dev)

    def rows_mode(fn, C, cap):
        tot = torch.zeros(2, C, device=dev)
        fn(tot[0], tot[1], 0)
        outs = []
        for _ in range(2):
            slab = torch.full((2, cap * C), float("nan"), device=dev)
            rows = fn(slab[0], slab[1], cap)
            assert 1 <= rows <= cap
            outs.append(slab[:, :rows * C].view(2, rows, C).clone())
        return tot, outs, rows
    tot, outs, rows = rows_mode(lambda a, b, cap: ops.bnrelu_maxpool_fwd(x, sc, sh, y, amax, a, b, stat_rows=cap), Cn, 7)
    assert rows == 7
    _check(tot, outs, "maxpool fwd")
    g, gx = nhwc(65, B, H // 2, W // 2, Cn, dev), nhwc(66, B, H // 2, W // 2, Cn, dev)
    ga, gb, gc = rnd(67, (Cn,), 0.5, 1.5).to(dev), rnd(68, (Cn,), -0.3, 0.3).to(dev), rnd(69, (Cn,), -0.2, 0.2).to(dev)
    dz = torch.zeros(B, H, W, Cn, dtype=torch.bfloat16, device=dev)
    tot, outs, rows = rows_mode(lambda a, b, cap: ops.bnrelu_maxpool_bwd(x, sc, sh, mu, r, amax, g, gx, ga, gb, gc, dz, a, b,
                                                                         stat_rows=cap), Cn, 64)
    _check(tot, outs, "maxpool bwd", rel=5e-4)
    # un-pool + mask (transition backward) and the head's global-average-pool backward
    C2 = 256
    d = nhwc(70, B, H // 2, W // 2, C2, dev)
    x2 = nhwc(71, B, H, W, C2, dev)
    v5 = [rnd(72 + i, (C2,), lo, hi).to(dev) for i, (lo, hi) in enumerate([(-0.3, 1.5), (-0.5, 0.5), (-0.5, 0.5), (0.5, 2.0), (-0.3, 1.5)])]
    gout = torch.zeros(B, H, W, C2, dtype=torch.bfloat16, device=dev)
    tot, outs, rows = rows_mode(lambda a, b, cap: ops.unpool2_mask(d, x2, v5[0], v5[1], v5[2], v5[3], v5[4], gout, a, b, stat_rows=cap),
                                C2, 100)
    _check(tot, outs, "unpool2 mask", rel=5e-4)
    dp = rnd(80, (B, C2)).to(dev)
    tot, outs, rows = rows_mode(lambda a, b, cap: ops.gap_relu_bn_bwd(dp, x2, v5[0], v5[1], v5[2], v5[3], v5[4], gout, a, b,
                                                                      stat_rows=cap), C2, B)
    assert rows == B
    _check(tot, outs, "gap backward", rel=5e-4)


def test_coefficient_kernels_over_many_rows(dev):
    from chexpert_amd import ops
    Cn, rows, n = 200, 777, 5000.0
    part = rnd(90, (2, rows, Cn), 0.0, 1.0)
    part[1] = part[1] + part[0] ** 2 + 5.0                 # keeps the variance positive
    gamma, beta = rnd(91, (Cn,), 0.5, 1.5), rnd(92, (Cn,), -0.5, 0.5)
    rm, rv = rnd(93, (Cn,)), rnd(94, (Cn,), 0.5, 1.5)
    S = part.double().sum(1)
    mean = S[0] / n
    var = S[1] / n - mean ** 2
    rstd = 1 / torch.sqrt(var + 1e-5)
    pd = part.to(dev)
    out = [torch.zeros(Cn, device=dev) for _ in range(4)]
    rmd, rvd = rm.clone().to(dev), rv.clone().to(dev)
    ops.bn_coef(pd[0], pd[1], n, gamma.to(dev), beta.to(dev), 1e-5, 0.1, rmd, rvd, out[0], out[1], out[2], out[3], Cn,
                replicas=rows, rstride=Cn)
    for got, want in ((out[0], gamma * rstd), (out[1], beta - mean * gamma * rstd), (out[2], mean), (out[3], rstd),
                      (rmd, 0.9 * rm + 0.1 * mean), (rvd, 0.9 * rv + 0.1 * var * n / (n - 1))):
        assert (got.cpu().double() - want).abs().max().item() <= 2e-6 * want.abs().max().item()
    # moments: channels [40, 72) fresh from 33 rows, the others already in mean / rstd
    mean_d, rstd_d = rnd(95, (Cn,)).to(dev), rnd(96, (Cn,), 0.5, 2.0).to(dev)
    m0, r0 = mean_d.cpu().clone(), rstd_d.cpu().clone()
    fr = rnd(97, (2, 33, 32), 0.0, 1.0)
    fr[1] = fr[1] + fr[0] ** 2 + 3.0
    frd = fr.to(dev)
    sc, sh = torch.zeros(Cn, device=dev), torch.zeros(Cn, device=dev)
    rmd, rvd = rm.clone().to(dev), rv.clone().to(dev)
    ops.bn_coef_moments(mean_d, rstd_d, 300.0, gamma.to(dev), beta.to(dev), 1e-5, 0.1, rmd, rvd, sc, sh, Cn,
                        fresh=(frd[0], frd[1], 33, 32, 40, 32))
    Sf = fr.double().sum(1)
    mf = Sf[0] / 300.0
    vf = Sf[1] / 300.0 - mf ** 2
    m_ref, r_ref = m0.double().clone(), r0.double().clone()
    m_ref[40:72], r_ref[40:72] = mf, 1 / torch.sqrt(vf + 1e-5)
    for got, want in ((mean_d, m_ref), (rstd_d, r_ref), (sc, gamma * r_ref), (sh, beta - m_ref * gamma * r_ref),
                      (rmd, 0.9 * rm + 0.1 * m_ref), (rvd, 0.9 * rv + 0.1 * (1 / r_ref ** 2 - 1e-5) * 300.0 / 299.0)):
        assert (got.cpu().double() - want).abs().max().item() <= 4e-6 * want.abs().max().item()
    # backward coefficients over many rows
    S12 = rnd(98, (2, rows, Cn), -1.0, 1.0).to(dev)
    dg, db, A, Bc = (torch.zeros(Cn, device=dev) for _ in range(4))
    pa, pb, pc = (torch.zeros(Cn, device=dev) for _ in range(3))
    mu, rr = rnd(99, (Cn,)), rnd(100, (Cn,), 0.5, 2.0)
    ops.bn_bwd_coef(S12[0], S12[1], n, gamma.to(dev), mu.to(dev), rr.to(dev), dg, db, A, Bc, pa, pb, pc, Cn, replicas=rows, rstride=Cn)
    s1, s2 = S12[0].cpu().double().sum(0), S12[1].cpu().double().sum(0)
    for got, want in ((dg, s2), (db, s1), (A, rr * gamma * s1 / n), (Bc, rr * gamma * s2 / n), (pa, gamma * rr),
                      (pb, -gamma * rr * rr * s2 / n), (pc, gamma * rr * (mu * rr * s2 - s1) / n)):
        assert (got.cpu().double() - want).abs().max().item() <= 2e-5 * (want.abs().max().item() + 1e-6)


@pytest.mark.parametrize("cfg,B,S", [((2, 2, 2, 2), 4, 64), ((6, 12, 24, 16), 2, 320)])
def test_training_step_is_reproducible_bit_for_bit(dev, cfg, B, S):
    from chexpert_amd.models import DenseNet
    torch.manual_seed(11)
    model = DenseNet(32, cfg, 64, num_classes=5).to(dev).train()
    assert model._eng().det
    x, t = synth.xray_batch(500, B, S).to(dev), synth.targets(501, B, 5).to(dev)
    runs = []
    for _ in range(3):
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        model.zero_grad()
        loss, logits = model.forward_backward(x, t)
        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        runs.append((loss.clone(), logits.clone(), grads, {k: v.clone() for k, v in model.state_dict().items() if "running" in k}))
        model.load_state_dict(sd)                   # the next run starts from the same running statistics
    for i in (1, 2):
        assert torch.equal(runs[0][0], runs[i][0]) and torch.equal(runs[0][1], runs[i][1]), "loss / logits differ between runs"
        for k in runs[0][3]:
            assert torch.equal(runs[0][3][k], runs[i][3][k]), k
        for k, g0 in runs[0][2].items():
            # statistic rows and weight-gradient slabs: no sum on the path depends on the order workgroups finish in
            assert torch.equal(g0, runs[i][2][k]), "%s differs between runs" % k


def test_resnet_training_step_is_reproducible_bit_for_bit(dev):
    """The plain Bottleneck ResNet engine on deterministic statistic rows (conv epilogues, the residual-join backward
    cx_relu_bwd_stats with its three sums, the stem pool backward)."""
    from chexpert_amd.models import Bottleneck, ResNet
    torch.manual_seed(12)
    model = ResNet(Bottleneck, [1, 2, 2, 1], num_classes=5).to(dev).train()
    assert model._eng().det
    x, t = synth.xray_batch(510, 4, 64).to(dev), synth.targets(511, 4, 5).to(dev)
    runs = []
    for _ in range(3):
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        model.zero_grad()
        loss, logits = model.forward_backward(x, t)
        runs.append((loss.clone(), logits.clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters()}))
        model.load_state_dict(sd)
    for i in (1, 2):
        assert torch.equal(runs[0][0], runs[i][0]) and torch.equal(runs[0][1], runs[i][1])
        for k, g0 in runs[0][2].items():
            assert torch.equal(g0, runs[i][2][k]), "%s differs between runs" % k


def _three_identical_steps(model, x, t):
    runs = []
    for _ in range(3):
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        model.zero_grad()
        loss, logits = model.forward_backward(x, t)
        runs.append((loss.clone(), logits.clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters()},
                     {k: v.clone() for k, v in model.state_dict().items() if "running" in k}))
        model.load_state_dict(sd)
    for i in (1, 2):
        assert torch.equal(runs[0][0], runs[i][0]) and torch.equal(runs[0][1], runs[i][1]), "loss / logits differ between runs"
        for k in runs[0][3]:
            assert torch.equal(runs[0][3][k], runs[i][3][k]), k
        for k, g0 in runs[0][2].items():
            assert torch.equal(g0, runs[i][2][k]), "%s differs between runs" % k


@pytest.mark.parametrize("cfg,B,S", [((6, 4, 2, 2), 4, 64), ((6, 12, 24, 16), 2, 320)])
def test_aa_densenet_training_step_is_reproducible_bit_for_bit(dev, cfg, B, S):
    """The attention-augmented DenseNet (chexpert.py:475-480): InstanceNorm sums with one owner per (image, channel), the two
    producers of a block's first channels reduced one after the other, attention table gradients and the out-projection weight
    gradient through slabs summed in workgroup order."""
    from chexpert_amd.models import DenseNet
    torch.manual_seed(13)
    model = DenseNet(32, cfg, 64, num_classes=5, attn_params={"k": 0.2, "v": 0.1, "nh": 8, "relative": True, "input_dims": (S, S)})
    model = model.to(dev).train()
    assert model._eng().det
    _three_identical_steps(model, synth.xray_batch(520, B, S).to(dev), synth.targets(521, B, 5).to(dev))


@pytest.mark.parametrize("kind,layers,B,S", [("bottleneck", [1, 2, 2, 1], 4, 64), ("bottleneck", [3, 8, 36, 3], 1, 320),
                                             ("basic", [2, 2, 2, 2], 4, 128)])
def test_aa_resnet_training_step_is_reproducible_bit_for_bit(dev, kind, layers, B, S):
    """aaresnet (chexpert.py:486-494; BasicBlock form: models/test_model.py --attn): AAConv2d inside the blocks."""
    from chexpert_amd.models import BasicBlock, Bottleneck, ResNet
    torch.manual_seed(14)
    model = ResNet(Bottleneck if kind == "bottleneck" else BasicBlock, layers, num_classes=5,
                   attn_params={"k": 0.2, "v": 0.1, "nh": 8, "relative": True, "input_dims": (S, S)}).to(dev).train()
    assert model._eng().det
    _three_identical_steps(model, synth.xray_batch(530, B, S).to(dev), synth.targets(531, B, 5).to(dev))


@pytest.mark.parametrize("name,B,S", [("efficientnet-b0", 4, 96), ("efficientnet-b0", 2, 224), ("efficientnet-b4", 2, 380)])
def test_efficientnet_training_step_is_reproducible_bit_for_bit(dev, name, B, S):
    """EfficientNet (models/efficientnet.py:78-185): depthwise / Swish / BatchNorm statistics as rows, squeeze-excite sums with one
    owner each, depthwise and 1x1 weight gradients through slabs -- with the stochastic parts (Dropout / DropConnect masks) fixed by
    the model's counter-based seeds, two steps are the same bits."""
    from chexpert_amd.models import construct_model
    from chexpert_amd.models.efficientnet import DropMarker
    torch.manual_seed(15)
    model = construct_model(name, 5).to(dev).train()
    for mod in model.modules():
        if isinstance(mod, DropMarker):
            mod.p = 0.0
    assert model._eng().det
    _three_identical_steps(model, synth.xray_batch(540, B, S).to(dev), synth.targets(541, B, 5).to(dev))


@pytest.mark.parametrize("case", ["fused_1x1", "fused_1x1_narrow", "strip_3x3", "strip_3x3_wide", "ring_3x3", "pool2", "stem", "generic_3x3s2",
                                  "wgrad_mm"])
def test_weight_gradient_slabs_equal_atomics_and_repeat_bit_for_bit(dev, det_wgrad, case):
    """Every weight-gradient kernel of the DenseNet / ResNet paths: with the slab workspace (ops.wgrad_scratch) two
    launches give identical bits, and the result equals the atomic mode's to fp32 rounding."""
    from chexpert_amd import ops
    ones = lambda n: torch.ones(n, device=dev)
    zeros = lambda n: torch.zeros(n, device=dev)
    vec = lambda seed, n, lo, hi: rnd(seed, (n,), lo, hi).to(dev)
    if case in ("fused_1x1", "fused_1x1_narrow"):
        B, H, W, cin = (6, 20, 20, 352) if case == "fused_1x1" else (3, 16, 16, 64)
        dz, y1 = nhwc(60, B, H, W, 128, dev), nhwc(61, B, H, W, 128, dev)
        xs, gbuf = nhwc(62, B, H, W, cin, dev), nhwc(63, B, H, W, cin, dev)
        w = ops.pack_weights(bf(rnd(64, (128, cin, 1, 1), -0.1, 0.1)).to(dev), transpose=True)
        st = torch.zeros(2, 64 * cin, device=dev)
        def run(dw):
            ops.conv_gemm(dz, w, gbuf.clone(), N=cin, prologue=ops.PRO_AFFINE2, x2=y1, pa=vec(65, 128, .5, 1.5), pb=vec(66, 128, -.3, .3),
                          pc=vec(67, 128, -.2, .2), epilogue=ops.EPI_MASK, ex=xs, e_sc=vec(68, cin, .5, 1.5), e_sh=vec(69, cin, -.3, .3),
                          e_mu=zeros(cin), e_r=ones(cin), e_scale=ones(cin), stat_sum=st[0], stat_sq=st[1], stat_det=True,
                          stat_replicas=64, stat_rstride=cin, accumulate=True, fused_dw=dw)
        shape = (128, cin, 1, 1)
    else:
        ksz, K, N, B, H, W, mode, stride, xpro = {
            "strip_3x3": (3, 128, 32, 5, 20, 20, ops.MODE_CONV, 1, 1), "strip_3x3_wide": (3, 256, 96, 2, 20, 20, ops.MODE_CONV, 1, 1),
            "ring_3x3": (3, 128, 32, 4, 80, 80, ops.MODE_CONV, 1, 1), "pool2": (1, 256, 128, 3, 20, 20, ops.MODE_POOL2, 1, 1),
            "stem": (7, 32, 64, 3, 96, 96, ops.MODE_STEM, 2, 0), "generic_3x3s2": (3, 64, 128, 2, 18, 18, ops.MODE_CONV, 2, 1),
            "wgrad_mm": (1, 256, 128, 4, 16, 16, ops.MODE_CONV, 1, 1)}[case]
        if mode == ops.MODE_STEM:
            x = ops.nchw3_to_nhwc4(synth.xray_batch(70, B, H).to(dev))
            Ho, Wo = H // 2, W // 2
        else:
            x = nhwc(70, B, H, W, K, dev)
            Ho, Wo = (H // 2, W // 2) if (mode == ops.MODE_POOL2 or stride == 2) else (H, W)
        g, g2 = nhwc(71, B, Ho, Wo, N, dev), nhwc(72, B, Ho, Wo, N, dev)
        kw = dict(kh=ksz, kw=ksz, stride=stride, pad=ksz // 2) if mode == ops.MODE_CONV else dict(mode=mode)
        if mode == ops.MODE_STEM:
            kw["K"] = 32
        if xpro:
            kw.update(x_prologue=ops.PRO_AFFINE_RELU, pa=vec(73, K, .5, 1.5), pb=vec(74, K, -.3, .3))
        def run(dw):
            ops.conv_wgrad(g, x, dw, g_prologue=ops.PRO_AFFINE2, g2=g2, ga=vec(75, N, .5, 1.5), gb=vec(76, N, -.3, .3), gc=vec(77, N, -.2, .2), **kw)
        shape = (N, 3, 7, 7) if mode == ops.MODE_STEM else (N, K, ksz, ksz)
    assert ops.WGRAD_SCRATCH_FLOATS > 0
    outs = []
    for _ in range(2):
        dw = torch.zeros(shape, device=dev)
        run(dw)
        outs.append(dw)
    keep, ops.WGRAD_SCRATCH_FLOATS = ops.WGRAD_SCRATCH_FLOATS, 0
    try:
        dwa = torch.zeros(shape, device=dev)
        run(dwa)
    finally:
        ops.WGRAD_SCRATCH_FLOATS = keep
    assert torch.equal(outs[0], outs[1]), "%s: two slab-mode launches differ" % case
    scale = dwa.abs().max().item() + 1e-12
    assert dwa.abs().max().item() > 0
    err = (outs[0] - dwa).abs().max().item()
    assert err <= 2e-5 * scale, "%s: slab mode %.3e away from the atomic mode (scale %.3e)" % (case, err, scale)
