"""GPU: the wide-channel implicit-GEMM kernel (csrc/conv_mm.hip: K % 64 == 0, N % 128 == 0 -- the ResNet bottleneck shapes) through
cx_conv_gemm against a PyTorch fp32 reference of the same op, for both tile forms, and against the generic kernel it replaces."""
import ctypes

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
def select():
    from chexpert_amd import ops as _ops          # per-call CxConv.kernel_hint (ABI 10), defaulted through ops.KERNEL_HINT

    def sel(on, wm):
        _ops.KERNEL_HINT = _ops.kernel_hint(on, wm)
    yield sel
    _ops.KERNEL_HINT = 0


def bf(t):
    return t.to(torch.bfloat16).float()


def rnd(seed, shape, lo=-1.0, hi=1.0):
    return synth.uniform(seed, shape, lo, hi)


def nhwc(seed, B, H, W, C, dev, lo=-1.5, hi=1.5):
    v = bf(rnd(seed, (B, H, W, C), lo, hi))
    return v.to(torch.bfloat16).to(dev), v.permute(0, 3, 1, 2).contiguous()


def to_nchw(t):
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


def close(got, want, rel=6e-3, what=""):
    scale = want.abs().max().item() + 1e-6
    err = (got - want).abs().max().item()
    assert err <= rel * scale, "%s: max err %.3e vs scale %.3e (rel %.2e)" % (what, err, scale, err / scale)


cv = lambda t: t.view(1, -1, 1, 1)


@pytest.mark.parametrize("form", [1, 3])
@pytest.mark.parametrize("B,H,W,K,N,ksz,stride,pro", [
    (2, 9, 11, 64, 256, 1, 1, 0),        # one k-step, ragged last pixel tile
    (3, 10, 12, 128, 256, 1, 1, 1),      # two n tiles
    (2, 12, 10, 64, 128, 3, 1, 1),       # 9 steps (odd)
    (2, 11, 13, 128, 128, 3, 2, 1),      # strided 3x3, 18 steps
    (5, 8, 8, 192, 128, 1, 2, 0),        # strided 1x1 (downsample), 3 steps
    (1, 20, 20, 256, 384, 3, 1, 1),      # 36 steps, three n tiles, M = 400 (1.56 tiles of 256)
    (2, 10, 12, 120, 128, 3, 1, 1),      # K not a multiple of 64: the second channel step of every tap is partial (56 of 64)
    (3, 9, 9, 200, 256, 1, 1, 0),        # 3 full steps + 8 channels
    (2, 12, 12, 72, 256, 3, 2, 1),       # one chunk past the first step, strided
    (2, 10, 12, 192, 24, 1, 1, 0),       # N % 8 tiles (EfficientNet projections): 24 of a 128-wide tile
    (3, 9, 9, 144, 56, 1, 1, 1),
    (2, 10, 10, 128, 328, 3, 1, 1),      # 2.56 tiles of 128 (AAConv query/key/value width)
    (2, 8, 8, 256, 160, 1, 2, 0),        # one full tile + 32 channels, strided
    (1, 12, 12, 384, 200, 1, 1, 1),
])
def test_forward_against_torch(dev, select, form, B, H, W, K, N, ksz, stride, pro):
    from chexpert_amd import ops
    select(1, form)          # 1 = 128 x 128, 3 = 128 x 256 (falls back to 1 where N % 256 != 0)
    xb, x = nhwc(5, B, H, W, K + 64, dev)            # the operand is a channel slice of a wider buffer
    xb, x = xb[..., 32:32 + K], x[:, 32:32 + K]
    w = bf(rnd(6, (N, K, ksz, ksz), -0.1, 0.1))
    pa, pb = rnd(7, (K,), -0.3, 1.5), rnd(8, (K,), -0.5, 0.5)
    a = bf(F.relu(x * cv(pa) + cv(pb))) if pro else x
    want = F.conv2d(a, w, stride=stride, padding=ksz // 2)
    Ho, Wo = want.shape[2:]
    buf = torch.full((B, Ho, Wo, 96 + N), -3.0, dtype=torch.bfloat16, device=dev)
    rows_cap = 64
    ssum, ssq = torch.zeros(rows_cap, N, device=dev), torch.zeros(rows_cap, N, device=dev)
    rows = ops.conv_gemm(xb, ops.pack_weights(w.to(dev)), buf[..., 96:], N=N, kh=ksz, kw=ksz, stride=stride, pad=ksz // 2,
                         prologue=ops.PRO_AFFINE_RELU if pro else ops.PRO_NONE, pa=pa.to(dev) if pro else None,
                         pb=pb.to(dev) if pro else None, stat_sum=ssum, stat_sq=ssq, stat_det=True, stat_replicas=rows_cap,
                         stat_rstride=N)
    M = B * Ho * Wo
    assert rows == (M + 127) // 128, "statistic rows = pixel tiles of the selected kernel"
    got = to_nchw(buf[..., 96:])
    close(got, want, what="y")
    assert (buf[..., :96].float() == -3.0).all()
    close(ssum[:rows].sum(0).cpu(), got.sum((0, 2, 3)), rel=1e-4, what="sum")
    close(ssq[:rows].sum(0).cpu(), (got * got).sum((0, 2, 3)), rel=1e-4, what="sum of squares")
    assert (ssum[rows:] == 0).all()


@pytest.mark.parametrize("form", [1, 3])
@pytest.mark.parametrize("B,H,W,K,N,ksz,ts,acc,pro", [
    (2, 9, 10, 128, 128, 1, 1, False, 2),
    (2, 9, 10, 256, 128, 1, 1, True, 2),
    (3, 10, 12, 64, 256, 3, 1, False, 2),
    (2, 6, 7, 128, 128, 3, 2, False, 2),        # input gradient of a stride-2 3x3: four parity-class launches (1, 2, 2, 4 taps)
    (3, 7, 5, 64, 256, 3, -2, True, 2),         # the same with odd input sizes (13 x 9): classes of unequal size
    (2, 6, 7, 128, 256, 1, 2, True, 2),         # stride-2 1x1 (downsample), accumulating: only the even-even class has a tap
    (2, 6, 7, 128, 128, 1, 2, False, 0),        # storing: the untouched classes need zeros (declined, generic kernel)
    (2, 6, 7, 328, 256, 1, 2, True, 0),         # AAConv query/key/value projection gradient: K = 2 dk + dv, accumulating
    (2, 5, 6, 120, 128, 3, 2, False, 2),        # AAConv 3x3 branch gradient: K = planes - dv
    (4, 10, 10, 128, 256, 3, 1, True, 2),       # 128 x 256 tiles with the two-tensor operand
    (2, 8, 8, 192, 128, 1, 1, True, 0),         # plain operand, accumulate (AA projection gradient form)
    (2, 9, 10, 192, 120, 1, 1, False, 2),       # N % 8 tiles: 120 of 128
    (2, 6, 7, 128, 40, 3, 2, False, 2),         # ... through the parity classes
    (2, 8, 8, 256, 328, 1, 1, True, 0),         # ... 2.56 tiles, accumulating
])
def test_input_gradient_mask_epilogue_against_torch(dev, select, form, B, H, W, K, N, ksz, ts, acc, pro):
    from chexpert_amd import ops
    select(1, form)
    odd, ts = (1, -ts) if ts < 0 else (0, ts)                    # negative ts: odd forward input size (2H-1, 2W-1)
    Ho, Wo = (H * ts - odd, W * ts - odd) if ts > 1 else (H, W)  # forward input size
    ub, u = nhwc(20, B, H, W, K, dev)
    vb, v = nhwc(21, B, H, W, K, dev)
    exb, ex = nhwc(22, B, Ho, Wo, N + 32, dev)
    oldb, old = nhwc(23, B, Ho, Wo, N + 32, dev)
    w = bf(rnd(24, (K, N, ksz, ksz), -0.1, 0.1))                  # forward weight (O=K, I=N): the gradient maps K -> N
    pa, pb, pc = rnd(25, (K,), 0.5, 1.5), rnd(26, (K,), -0.3, 0.3), rnd(27, (K,), -0.2, 0.2)
    e_sc, e_sh = rnd(28, (N,), -0.3, 1.5), rnd(29, (N,), -0.5, 0.5)
    e_mu, e_r, e_scale = rnd(30, (N,), -0.5, 0.5), rnd(31, (N,), 0.5, 2.0), rnd(32, (N,), -0.3, 1.5)
    dy = bf(u * cv(pa) + v * cv(pb) + cv(pc)) if pro == 2 else u
    acc_ref = F.conv_transpose2d(dy, w, stride=ts, padding=ksz // 2, output_padding=(ts - 1 - odd) if (ksz == 3 or ts > 1) else 0)
    assert acc_ref.shape[2:] == (Ho, Wo)
    exs = ex[:, :N]
    dz = torch.where((exs * cv(e_sc) + cv(e_sh)) > 0, acc_ref, torch.zeros(()))
    want = cv(e_scale) * dz + (old[:, :N] if acc else 0)
    S1, S2 = dz.sum((0, 2, 3)), (dz * (exs - cv(e_mu)) * cv(e_r)).sum((0, 2, 3))
    cap = 64
    s1, s2 = torch.zeros(cap, N, device=dev), torch.zeros(cap, N, device=dev)
    kw = dict(prologue=ops.PRO_AFFINE2, x2=vb, pa=pa.to(dev), pb=pb.to(dev), pc=pc.to(dev)) if pro == 2 else {}
    rows = ops.conv_gemm(ub, ops.pack_weights(w.to(dev), transpose=True), oldb[..., :N], N=N, kh=ksz, kw=ksz, pad=ksz // 2 if ts == 1 else ksz - 1 - ksz // 2,
                         tstride=ts, epilogue=ops.EPI_MASK, ex=exb[..., :N], e_sc=e_sc.to(dev), e_sh=e_sh.to(dev), e_mu=e_mu.to(dev),
                         e_r=e_r.to(dev), e_scale=e_scale.to(dev), stat_sum=s1, stat_sq=s2, stat_det=True, stat_replicas=cap,
                         stat_rstride=N, accumulate=acc, **kw)
    close(to_nchw(oldb[..., :N]), want, rel=8e-3, what="g")
    assert torch.equal(to_nchw(oldb[..., N:]), old[:, N:]), "wrote outside the slice"
    close(s1[:rows].sum(0).cpu(), S1, rel=2e-3, what="S1")
    close(s2[:rows].sum(0).cpu(), S2, rel=2e-3, what="S2")


def test_same_result_as_the_generic_kernel(dev, select):
    """AFFINE2 operand, plain store, no statistics: the wide-channel kernel and conv_gemm.hip's 128x128x32 kernel agree to bf16
    rounding of identical fp32 sums (different accumulation order only)."""
    from chexpert_amd import ops
    B, H, W, K, N = 4, 20, 20, 256, 1024
    ub, _ = nhwc(40, B, H, W, K, dev)
    vb, _ = nhwc(41, B, H, W, K, dev)
    wp = ops.pack_weights(bf(rnd(42, (N, K, 1, 1), -0.1, 0.1)).to(dev))
    pa, pb, pc = (rnd(43 + i, (K,), -1.0, 1.0).to(dev) for i in range(3))
    outs = []
    for on, form in [(0, 0), (1, 1), (1, 3)]:
        select(on, form)
        y = torch.empty(B, H, W, N, dtype=torch.bfloat16, device=dev)
        ops.conv_gemm(ub, wp, y, N=N, prologue=ops.PRO_AFFINE2, x2=vb, pa=pa, pb=pb, pc=pc)
        outs.append(y.float())
    scale = outs[0].abs().max().item()
    for o in outs[1:]:
        assert (o - outs[0]).abs().max().item() <= 8e-3 * scale
    assert torch.equal(outs[1], outs[2]), "both tile forms add the k-steps in the same order"


# ------------------------------------------------------------------------------------------------ weight gradient (wgrad_mm.hip)
@pytest.fixture
def select_w():
    from chexpert_amd import ops as _ops

    def sel(on, form):
        _ops.KERNEL_HINT = _ops.kernel_hint(on, form)
    yield sel
    _ops.KERNEL_HINT = 0


@pytest.mark.parametrize("form", [1, 2, 3])
@pytest.mark.parametrize("K,N,gpro,xpro,B,H,W,splits", [
    (256, 128, 2, 1, 2, 8, 16, 0),        # one 256-channel tile, four steps per image pair
    (64, 256, 2, 1, 3, 8, 8, 2),          # partial channel tile (64 of 256 / 128), three steps split in two ranges
    (320, 128, 0, 0, 1, 16, 20, 5),       # plain operands, two channel tiles with a partial second one, one step per range
    (512, 256, 2, 0, 4, 12, 16, 3),       # 12 steps in three ranges (pairs + tails of the step pipeline)
    (128, 384, 0, 1, 7, 8, 8, 1),         # 7 steps in one range (odd count)
    (192, 24, 2, 1, 2, 8, 8, 0),          # partial last N tile (EfficientNet projections): 24 of 128
    (144, 120, 0, 0, 3, 8, 8, 2),         # 120 of 128, partial channel tile as well
    (64, 328, 2, 0, 1, 8, 16, 1),         # 2.56 tiles
])
def test_1x1_weight_gradient_against_torch(dev, select_w, form, K, N, gpro, xpro, B, H, W, splits):
    from chexpert_amd import ops
    select_w(1, form)           # 1 = 128 x 128, 2 = 256 x 128 (needs N % 256 == 0, else 1), 3 = 128 x 256
    gb_, g = nhwc(40, B, H, W, N + 32, dev)
    g2b, g2 = nhwc(41, B, H, W, N, dev)
    xb, x = nhwc(42, B, H, W, K + 64, dev)
    ga, gbv, gc = rnd(43, (N,), 0.5, 1.5), rnd(44, (N,), -0.3, 0.3), rnd(45, (N,), -0.2, 0.2)
    pa, pb = rnd(46, (K,), -0.3, 1.5), rnd(47, (K,), -0.5, 0.5)
    G = bf(g[:, :N] * cv(ga) + g2 * cv(gbv) + cv(gc)) if gpro else g[:, :N]
    A = bf(F.relu(x[:, 32:32 + K] * cv(pa) + cv(pb))) if xpro else x[:, 32:32 + K]
    want = torch.nn.grad.conv2d_weight(A, (N, K, 1, 1), G)
    dw0 = rnd(48, (N, K, 1, 1), -1, 1)
    dw = dw0.clone().to(dev)
    ops.conv_wgrad(gb_[..., :N], xb[..., 32:32 + K], dw, g_prologue=gpro, g2=g2b if gpro else None, ga=ga.to(dev), gb=gbv.to(dev),
                   gc=gc.to(dev), x_prologue=xpro, pa=pa.to(dev), pb=pb.to(dev), splits=splits)
    close(dw.cpu() - dw0, want, rel=2e-3, what="dW")


@pytest.mark.parametrize("K,N,gpro,xpro,B,H,W,splits", [
    (128, 128, 2, 1, 2, 10, 10, 0),        # 240 padded positions: four steps, the last one ragged
    (256, 128, 2, 1, 3, 7, 20, 2),         # two channel tiles, 462 positions in two ranges
    (128, 256, 0, 1, 1, 40, 40, 5),        # plain gradient operand, 1680 positions in five ranges (tails of the step pipeline)
    (128, 128, 2, 0, 5, 4, 6, 1),          # plain activation operand, tiny map (every row touches the zero rows)
    (128, 128, 2, 1, 2, 80, 80, 3),        # one padded row (82 positions) longer than a step
    (128, 120, 2, 1, 2, 20, 20, 2),        # partial last N tile: the 3x3 branch of an attention-augmented transition (128 - 8 channels)
    (256, 240, 0, 1, 3, 10, 12, 1),        # two N tiles, the second partial; plain gradient operand
    (128, 104, 2, 0, 2, 9, 7, 3),          # partial tile, plain activation operand
])
def test_3x3_weight_gradient_against_torch(dev, select_w, K, N, gpro, xpro, B, H, W, splits):
    """wgrad3_kernel: the three taps of a kernel row per workgroup, pixels walked in the zero-padded index space.  (It leaves its
    partial tiles through the slab workspace only; without one the strip kernel runs.)"""
    from chexpert_amd import ops
    keep, ops.WGRAD_SCRATCH_FLOATS = ops.WGRAD_SCRATCH_FLOATS, 16 << 20
    gb_, g = nhwc(50, B, H, W, N + 32, dev)
    g2b, g2 = nhwc(51, B, H, W, N, dev)
    xb, x = nhwc(52, B, H, W, K + 64, dev)
    ga, gbv, gc = rnd(53, (N,), 0.5, 1.5), rnd(54, (N,), -0.3, 0.3), rnd(55, (N,), -0.2, 0.2)
    pa, pb = rnd(56, (K,), -0.3, 1.5), rnd(57, (K,), -0.5, 0.5)
    G = bf(g[:, :N] * cv(ga) + g2 * cv(gbv) + cv(gc)) if gpro else g[:, :N]
    A = bf(F.relu(x[:, 32:32 + K] * cv(pa) + cv(pb))) if xpro else x[:, 32:32 + K]
    want = torch.nn.grad.conv2d_weight(A, (N, K, 3, 3), G, padding=1)
    outs = []
    for form in (3, 0):                     # wgrad3_kernel, then the strip kernel it replaces on these shapes
        select_w(1, form)
        dw0 = rnd(58, (N, K, 3, 3), -1, 1)
        dw = dw0.clone().to(dev)
        ops.conv_wgrad(gb_[..., :N], xb[..., 32:32 + K], dw, kh=3, kw=3, pad=1, g_prologue=gpro, g2=g2b if gpro else None, ga=ga.to(dev),
                       gb=gbv.to(dev), gc=gc.to(dev), x_prologue=xpro, pa=pa.to(dev), pb=pb.to(dev), splits=splits if form else 0)
        outs.append(dw.cpu() - dw0)
        if form == 3 or xpro:               # (the strip kernel takes the BN + ReLU activation operand only)
            close(outs[-1], want, rel=2e-3, what="dW (form %d)" % form)
    ops.WGRAD_SCRATCH_FLOATS = keep


@pytest.mark.parametrize("K,N,gpro,xpro,B,H,W,splits", [
    (128, 128, 2, 1, 2, 10, 10, 0),        # 5 x 5 gradient image: 60 padded positions, one ragged step
    (256, 120, 2, 1, 2, 20, 20, 2),        # the 3x3 branch of an attention-augmented transition (partial N tile), two channel tiles
    (128, 128, 0, 1, 3, 16, 24, 3),        # plain gradient operand, non-square
    (128, 256, 2, 0, 2, 9, 11, 1),         # odd sizes: the last input row / column has no odd-column / odd-row partner
    (256, 256, 2, 1, 1, 40, 40, 4),        # ResNet layer3.0.conv2 at one image: 420 positions in four ranges
    (128, 128, 2, 1, 2, 130, 6, 5),        # tall narrow map: a padded row of 4 positions
])
def test_3x3_stride2_weight_gradient_against_torch(dev, select_w, K, N, gpro, xpro, B, H, W, splits):
    """wgrad3_kernel<.., S2>: the stride-2 3x3 layers (the first 3x3 of a ResNet stage, the 3x3 branch of an attention-augmented
    transition) -- pixels walked in the padded OUTPUT index space, the activation row staged as an even-column and an odd-column strip
    so that the taps of a kernel row are unit shifts again; against torch and against the generic kernel it replaces on these shapes."""
    from chexpert_amd import ops
    keep, ops.WGRAD_SCRATCH_FLOATS = ops.WGRAD_SCRATCH_FLOATS, 16 << 20
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    gb_, g = nhwc(50, B, Ho, Wo, N + 32, dev)
    g2b, g2 = nhwc(51, B, Ho, Wo, N, dev)
    xb, x = nhwc(52, B, H, W, K + 64, dev)
    ga, gbv, gc = rnd(53, (N,), 0.5, 1.5), rnd(54, (N,), -0.3, 0.3), rnd(55, (N,), -0.2, 0.2)
    pa, pb = rnd(56, (K,), -0.3, 1.5), rnd(57, (K,), -0.5, 0.5)
    G = bf(g[:, :N] * cv(ga) + g2 * cv(gbv) + cv(gc)) if gpro else g[:, :N]
    A = bf(F.relu(x[:, 32:32 + K] * cv(pa) + cv(pb))) if xpro else x[:, 32:32 + K]
    want = torch.nn.grad.conv2d_weight(A, (N, K, 3, 3), G, stride=2, padding=1)
    outs = []
    for on, form in ((1, 3), (0, 0)):       # wgrad3_kernel, then conv_wgrad.hip's generic kernel
        select_w(on, form)
        dw0 = rnd(58, (N, K, 3, 3), -1, 1)
        dw = dw0.clone().to(dev)
        ops.conv_wgrad(gb_[..., :N], xb[..., 32:32 + K], dw, kh=3, kw=3, stride=2, pad=1, g_prologue=gpro, g2=g2b if gpro else None,
                       ga=ga.to(dev), gb=gbv.to(dev), gc=gc.to(dev), x_prologue=xpro, pa=pa.to(dev), pb=pb.to(dev), splits=splits if on else 0)
        assert ops.last_kernel().startswith("wgrad3_kernel" if on else "wgrad_kernel"), ops.last_kernel()
        outs.append(dw.cpu() - dw0)
        close(outs[-1], want, rel=2e-3, what="dW (%s)" % ops.last_kernel())
    ops.WGRAD_SCRATCH_FLOATS = keep


# ------------------------------------------------------------------------------------------------ differential checks on random shapes
def test_random_shapes_agree_with_the_kernels_they_replace(dev, select, select_w):
    """conv_mm against the generic implicit GEMM, wgrad_mm / wgrad3 against conv_wgrad.hip's kernels, on shapes drawn at random
    (ragged pixel tiles, partial channel steps, several n tiles, strides, both prologues): same inputs, results to bf16 / fp32 rounding."""
    import random
    from chexpert_amd import ops
    rng = random.Random(1234)
    keep, ops.WGRAD_SCRATCH_FLOATS = ops.WGRAD_SCRATCH_FLOATS, 16 << 20
    try:
        for it in range(24):
            ksz = rng.choice([1, 1, 3])
            K = 8 * rng.randint(8, 40)
            N = 128 * rng.randint(1, 3)
            B, H, W = rng.randint(1, 4), rng.randint(3, 14), rng.randint(3, 14)
            stride = rng.choice([1, 1, 2]) if min(H, W) >= 4 else 1
            pro = rng.choice([ops.PRO_NONE, ops.PRO_AFFINE_RELU, ops.PRO_AFFINE2])
            xb, _ = nhwc(100 + it, B, H, W, K, dev)
            x2b, _ = nhwc(200 + it, B, H, W, K, dev)
            wp = ops.pack_weights(bf(rnd(300 + it, (N, K, ksz, ksz), -0.1, 0.1)).to(dev))
            pa, pb, pc = (rnd(400 + 3 * it + i, (K,), -1.0, 1.0).to(dev) for i in range(3))
            Ho, Wo = (H + 2 * (ksz // 2) - ksz) // stride + 1, (W + 2 * (ksz // 2) - ksz) // stride + 1
            kw = dict(N=N, kh=ksz, kw=ksz, stride=stride, pad=ksz // 2)
            if pro == ops.PRO_AFFINE_RELU:
                kw.update(prologue=pro, pa=pa, pb=pb)
            elif pro == ops.PRO_AFFINE2:
                kw.update(prologue=pro, x2=x2b, pa=pa, pb=pb, pc=pc)
            outs = []
            for on in (0, 1):
                select(on, 0)
                y = torch.empty(B, Ho, Wo, N, dtype=torch.bfloat16, device=dev)
                ops.conv_gemm(xb, wp, y, **kw)
                outs.append(y.float())
            scale = outs[0].abs().max().item() + 1e-6
            err = (outs[0] - outs[1]).abs().max().item()
            assert err <= 8e-3 * scale, ("conv", it, ksz, K, N, B, H, W, stride, pro, err / scale)
            if stride == 1 and (B * H * W) % 64 == 0 or (ksz == 3 and stride == 1 and K % 128 == 0):
                gb, _ = nhwc(500 + it, B, H, W, N, dev)
                g2b, _ = nhwc(600 + it, B, H, W, N, dev)
                ga, gbv, gc = (rnd(700 + 3 * it + i, (N,), -1.0, 1.0).to(dev) for i in range(3))
                dws = []
                for form in (0, 3):                 # conv_wgrad.hip's kernels (strip / pw / generic), then wgrad3 / wgrad_mm
                    select_w(1 if form else 0, form)
                    dw = torch.zeros(N, K, ksz, ksz, device=dev)
                    ops.conv_wgrad(gb, xb, dw, kh=ksz, kw=ksz, pad=ksz // 2, g_prologue=ops.PRO_AFFINE2, g2=g2b, ga=ga, gb=gbv, gc=gc,
                                   x_prologue=ops.PRO_AFFINE_RELU, pa=pa, pb=pb)
                    dws.append(dw)
                scale = dws[0].abs().max().item() + 1e-6
                err = (dws[0] - dws[1]).abs().max().item()
                assert err <= 2e-3 * scale, ("wgrad", it, ksz, K, N, B, H, W, err / scale)
    finally:
        ops.WGRAD_SCRATCH_FLOATS = keep


@pytest.mark.parametrize("form", [1, 3])
@pytest.mark.parametrize("B,H,W,K,N", [(2, 9, 10, 64, 256), (3, 10, 12, 128, 512), (1, 20, 20, 256, 1024), (2, 7, 7, 512, 128)])
def test_join_epilogue_equals_store_plus_relu_bwd_stats(dev, select, form, B, H, W, K, N):
    """CX_EPI_JOIN (ABI 8): the 1x1 input gradient that completes a residual join's output gradient also runs the join's backward
    (ReLU mask from the forward's sign bits + the sums of its BatchNorm's backward).  y must equal, bit for bit, what the
    accumulating CX_EPI_STORE launch followed by cx_relu_bwd_stats_mask leaves; the sums agree to fp32 summation order; against
    torch: dz = (g_identity + conv(dz1')) * [out > 0]."""
    from chexpert_amd import ops
    select(1, form)
    gb_, g = nhwc(31, B, H, W, K, dev)                 # dz1 and its second tensor (AFFINE2 prologue: BN1 backward)
    g2b, g2 = nhwc(32, B, H, W, K, dev)
    ga, gbv, gc = rnd(33, (K,), 0.5, 1.5), rnd(34, (K,), -0.5, 0.5), rnd(35, (K,), -0.2, 0.2)
    w = bf(rnd(36, (K, N, 1, 1), -0.1, 0.1))           # forward conv1: N -> K; its input gradient maps K -> N
    wt = w.permute(1, 0, 2, 3).contiguous()
    idb, idg = nhwc(37, B, H, W, N, dev)               # gradient that arrived through the identity path
    y3b, y3 = nhwc(38, B, H, W, N, dev)                # the join BatchNorm's input
    outb, out = nhwc(39, B, H, W, N, dev, -1.0, 1.0)   # join output before the ReLU (sign decides the mask)
    mu, r = rnd(40, (N,), -0.5, 0.5), rnd(41, (N,), 0.5, 2.0)
    # sign bits exactly as the forward writes them
    mask = torch.zeros(B * H * W * N // 8, dtype=torch.uint8, device=dev)
    jo = torch.empty_like(outb)
    ones, zeros = torch.ones(N, device=dev), torch.zeros(N, device=dev)
    ops.affine2_relu(outb, outb, ones, zeros, zeros, jo, mask)
    rows_cap = 64
    wp = ops.pack_weights(wt.to(dev))
    kw = dict(N=N, prologue=ops.PRO_AFFINE2, x2=g2b, pa=ga.to(dev), pb=gbv.to(dev), pc=gc.to(dev), accumulate=True)
    # reference path: store (accumulating) then the join pass
    y_ref = idb.clone()
    ops.conv_gemm(gb_, wp, y_ref, **kw)
    S = torch.zeros(3, rows_cap, N, device=dev)
    rows_r = ops.relu_bwd_stats(y_ref, jo, y3b, mu.to(dev), r.to(dev), None, None, None, y_ref, S[0], S[1], None, stat_rows=rows_cap, mask=mask)
    # fused
    y = idb.clone()
    T = torch.zeros(2, rows_cap, N, device=dev)
    rows = ops.conv_gemm(gb_, wp, y, epilogue=ops.EPI_JOIN, ex=y3b, e_mu=mu.to(dev), e_r=r.to(dev), emask=mask, stat_sum=T[0], stat_sq=T[1],
                         stat_det=True, stat_replicas=rows_cap, stat_rstride=N, **kw)
    assert rows == (B * H * W + 127) // 128
    assert torch.equal(y, y_ref), "fused join differs from store + relu_bwd_stats"
    s1, s2 = T[0, :rows].sum(0).cpu(), T[1, :rows].sum(0).cpu()
    close(s1, S[0, :rows_r].sum(0).cpu(), rel=1e-5, what="S1")
    close(s2, S[1, :rows_r].sum(0).cpu(), rel=1e-5, what="S2")
    # torch
    a = bf(g * cv(ga) + g2 * cv(gbv) + cv(gc))
    t = bf(idg + F.conv2d(a, wt))
    dz = torch.where(out > 0, t, torch.zeros(()))
    close(to_nchw(y), dz, what="dz")
    close(s1, dz.sum((0, 2, 3)), rel=2e-3, what="S1 vs torch")
    close(s2, (dz * (y3 - cv(mu)) * cv(r)).sum((0, 2, 3)), rel=2e-3, what="S2 vs torch")


def _stream_value(hi, lo):
    """fp32 value of the two-plane residual stream (common.h): bits = (hi << 16) + (int8 lo << 8)."""
    hb = hi.view(torch.int16).to(torch.int32) & 0xffff
    return ((hb << 16) + (lo.view(torch.int8).to(torch.int32) << 8)).view(torch.float32)


def _side_to_flat(plane, rows, C, per_chunk):
    """A side plane (lo: 8 bytes per 8-channel chunk; sign bits: 1) from the library's layout (include/chexpert_hip.h, cx_join_fwd:
    blocks of 64 channels where C % 64 == 0) to row-major [rows][C / 8][per_chunk]."""
    if C % 64:
        return plane.reshape(rows, C // 8, per_chunk)
    return plane.reshape(C // 64, rows, 8, per_chunk).permute(1, 0, 2, 3).reshape(rows, C // 8, per_chunk)


@pytest.mark.parametrize("rows,C,with_lo", [(50, 64, True), (333, 256, True), (1000, 1024, True), (77, 128, False), (41, 40, True)])
def test_join_fwd_two_plane_stream_against_torch(dev, rows, C, with_lo):
    """cx_join_fwd: out = relu(a*pa + (b + b_lo)*pb + pc) as hi (bf16, RNE) + lo (next 8 mantissa bits) + sign bits."""
    from chexpert_amd import ops
    a = bf(rnd(31, (1, rows, 1, C), -2.0, 2.0)).to(torch.bfloat16).to(dev)
    b32 = rnd(32, (1, rows, 1, C), 0.0, 3.0)
    b32[0, ::7] = 0.0
    pa, pb, pc = rnd(33, (C,), 0.5, 1.5).to(dev), (torch.ones(C) if with_lo else rnd(34, (C,), 0.5, 1.5)).to(dev), rnd(35, (C,), -1.0, 1.0).to(dev)
    # the identity operand as a previous join would have left it: written by the kernel itself from (0*a + 1*b32 + 0)
    b_hi = torch.empty(1, rows, 1, C, dtype=torch.bfloat16, device=dev)
    b_lo = torch.empty(rows * C, dtype=torch.int8, device=dev)
    zero, one = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    src = b32.to(torch.bfloat16).to(dev)                 # single-plane input, exact in bf16
    ops.join_fwd(a, src, None, zero, one, zero, b_hi, b_lo)
    assert torch.equal(b_hi, src) and int(b_lo.abs().max()) == 0          # a bf16 value has no lo part
    # now a genuine two-plane operand: a fp32 value v -> (hi, lo) through the kernel (t = 0*a + 1*hi(v0) + c, c = v - hi(v0))
    out = torch.empty_like(b_hi)
    out_lo = torch.empty_like(b_lo)
    mask = torch.empty(rows * C // 8, dtype=torch.uint8, device=dev)
    ops.join_fwd(a, b_hi, b_lo if with_lo else None, pa, pb, pc, out, out_lo, mask)
    t = torch.relu(a.float() * pa + src.float() * pb + pc)              # fp32, the kernel's fma order differs by <= 1 ulp(fp32)
    got = _stream_value(out.flatten(), _side_to_flat(out_lo, rows, C, 8).flatten())
    assert (out.flatten() == t.flatten().to(torch.bfloat16)).float().mean().item() > 0.999     # hi = RNE bf16 of t (fma order: rare 1-ulp ties)
    assert ((got - out.flatten().float()).abs() <= 2.0 ** -8 * out.flatten().float().abs()).all()   # lo stays within half a bf16 ulp of hi
    t64 = torch.relu(a.double() * pa.double() + src.double() * pb.double() + pc.double()).flatten()
    # 16 significant bits (half a unit = 2^-16 relative; a full unit, 2^-15, where the lo byte saturates at +127 -- one value in 512;
    # bf16 alone: 2^-9), on top of the fp32 rounding of the three-term sum itself
    err = (got.double() - t64).abs()
    assert (err <= 2.0 ** -15 * t64 + 2e-6).all() and (err <= 2.0 ** -16 * t64 + 2e-6).float().mean().item() > 0.99, (err / t64.clamp_min(1e-2)).max().item()
    mflat = _side_to_flat(mask, rows, C, 1).flatten()
    bits = ((mflat.view(-1, 1).to(torch.int32) >> torch.arange(8, device=dev).view(1, 8)) & 1).flatten().bool()
    agree = (bits == (t.flatten() > 0))
    assert agree.float().mean().item() > 0.9999                          # (a value within 1 ulp of zero may differ)
    # chained: feed (out, out_lo) back as the identity operand; the decoded operand must be what the first join stored
    out2, out2_lo = torch.empty_like(out), torch.empty_like(out_lo)
    ops.join_fwd(a, out, out_lo, zero, one, zero, out2, out2_lo)
    assert torch.equal(out2, out) and torch.equal(out2_lo, out_lo)       # 0*a + 1*(hi + lo) + 0 round-trips bit for bit
    # single-plane output form = cx_affine2_relu_mask
    o1, m1 = torch.empty_like(out), torch.empty_like(mask)
    ops.join_fwd(a, src, None, pa, pb, pc, o1, None, m1)
    o2, m2 = torch.empty_like(out), torch.empty_like(mask)
    ops.affine2_relu(a, src, pa, pb, pc, o2, m2)
    assert torch.equal(o1, o2) and torch.equal(m1, m2)


@pytest.mark.parametrize("B,H,W,K,N,with_lo,with_mask,lo_out", [
    (2, 9, 11, 256, 64, True, True, True),       # layer1's conv1: a quarter of a 256-wide tile, ragged last row block
    (3, 10, 12, 512, 128, True, True, True),
    (2, 12, 10, 1024, 256, True, False, True),   # eval mode: no sign bits
    (1, 10, 10, 2048, 512, True, True, True),    # two N tiles: only the first writes the side outputs
    (2, 8, 8, 256, 128, False, True, True),      # single-plane identity operand
    (2, 8, 8, 256, 128, False, True, False),     # single plane in and out: cx_affine2_relu_mask's bits
    (130, 20, 20, 1024, 256, True, True, True),  # 52000 rows: 512 tiles of 104 rows instead of 407 of 128 (equal rounds on 256 CUs)
    (3, 16, 16, 512, 128, True, True, False),    # two-plane operand, single-plane output (the last join of a stage that keeps lo)
])
def test_join_prologue_equals_the_standalone_join_plus_conv(dev, select, B, H, W, K, N, with_lo, with_mask, lo_out):
    """CX_PRO_JOIN (the join of the block below in the prologue of a Bottleneck's conv1, attn_aug_conv.py:188-211): the side outputs
    hi / lo / sign bits equal cx_join_fwd's bit for bit, and the convolution equals the plain 1x1 convolution of that hi plane."""
    from chexpert_amd import ops
    y3 = bf(rnd(41, (B, H, W, K), -2.0, 2.0)).to(torch.bfloat16).to(dev)
    idv = rnd(42, (B, H, W, K), 0.0, 2.5)
    pa, pc = rnd(43, (K,), 0.5, 1.5).to(dev), rnd(44, (K,), -1.5, 0.5).to(dev)
    zero, one = torch.zeros(K, device=dev), torch.ones(K, device=dev)
    id_hi = torch.empty(B, H, W, K, dtype=torch.bfloat16, device=dev)
    id_lo = torch.empty(B * H * W * K, dtype=torch.int8, device=dev)
    # a two-plane identity operand with a non-trivial lo part: relu(1 * bf16(v) + c) with fp32 c
    ops.join_fwd(idv.to(torch.bfloat16).to(dev), idv.to(torch.bfloat16).to(dev), None, zero, one, rnd(45, (K,), 0.0, 0.01).to(dev), id_hi, id_lo)
    assert int(id_lo.abs().max()) > 0
    lo_in = id_lo if with_lo else None
    want_hi, want_lo = torch.empty_like(id_hi), torch.empty_like(id_lo)
    want_mask = torch.empty(B * H * W * K // 8, dtype=torch.uint8, device=dev)
    ops.join_fwd(y3, id_hi, lo_in, pa, one, pc, want_hi, want_lo if lo_out else None, want_mask)
    w = bf(rnd(46, (N, K, 1, 1), -0.1, 0.1))
    wp = ops.pack_weights(w.to(dev))
    select(1, 3)
    want_y = torch.empty(B, H, W, N, dtype=torch.bfloat16, device=dev)
    ssum, ssq = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
    ops.conv_gemm(want_hi, wp, want_y, N=N, stat_sum=ssum, stat_sq=ssq)
    got_hi, got_lo, got_mask = torch.zeros_like(id_hi), torch.zeros_like(id_lo), torch.zeros_like(want_mask)
    got_y = torch.empty_like(want_y)
    gsum, gsq = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
    ops.conv_gemm(y3, wp, got_y, N=N, prologue=ops.PRO_JOIN, x2=id_hi, x3=lo_in, pa=pa, pb=one, pc=pc, pro_out=got_hi,
                  po_lo=got_lo if lo_out else None, po_mask=got_mask if with_mask else None, stat_sum=gsum, stat_sq=gsq)
    assert ops.lib().cx_last_kernel().decode().startswith("conv_mm_kernel<2, 4, 3, 0")
    assert torch.equal(got_hi, want_hi) and (not lo_out or torch.equal(got_lo, want_lo))
    if with_mask:
        assert torch.equal(got_mask, want_mask)
    close(to_nchw(got_y), F.conv2d(to_nchw(want_hi), w), what="conv of the joined tensor")
    if N % 256 == 0:                       # the same tile form and k order: the same bits
        assert torch.equal(got_y, want_y)
    else:
        close(to_nchw(got_y), to_nchw(want_y), rel=1e-2, what="against the plain convolution")
    close(gsum.cpu(), ssum.cpu(), rel=2e-3, what="statistics")
