"""GPU: every HIP kernel, through the C ABI, against a plain PyTorch fp32 reference of the same op.

Inputs are rounded to bf16 first (the storage type of the path) and the reference applies the same
intermediate roundings the kernel documents (A operand after the fused prologue, bf16 outputs), so the
tolerances below only cover fp32 accumulation order and the final bf16 rounding (2^-9 relative)."""
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
    _lib.lib()          # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


def bf(t):
    return t.to(torch.bfloat16).float()


def rnd(seed, shape, lo=-1.0, hi=1.0):
    return synth.uniform(seed, shape, lo, hi)


def nhwc_buf(seed, B, H, W, C, dev, lo=-1.5, hi=1.5):
    """bf16 NHWC buffer on the GPU + its fp32 NCHW value on the CPU."""
    v = bf(rnd(seed, (B, H, W, C), lo, hi))
    return v.to(torch.bfloat16).to(dev), v.permute(0, 3, 1, 2).contiguous()


def to_nchw(t):
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


def close(got, want, rel=6e-3, what=""):
    scale = want.abs().max().item() + 1e-6
    err = (got - want).abs().max().item()
    assert err <= rel * scale, "%s: max err %.3e vs scale %.3e (rel %.2e)" % (what, err, scale, err / scale)


# ------------------------------------------------------------------------------------------------ conv forward
@pytest.mark.parametrize("B,H,W,Ctot,K,N", [(2, 9, 11, 160, 96, 128), (1, 16, 16, 64, 64, 128), (3, 5, 7, 256, 224, 96),
                                             (2, 6, 6, 1024, 1024, 128), (2, 9, 11, 352, 320, 128), (1, 3, 5, 32, 32, 128)])
def test_conv1x1_bnrelu_store_stats(dev, B, H, W, Ctot, K, N):
    from chexpert_amd import ops
    xb, x = nhwc_buf(1, B, H, W, Ctot, dev)
    w = bf(rnd(2, (N, K, 1, 1), -0.2, 0.2))
    pa, pb = rnd(3, (K,), -0.3, 1.5), rnd(4, (K,), -0.5, 0.5)
    a = bf(F.relu(x[:, :K] * pa.view(1, -1, 1, 1) + pb.view(1, -1, 1, 1)))
    want = F.conv2d(a, w)
    y = torch.full((B, H, W, N + 32), 7.0, dtype=torch.bfloat16, device=dev)
    ssum, ssq = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
    ops.conv_gemm(xb[..., :K], ops.pack_weights(w.to(dev)), y[..., :N], N=N, prologue=ops.PRO_AFFINE_RELU,
                  pa=pa.to(dev), pb=pb.to(dev), stat_sum=ssum, stat_sq=ssq)
    got = to_nchw(y[..., :N])
    close(got, want, what="y")
    assert (y[..., N:].float() == 7.0).all(), "wrote outside the channel slice"
    close(ssum.cpu(), got.sum((0, 2, 3)), rel=1e-4, what="sum")
    close(ssq.cpu(), (got * got).sum((0, 2, 3)), rel=1e-4, what="sumsq")


@pytest.mark.parametrize("pro,K,B,H,W", [(1, 64, 3, 37, 41, ), (1, 128, 2, 50, 33), (1, 192, 5, 31, 29), (1, 256, 4, 40, 40), (0, 224, 2, 19, 23),
                                          (1, 32, 1, 9, 7), (1, 96, 2, 21, 17), (0, 160, 2, 16, 16)])
def test_conv1x1_forward_row_coalesced_kernel(dev, pro, K, B, H, W):
    """pw_fwd2_kernel (csrc/conv1x1_fwd2.hip: K <= 256, every global access a whole row): odd pixel counts (a partial last 64-pixel
    tile, more tiles than workgroups at the larger sizes), the operand a channel slice of a wider buffer, deterministic statistic
    rows, and the same launch twice gives the same bits."""
    from chexpert_amd import ops
    N = 128
    xb, x = nhwc_buf(95, B, H, W, K + 64, dev)
    xb, x = xb[..., 32:32 + K], x[:, 32:32 + K]
    w = bf(rnd(96, (N, K, 1, 1), -0.2, 0.2))
    pa, pb = rnd(97, (K,), -0.3, 1.5), rnd(98, (K,), -0.5, 0.5)
    a = bf(F.relu(x * pa.view(1, -1, 1, 1) + pb.view(1, -1, 1, 1))) if pro else x
    want = F.conv2d(a, w)
    kw = dict(prologue=ops.PRO_AFFINE_RELU, pa=pa.to(dev), pb=pb.to(dev)) if pro else {}
    cap = 300
    outs = []
    for _ in range(2):
        y = torch.full((B, H, W, N + 8), 7.0, dtype=torch.bfloat16, device=dev)
        st = torch.full((2, cap, N), float("nan"), device=dev)
        rows = ops.conv_gemm(xb, ops.pack_weights(w.to(dev)), y[..., :N], N=N, stat_sum=st[0], stat_sq=st[1], stat_det=True, stat_replicas=cap,
                             stat_rstride=N, **kw)
        assert ops.lib().cx_last_kernel().decode().startswith("pw_fwd2_kernel")
        assert 0 < rows <= min(256, (B * H * W + 63) // 64)   and torch.isfinite(st[:, :rows]).all() and torch.isnan(st[:, rows:]).all()
        outs.append((y.clone(), st[:, :rows].clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    y, st = outs[0]
    got = to_nchw(y[..., :N])
    close(got, want, what="y")
    assert (y[..., N:].float() == 7.0).all(), "wrote outside the channel slice"
    close(st[0].sum(0).cpu(), got.double().sum((0, 2, 3)).float(), rel=1e-4, what="sum")
    close(st[1].sum(0).cpu(), (got.double() ** 2).sum((0, 2, 3)).float(), rel=1e-4, what="sumsq")


@pytest.mark.gpu
@pytest.mark.parametrize("pro,K", [(1, 96), (0, 160), (1, 288)])
def test_conv1x1_persistent_many_tiles(dev, pro, K):
    """Bottleneck 1x1 forward (persistent kernel): more 128-pixel tiles than workgroups, a pixel count that is not a multiple
    of 128, a partial last K block, statistics spread over replicas; K = 288 restages the weights per tile."""
    from chexpert_amd import ops
    B, H, W, N, R = 6, 101, 127, 128, 4                   # 76 962 pixels = 602 tiles > 512 workgroups
    xb, x = nhwc_buf(90, B, H, W, K + 32, dev)
    w = bf(rnd(91, (N, K, 1, 1), -0.2, 0.2))
    pa, pb = rnd(92, (K,), -0.3, 1.5), rnd(93, (K,), -0.5, 0.5)
    a = bf(F.relu(x[:, :K] * pa.view(1, -1, 1, 1) + pb.view(1, -1, 1, 1))) if pro else x[:, :K]
    want = F.conv2d(a, w)
    y = torch.full((B, H, W, N + 8), 7.0, dtype=torch.bfloat16, device=dev)
    st = torch.zeros(2, R, N, device=dev)
    kw = dict(prologue=ops.PRO_AFFINE_RELU, pa=pa.to(dev), pb=pb.to(dev)) if pro else {}
    ops.conv_gemm(xb[..., :K], ops.pack_weights(w.to(dev)), y[..., :N], N=N, stat_sum=st[0], stat_sq=st[1], stat_replicas=R,
                  stat_rstride=N, **kw)
    got = to_nchw(y[..., :N])
    close(got, want, what="y")
    assert (y[..., N:].float() == 7.0).all(), "wrote outside the channel slice"
    assert (st.abs().sum(2) > 0).all(), "a replica received no statistics"
    close(st[0].sum(0).cpu(), got.double().sum((0, 2, 3)).float(), rel=1e-4, what="sum")
    close(st[1].sum(0).cpu(), (got.double() ** 2).sum((0, 2, 3)).float(), rel=1e-4, what="sumsq")


@pytest.mark.parametrize("B,H,W,K,N,stride,pro", [(2, 10, 12, 128, 32, 1, 1), (1, 7, 9, 128, 32, 1, 1), (2, 12, 12, 64, 64, 2, 0),
                                                    (1, 8, 8, 32, 128, 1, 0),
                                                    # strip kernel geometries: W=80 (R=1, ranges crossing images), W=40 (R=2, odd H),
                                                    # W=20 (R=4), W=10 (R=8 > H remainder)
                                                    (3, 10, 80, 128, 32, 1, 1), (2, 9, 40, 128, 32, 1, 1), (5, 20, 20, 128, 32, 1, 1),
                                                    (7, 10, 10, 128, 32, 1, 1),
                                                    # wide maps as two column tiles (W >= 64, even): halo columns between the tiles
                                                    (2, 13, 64, 128, 32, 1, 1), (1, 6, 96, 128, 32, 1, 1), (300, 2, 66, 128, 32, 1, 1)])
def test_conv3x3_slice_output(dev, B, H, W, K, N, stride, pro):
    from chexpert_amd import ops
    xb, x = nhwc_buf(5, B, H, W, K, dev)
    w = bf(rnd(6, (N, K, 3, 3), -0.1, 0.1))
    pa, pb = rnd(7, (K,), -0.3, 1.5), rnd(8, (K,), -0.5, 0.5)
    a = bf(F.relu(x * pa.view(1, -1, 1, 1) + pb.view(1, -1, 1, 1))) if pro else x
    want = F.conv2d(a, w, stride=stride, padding=1)
    Ho, Wo = want.shape[2:]
    buf = torch.full((B, Ho, Wo, 96 + N), -3.0, dtype=torch.bfloat16, device=dev)
    ssum, ssq = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
    ops.conv_gemm(xb, ops.pack_weights(w.to(dev)), buf[..., 96:], N=N, kh=3, kw=3, stride=stride, pad=1,
                  prologue=ops.PRO_AFFINE_RELU if pro else ops.PRO_NONE, pa=pa.to(dev) if pro else None,
                  pb=pb.to(dev) if pro else None, stat_sum=ssum, stat_sq=ssq)
    got = to_nchw(buf[..., 96:])
    close(got, want, what="y")
    assert (buf[..., :96].float() == -3.0).all()
    close(ssum.cpu(), got.sum((0, 2, 3)), rel=1e-4, what="sum")


@pytest.mark.parametrize("B,H,W", [(2, 80, 80), (3, 40, 40), (5, 20, 20), (7, 10, 10), (1, 7, 9), (9, 5, 4), (300, 2, 66), (17, 33, 66), (1, 1, 8)])
def test_conv3x3_producer_consumer_forward(dev, B, H, W):
    """conv3x3_pc_fwd_kernel (csrc/conv3x3_pc.hip; CxConv.kernel_hint form 8: the measured alternative to the ring kernel -- staging
    waves / multiplying waves, weights in registers, input channels split over the multiplying waves): the dense layer's 3x3
    (torchvision `_DenseLayer.conv2` behind norm2 + relu2, as imported at models/attn_aug_conv.py:13) against F.conv2d, with the
    deterministic statistic rows; image boundaries inside a workgroup's row range, column tiles, ranges starting inside an image."""
    from chexpert_amd import _lib, ops
    K, N = 128, 32
    xb, x = nhwc_buf(5, B, H, W, K, dev)
    w = bf(rnd(6, (N, K, 3, 3), -0.1, 0.1))
    pa, pb = rnd(7, (K,), -0.3, 1.5), rnd(8, (K,), -0.5, 0.5)
    a = bf(F.relu(x * pa.view(1, -1, 1, 1) + pb.view(1, -1, 1, 1)))
    want = F.conv2d(a, w, padding=1)
    buf = torch.full((B, H, W, 96 + N), -3.0, dtype=torch.bfloat16, device=dev)
    cap = 512
    st = torch.zeros(2, cap, N, device=dev)
    rows = ops.conv_gemm(xb, ops.pack_weights(w.to(dev)), buf[..., 96:], N=N, kh=3, kw=3, pad=1, prologue=ops.PRO_AFFINE_RELU, pa=pa.to(dev),
                         pb=pb.to(dev), stat_sum=st[0], stat_sq=st[1], stat_det=True, stat_replicas=cap, stat_rstride=N,
                         hint=ops.kernel_hint(-1, 8))
    assert _lib.lib().cx_last_kernel().decode().startswith("conv3x3_pc_fwd_kernel")
    got = to_nchw(buf[..., 96:])
    close(got, want, what="y")
    assert (buf[..., :96].float() == -3.0).all()
    close(st[0, :rows].double().sum(0).float().cpu(), got.double().sum((0, 2, 3)).float(), rel=1e-4, what="sum")
    close(st[1, :rows].double().sum(0).float().cpu(), (got.double() ** 2).sum((0, 2, 3)).float(), rel=1e-4, what="sumsq")
    assert (st[:, rows:] == 0).all(), "rows past the reported count were written"


@pytest.mark.parametrize("B,H,W,a2", [(2, 80, 80, 0), (3, 40, 40, 1), (5, 20, 20, 0), (7, 10, 10, 1), (1, 7, 9, 1), (9, 5, 4, 0), (17, 33, 66, 1), (300, 10, 10, 0)])
def test_conv3x3_producer_consumer_weight_gradient(dev, B, H, W, a2):
    """conv3x3_pc_wgrad_kernel (csrc/conv3x3_pc.hip; CxWgrad.kernel_hint form 8: the measured alternative to the ring / strip weight-
    gradient kernels): dW of the dense layer's 3x3 against torch's conv2d_weight on the bf16-rounded operands, with the dense gradient
    slice and with the deferred BatchNorm correction (AFFINE2) applied while the gradient rows are staged; reproducible slab sums."""
    from chexpert_amd import _lib, ops
    ops.set_det_wgrad(True)
    g_ = torch.Generator(device="cpu").manual_seed(11)
    y1 = bf(torch.randn(B, H, W, 128, generator=g_) * 0.7).to(torch.bfloat16).to(dev)
    gbuf = bf(torch.randn(B, H, W, 96, generator=g_) * 0.5).to(torch.bfloat16).to(dev)
    xbuf = bf(torch.randn(B, H, W, 96, generator=g_) * 0.5).to(torch.bfloat16).to(dev)
    gs, xs = gbuf[..., 64:96], xbuf[..., 32:64]
    sc, sh = (torch.rand(128, generator=g_) + 0.5).to(dev), (torch.randn(128, generator=g_) * 0.3).to(dev)
    qa, qb, qc = (torch.rand(32, generator=g_) + 0.5).to(dev), (torch.randn(32, generator=g_) * 0.2).to(dev), (torch.randn(32, generator=g_) * 0.1).to(dev)
    kw = dict(g_prologue=ops.PRO_AFFINE2, g2=xs, ga=qa, gb=qb, gc=qc) if a2 else {}
    outs = []
    for _ in range(2):
        dw = torch.zeros(32, 128, 3, 3, device=dev)
        ops.conv_wgrad(gs, y1, dw, kh=3, kw=3, pad=1, x_prologue=ops.PRO_AFFINE_RELU, pa=sc, pb=sh, hint=ops.kernel_hint(-1, 8), **kw)
        assert _lib.lib().cx_last_kernel().decode().startswith("conv3x3_pc_wgrad_kernel")
        outs.append(dw)
    assert torch.equal(outs[0], outs[1]), "the slab sums are ordered: two runs give the same bits"
    a = bf(F.relu(y1.float() * sc + sh)).permute(0, 3, 1, 2)
    gy = bf(gs.float() * qa + xs.float() * qb + qc) if a2 else gs.float()
    ref = torch.nn.grad.conv2d_weight(a, (32, 128, 3, 3), gy.permute(0, 3, 1, 2).contiguous(), padding=1)
    assert (outs[0] - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()


def test_transition_pool2_commutes_with_conv(dev):
    from chexpert_amd import ops
    B, H, W, K, N = 2, 8, 12, 64, 96
    xb, x = nhwc_buf(9, B, H, W, K, dev)
    w = bf(rnd(10, (N, K, 1, 1), -0.2, 0.2))
    pa, pb = rnd(11, (K,), -0.3, 1.5), rnd(12, (K,), -0.5, 0.5)
    act = F.relu(x * pa.view(1, -1, 1, 1) + pb.view(1, -1, 1, 1))
    # reference order of operations (attn_aug_conv.py:431-434): conv then avg-pool
    want_ref_order = F.avg_pool2d(F.conv2d(act, w), 2, 2)
    want = F.conv2d(bf(F.avg_pool2d(act, 2, 2)), w)
    y = torch.zeros(B, H // 2, W // 2, N, dtype=torch.bfloat16, device=dev)
    ops.conv_gemm(xb, ops.pack_weights(w.to(dev)), y, N=N, mode=ops.MODE_POOL2, prologue=ops.PRO_AFFINE_RELU, pa=pa.to(dev),
                  pb=pb.to(dev))
    close(to_nchw(y), want, what="pool2")
    close(to_nchw(y), want_ref_order, rel=1.5e-2, what="pool2 vs reference op order")


def test_stem_conv7x7(dev):
    from chexpert_amd import ops
    B, H, W = 2, 32, 48
    x = bf(synth.xray_batch(3, B, 64)[:, :, :H, :W].contiguous() * rnd(13, (1, 3, 1, 1), 0.5, 1.0))
    w = bf(rnd(14, (64, 3, 7, 7), -0.1, 0.1))
    want = F.conv2d(x, w, stride=2, padding=3)
    x4 = ops.nchw3_to_nhwc4(x.to(dev))
    y = torch.zeros(B, H // 2, W // 2, 64, dtype=torch.bfloat16, device=dev)
    ssum, ssq = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    ops.conv_gemm(x4, ops.pack_weights(w.to(dev), stem=True), y, N=64, mode=ops.MODE_STEM, stat_sum=ssum, stat_sq=ssq)
    got = to_nchw(y)
    close(got, want, what="stem")
    close(ssum.cpu(), got.sum((0, 2, 3)), rel=1e-4, what="sum")


# ------------------------------------------------------------------------------------------------ dgrad
@pytest.mark.parametrize("ksz,K,N,acc,B,H,W", [(1, 128, 96, False, 2, 9, 10), (1, 128, 96, True, 2, 9, 10), (3, 32, 128, False, 2, 9, 10),
                                                (1, 128, 256, True, 2, 9, 10), (1, 128, 72, True, 3, 7, 9), (1, 128, 8, False, 1, 5, 5),
                                                # strip dgrad geometries
                                                (3, 32, 128, False, 3, 10, 80), (3, 32, 128, False, 2, 9, 40), (3, 32, 128, False, 5, 20, 20)])
def test_dgrad_affine2_mask_epilogue(dev, ksz, K, N, acc, B, H, W):
    from chexpert_amd import ops
    ub, u = nhwc_buf(20, B, H, W, K, dev)
    vb, v = nhwc_buf(21, B, H, W, K, dev)
    exb, ex = nhwc_buf(22, B, H, W, N + 32, dev)
    oldb, old = nhwc_buf(23, B, H, W, N + 32, dev)
    w = bf(rnd(24, (K, N, ksz, ksz), -0.1, 0.1))          # forward conv weight: (O=K, I=N): dgrad maps K -> N
    pa, pb, pc = rnd(25, (K,), 0.5, 1.5), rnd(26, (K,), -0.3, 0.3), rnd(27, (K,), -0.2, 0.2)
    e_sc, e_sh = rnd(28, (N,), -0.3, 1.5), rnd(29, (N,), -0.5, 0.5)
    e_mu, e_r, e_scale = rnd(30, (N,), -0.5, 0.5), rnd(31, (N,), 0.5, 2.0), rnd(32, (N,), -0.3, 1.5)
    cv = lambda t: t.view(1, -1, 1, 1)
    dy = bf(u * cv(pa) + v * cv(pb) + cv(pc))
    acc_ref = F.conv_transpose2d(dy, w, padding=ksz // 2)            # = input gradient of the forward conv
    exs = ex[:, :N]
    mask = (exs * cv(e_sc) + cv(e_sh)) > 0
    dz = torch.where(mask, acc_ref, torch.zeros(()))
    want = cv(e_scale) * dz + (old[:, :N] if acc else 0)
    S1 = dz.sum((0, 2, 3))
    S2 = (dz * (exs - cv(e_mu)) * cv(e_r)).sum((0, 2, 3))
    s1, s2 = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
    wp = ops.pack_weights(w.to(dev), transpose=True)
    ops.conv_gemm(ub, wp, oldb[..., :N], N=N, kh=ksz, kw=ksz, pad=ksz // 2, prologue=ops.PRO_AFFINE2, x2=vb, pa=pa.to(dev),
                  pb=pb.to(dev), pc=pc.to(dev), epilogue=ops.EPI_MASK, ex=exb[..., :N], e_sc=e_sc.to(dev), e_sh=e_sh.to(dev),
                  e_mu=e_mu.to(dev), e_r=e_r.to(dev), e_scale=e_scale.to(dev), stat_sum=s1, stat_sq=s2, accumulate=acc)
    close(to_nchw(oldb[..., :N]), want, rel=8e-3, what="g")
    assert torch.equal(to_nchw(oldb[..., N:]), old[:, N:]), "wrote outside the slice"
    close(s1.cpu(), S1, rel=2e-3, what="S1")
    close(s2.cpu(), S2, rel=2e-3, what="S2")


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,ring", [(3, 10, 80, True), (2, 9, 40, True), (5, 20, 20, True), (2, 10, 10, True), (2, 3, 3, False)])
def test_dgrad_3x3_dense_side_output_of_the_prologue(dev, B, H, W, ring):
    """CxConv.pro_out (ABI 7): the 3x3 input-gradient kernel of a dense layer also stores its prologue result -- the gradient slice
    after the deferred BatchNorm correction -- as a dense (M, 32) tensor, bit-equal to the bf16 rounding of the fp32 expression,
    and the weight gradient computed from it (g_prologue NONE) equals the one computed from the two strided slices (AFFINE2).
    Kernels without the side output leave the tensor alone and say so (last_pro_out)."""
    from chexpert_amd import ops
    K, N = 32, 128
    ub, u = nhwc_buf(40, B, H, W, K + 96, dev)            # slices of wider buffers, as in a dense block
    vb, v = nhwc_buf(41, B, H, W, K + 96, dev)
    exb, ex = nhwc_buf(42, B, H, W, N, dev)
    w = bf(rnd(43, (K, N, 3, 3), -0.1, 0.1))
    pa, pb, pc = rnd(44, (K,), 0.5, 1.5), rnd(45, (K,), -0.3, 0.3), rnd(46, (K,), -0.2, 0.2)
    ones, zeros = torch.ones(N, device=dev), torch.zeros(N, device=dev)
    out = torch.zeros(B, H, W, N, dtype=torch.bfloat16, device=dev)
    side = torch.full((B, H, W, K), 7.0, dtype=torch.bfloat16, device=dev)
    s1, s2 = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
    us, vs = ub[..., 64:64 + K], vb[..., 64:64 + K]
    ops.conv_gemm(us, ops.pack_weights(w.to(dev), transpose=True), out, N=N, kh=3, kw=3, pad=1, prologue=ops.PRO_AFFINE2, x2=vs,
                  pa=pa.to(dev), pb=pb.to(dev), pc=pc.to(dev), epilogue=ops.EPI_MASK, ex=exb, e_sc=ones, e_sh=ones * 100, e_mu=zeros,
                  e_r=ones, e_scale=ones, stat_sum=s1, stat_sq=s2, pro_out=side)
    assert ops.last_pro_out() == ring
    if not ring:
        assert (side == 7.0).all().item()
        return
    cv = lambda t: t.view(1, 1, 1, -1).to(dev)
    want = torch.addcmul(torch.addcmul(cv(pc), vs.float(), cv(pb)), us.float(), cv(pa)).to(torch.bfloat16)   # fmaf(u, a, fmaf(v, b, c))
    assert (side.float() - want.float()).abs().max().item() <= 2.0 ** -7 * want.float().abs().max().item()    # (one bf16 ulp: fma contraction)
    assert (side == want).float().mean().item() > 0.99
    # the weight gradient from the dense tensor against the one from the strided pair
    y1b, _ = nhwc_buf(47, B, H, W, N, dev)
    pa2, pb2 = rnd(48, (N,), 0.5, 1.5).to(dev), rnd(49, (N,), -0.3, 0.3).to(dev)
    dw_a, dw_b = torch.zeros(K, N, 3, 3, device=dev), torch.zeros(K, N, 3, 3, device=dev)
    ops.conv_wgrad(us, y1b, dw_a, kh=3, kw=3, pad=1, g_prologue=ops.PRO_AFFINE2, g2=vs, ga=pa.to(dev), gb=pb.to(dev), gc=pc.to(dev),
                   x_prologue=ops.PRO_AFFINE_RELU, pa=pa2, pb=pb2)
    ops.conv_wgrad(side, y1b, dw_b, kh=3, kw=3, pad=1, x_prologue=ops.PRO_AFFINE_RELU, pa=pa2, pb=pb2)
    close(dw_b.cpu(), dw_a.cpu(), rel=2e-3, what="dW from the dense slice")


@pytest.mark.gpu
@pytest.mark.parametrize("pro,acc", [(2, True), (0, False)])
def test_dgrad_1x1_many_tiles_replicated_stats(dev, pro, acc):
    """The dZ-resident 1x1 input-gradient kernel with several N tiles per workgroup (grid large enough for 4), a partial
    last tile, a pixel count that is not a multiple of 128 and statistics spread over 4 replicas."""
    from chexpert_amd import ops
    B, H, W, K, N, R = 8, 127, 129, 128, 264, 4
    ub, u = nhwc_buf(70, B, H, W, K, dev)
    vb, v = nhwc_buf(71, B, H, W, K, dev)
    exb, ex = nhwc_buf(72, B, H, W, N, dev)
    oldb, old = nhwc_buf(73, B, H, W, N, dev)
    w = bf(rnd(74, (K, N, 1, 1), -0.1, 0.1))
    pa, pb, pc = rnd(75, (K,), 0.5, 1.5), rnd(76, (K,), -0.3, 0.3), rnd(77, (K,), -0.2, 0.2)
    e_sc, e_sh = rnd(78, (N,), -0.3, 1.5), rnd(79, (N,), -0.5, 0.5)
    e_mu, e_r, e_scale = rnd(80, (N,), -0.5, 0.5), rnd(81, (N,), 0.5, 2.0), rnd(82, (N,), -0.3, 1.5)
    cv = lambda t: t.view(1, -1, 1, 1)
    dy = bf(u * cv(pa) + v * cv(pb) + cv(pc)) if pro == 2 else u
    acc_ref = F.conv_transpose2d(dy, w)
    mask = (ex * cv(e_sc) + cv(e_sh)) > 0
    dz = torch.where(mask, acc_ref, torch.zeros(()))
    want = cv(e_scale) * dz + (old if acc else 0)
    S1 = dz.double().sum((0, 2, 3)).float()
    S2 = (dz * (ex - cv(e_mu)) * cv(e_r)).double().sum((0, 2, 3)).float()
    st = torch.zeros(2, R, N + 8, device=dev)
    kw = dict(prologue=ops.PRO_AFFINE2, x2=vb, pa=pa.to(dev), pb=pb.to(dev), pc=pc.to(dev)) if pro == 2 else {}
    ops.conv_gemm(ub, ops.pack_weights(w.to(dev), transpose=True), oldb, N=N, epilogue=ops.EPI_MASK, ex=exb, e_sc=e_sc.to(dev),
                  e_sh=e_sh.to(dev), e_mu=e_mu.to(dev), e_r=e_r.to(dev), e_scale=e_scale.to(dev), stat_sum=st[0], stat_sq=st[1],
                  accumulate=acc, stat_replicas=R, stat_rstride=N + 8, **kw)
    close(to_nchw(oldb), want, rel=8e-3, what="g")
    assert (st[:, :, :N].abs().sum(2) > 0).all(), "a replica received no statistics"
    assert float(st[:, :, N:].abs().sum()) == 0.0
    close(st[0].sum(0)[:N].cpu(), S1, rel=2e-3, what="S1")
    close(st[1].sum(0)[:N].cpu(), S2, rel=2e-3, what="S2")


# ------------------------------------------------------------------------------------------------ wgrad
@pytest.mark.parametrize("ksz,K,N,gpro,xpro,B,H,W", [(1, 96, 128, 2, 1, 3, 9, 7), (3, 128, 32, 2, 1, 3, 9, 7), (1, 64, 64, 0, 0, 3, 9, 7),
                                                      (3, 64, 64, 0, 1, 3, 9, 7), (1, 224, 128, 2, 1, 3, 9, 7),
                                                      # strip wgrad geometries
                                                      (3, 128, 32, 2, 1, 3, 10, 80), (3, 128, 32, 0, 1, 2, 9, 40),
                                                      (3, 128, 32, 2, 1, 5, 20, 20), (3, 128, 32, 2, 1, 9, 10, 10), (3, 128, 32, 2, 1, 2, 33, 40),
                                                      # the same kernel on other channel counts (ResNet 3x3: K x N tile pairs)
                                                      (3, 256, 96, 2, 1, 2, 20, 20), (3, 64, 128, 2, 1, 2, 33, 40), (3, 96, 64, 0, 1, 1, 80, 80),
                                                      # output channels not a multiple of 32 (AAConv conv branch: planes - dv)
                                                      (3, 256, 232, 2, 1, 2, 20, 20), (3, 64, 40, 0, 1, 2, 9, 40),
                                                      # ring wgrad geometries (W >= 56): R = 3 at W = 80, partial last step, no g2
                                                      (3, 128, 32, 2, 1, 2, 80, 80), (3, 128, 32, 0, 1, 3, 57, 64)])
def test_wgrad(dev, ksz, K, N, gpro, xpro, B, H, W):
    from chexpert_amd import ops
    gb_, g = nhwc_buf(40, B, H, W, N + 32, dev)
    g2b, g2 = nhwc_buf(41, B, H, W, N, dev)
    xb, x = nhwc_buf(42, B, H, W, K + 64, dev)
    ga, gbv, gc = rnd(43, (N,), 0.5, 1.5), rnd(44, (N,), -0.3, 0.3), rnd(45, (N,), -0.2, 0.2)
    pa, pb = rnd(46, (K,), -0.3, 1.5), rnd(47, (K,), -0.5, 0.5)
    cv = lambda t: t.view(1, -1, 1, 1)
    G = bf(g[:, :N] * cv(ga) + g2 * cv(gbv) + cv(gc)) if gpro else g[:, :N]
    A = bf(F.relu(x[:, :K] * cv(pa) + cv(pb))) if xpro else x[:, :K]
    want = torch.nn.grad.conv2d_weight(A, (N, K, ksz, ksz), G, padding=ksz // 2)
    dw0 = rnd(48, (N, K, ksz, ksz), -1, 1)
    dw = dw0.clone().to(dev)
    ops.conv_wgrad(gb_[..., :N], xb[..., :K], dw, kh=ksz, kw=ksz, pad=ksz // 2, g_prologue=gpro, g2=g2b if gpro else None,
                   ga=ga.to(dev), gb=gbv.to(dev), gc=gc.to(dev), x_prologue=xpro, pa=pa.to(dev), pb=pb.to(dev))
    close(dw.cpu() - dw0, want, rel=2e-3, what="dW")


def test_wgrad_pool2_and_stem(dev):
    from chexpert_amd import ops
    B, H, W, K, N = 2, 8, 12, 64, 128
    xb, x = nhwc_buf(50, B, H, W, K, dev)
    gb_, g = nhwc_buf(51, B, H // 2, W // 2, N, dev)
    pa, pb = rnd(52, (K,), -0.3, 1.5), rnd(53, (K,), -0.5, 0.5)
    cv = lambda t: t.view(1, -1, 1, 1)
    A = bf(F.avg_pool2d(F.relu(x * cv(pa) + cv(pb)), 2, 2))
    want = torch.nn.grad.conv2d_weight(A, (N, K, 1, 1), g)
    dw = torch.zeros(N, K, 1, 1, device=dev)
    ops.conv_wgrad(gb_, xb, dw, mode=ops.MODE_POOL2, x_prologue=ops.PRO_AFFINE_RELU, pa=pa.to(dev), pb=pb.to(dev))
    close(dw.cpu(), want, rel=2e-3, what="dW pool2")
    # stem
    Hs, Ws = 24, 32
    xs = bf(rnd(54, (B, 3, Hs, Ws), -2, 2))
    gsb, gs = nhwc_buf(55, B, Hs // 2, Ws // 2, 64, dev)
    want = torch.nn.grad.conv2d_weight(xs, (64, 3, 7, 7), gs, stride=2, padding=3)
    dw = torch.zeros(64, 3, 7, 7, device=dev)
    ops.conv_wgrad(gsb, ops.nchw3_to_nhwc4(xs.to(dev)), dw, mode=ops.MODE_STEM)
    close(dw.cpu(), want, rel=2e-3, what="dW stem")


# ------------------------------------------------------------------------------------------------ elementwise
def test_pack_weights_layouts(dev):
    from chexpert_amd import ops
    w = bf(rnd(60, (6, 5, 3, 3)))
    p0 = ops.pack_weights(w.to(dev)).float().cpu().view(9, 6, 5)
    assert torch.equal(p0, w.permute(2, 3, 0, 1).reshape(9, 6, 5))
    p1 = ops.pack_weights(w.to(dev), transpose=True).float().cpu().view(9, 5, 6)
    assert torch.equal(p1, w.flip(2, 3).permute(2, 3, 1, 0).reshape(9, 5, 6))


def test_bn_coef_and_running_stats(dev):
    from chexpert_amd import ops
    Cn, cnt = 96, 1000.0
    x = rnd(61, (1000, Cn), -2, 3)
    s, q = x.sum(0).to(dev), (x * x).sum(0).to(dev)
    gamma, beta = rnd(62, (Cn,), 0.5, 1.5), rnd(63, (Cn,), -0.5, 0.5)
    rm, rv = rnd(64, (Cn,)), rnd(65, (Cn,), 0.5, 1.5)
    rm_d, rv_d = rm.clone().to(dev), rv.clone().to(dev)
    out = [torch.zeros(Cn, device=dev) for _ in range(4)]
    ops.bn_coef(s, q, cnt, gamma.to(dev), beta.to(dev), 1e-5, 0.1, rm_d, rv_d, *out)
    mean, var = x.mean(0), x.var(0, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    close(out[0].cpu(), gamma * rstd, rel=1e-5)
    close(out[1].cpu(), beta - mean * gamma * rstd, rel=1e-5)
    close(out[2].cpu(), mean, rel=1e-5)
    close(out[3].cpu(), rstd, rel=1e-5)
    close(rm_d.cpu(), 0.9 * rm + 0.1 * mean, rel=1e-5)
    close(rv_d.cpu(), 0.9 * rv + 0.1 * x.var(0, unbiased=True), rel=1e-5)


def test_stem_maxpool_fwd_bwd(dev):
    from chexpert_amd import ops
    B, H, W, Cn = 2, 12, 16, 64
    xb, x = nhwc_buf(70, B, H, W, Cn, dev)
    sc, sh = rnd(71, (Cn,), -0.3, 1.5), rnd(72, (Cn,), -0.5, 0.5)
    cv = lambda t: t.view(1, -1, 1, 1)
    xr = x.clone().requires_grad_(True)
    act = F.relu(xr * cv(sc) + cv(sh))
    pooled = F.max_pool2d(act, 3, 2, 1)
    y = torch.zeros(B, H // 2, W // 2, Cn + 32, dtype=torch.bfloat16, device=dev)
    amax = torch.zeros(B, H // 2, W // 2, Cn, dtype=torch.uint8, device=dev)
    ssum, ssq = torch.zeros(Cn, device=dev), torch.zeros(Cn, device=dev)
    ops.bnrelu_maxpool_fwd(xb, sc.to(dev), sh.to(dev), y[..., :Cn], amax, ssum, ssq)
    got = to_nchw(y[..., :Cn])
    close(got, pooled.detach(), rel=4e-3, what="pooled")
    close(ssum.cpu(), got.sum((0, 2, 3)), rel=1e-4)
    # backward: dY = g*ga + gx*gb + gc routed to the arg-max, masked by the ReLU
    gb_, g = nhwc_buf(73, B, H // 2, W // 2, Cn, dev)
    gxb, gx = nhwc_buf(74, B, H // 2, W // 2, Cn, dev)
    ga, gbv, gc = rnd(75, (Cn,), 0.5, 1.5), rnd(76, (Cn,), -0.3, 0.3), rnd(77, (Cn,), -0.2, 0.2)
    mu, r = rnd(78, (Cn,), -0.5, 0.5), rnd(79, (Cn,), 0.5, 2.0)
    dY = g * cv(ga) + gx * cv(gbv) + cv(gc)
    pooled.backward(dY)                      # -> gradient wrt x; divide the BN scale back out to get dz
    # d act/d x = sc on the active set, so dz (grad wrt the BN output) = xr.grad / sc where sc != 0
    dz_ref = torch.where(cv(sc).abs() > 1e-12, xr.grad / cv(sc), torch.zeros(()))
    dz = torch.zeros(B, H, W, Cn, dtype=torch.bfloat16, device=dev)
    S1, S2 = torch.zeros(Cn, device=dev), torch.zeros(Cn, device=dev)
    ops.bnrelu_maxpool_bwd(xb, sc.to(dev), sh.to(dev), mu.to(dev), r.to(dev), amax, gb_, gxb, ga.to(dev), gbv.to(dev),
                           gc.to(dev), dz, S1, S2)
    close(to_nchw(dz), dz_ref, rel=6e-3, what="dz")
    close(S1.cpu(), dz_ref.sum((0, 2, 3)), rel=4e-3, what="S1")
    close(S2.cpu(), (dz_ref * (x - cv(mu)) * cv(r)).sum((0, 2, 3)), rel=4e-3, what="S2")


def test_head_loss_and_backward(dev):
    from chexpert_amd import ops
    B, H, W, Cn, n = 3, 5, 5, 128, 5
    xb, x = nhwc_buf(80, B, H, W, Cn + 64, dev)
    sc, sh = rnd(81, (Cn,), -0.3, 1.5), rnd(82, (Cn,), -0.5, 0.5)
    mu, r, es = rnd(83, (Cn,), -0.5, 0.5), rnd(84, (Cn,), 0.5, 2.0), rnd(85, (Cn,), -0.3, 1.5)
    wl, bl = rnd(86, (n, Cn), -0.2, 0.2).requires_grad_(True), rnd(87, (n,), -0.1, 0.1).requires_grad_(True)
    tgt = synth.targets(88, B, n)
    cv = lambda t: t.view(1, -1, 1, 1)
    xs = x[:, :Cn]
    z = (xs * cv(sc) + cv(sh)).requires_grad_(True)
    pooled_ref = F.relu(z).mean((2, 3))
    logits_ref = F.linear(pooled_ref, wl, bl)
    le = F.binary_cross_entropy_with_logits(logits_ref, tgt, reduction="none")
    loss_ref = le.sum(1).mean(0)
    loss_ref.backward()
    pooled, logits = torch.zeros(B, Cn, device=dev), torch.zeros(B, n, device=dev)
    ops.head_fwd(xb[..., :Cn], sc.to(dev), sh.to(dev), wl.detach().to(dev), bl.detach().to(dev), pooled, logits)
    close(logits.cpu(), logits_ref.detach(), rel=1e-5, what="logits")
    loss, lel, dl = torch.zeros(1, device=dev), torch.zeros(B, n, device=dev), torch.zeros(B, n, device=dev)
    ops.bce_fwd_bwd(logits, tgt.to(dev), loss, lel, dl)
    assert abs(loss.item() - loss_ref.item()) < 1e-5
    close(lel.cpu(), le.detach(), rel=1e-5)
    dw, db, dp = torch.zeros(n, Cn, device=dev), torch.zeros(n, device=dev), torch.zeros(B, Cn, device=dev)
    ops.head_bwd(dl, pooled, wl.detach().to(dev), dw, db, dp)
    close(dw.cpu(), wl.grad, rel=1e-4, what="dW")
    close(db.cpu(), bl.grad, rel=1e-4, what="db")
    g = torch.full((B, H, W, Cn + 32), 5.0, dtype=torch.bfloat16, device=dev)
    S1, S2 = torch.zeros(Cn, device=dev), torch.zeros(Cn, device=dev)
    ops.gap_relu_bn_bwd(dp, xb[..., :Cn], sc.to(dev), sh.to(dev), mu.to(dev), r.to(dev), es.to(dev), g[..., :Cn], S1, S2)
    dz_ref = z.grad
    close(to_nchw(g[..., :Cn]), cv(es) * dz_ref, rel=6e-3, what="g")
    close(S1.cpu(), dz_ref.sum((0, 2, 3)), rel=1e-4)
    close(S2.cpu(), (dz_ref * (xs - cv(mu)) * cv(r)).sum((0, 2, 3)), rel=1e-4)


@pytest.mark.parametrize("B,n", [(5, 10), (64, 100), (300, 7), (2, 130)])
def test_softmax_cross_entropy_matches_torch(dev, B, n):
    """cx_softmax_ce_fwd_bwd against F.cross_entropy (the CIFAR harness criterion, models/test_model.py:331): loss, per-sample terms,
    gradient; the module form goes through autograd.  fp32: 1e-5."""
    from chexpert_amd import ops
    from chexpert_amd.cifar import CrossEntropyLoss
    logits = (rnd(90 + B, (B, n), -6.0, 6.0)).requires_grad_(True)
    tgt = torch.randint(0, n, (B,), generator=torch.Generator().manual_seed(n))
    le = F.cross_entropy(logits, tgt, reduction="none")
    le.mean().backward()
    loss, lel, dl = torch.zeros(1, device=dev), torch.zeros(B, device=dev), torch.zeros(B, n, device=dev)
    ops.softmax_ce_fwd_bwd(logits.detach().to(dev), tgt.to(dev), loss, lel, dl, 3.0)
    assert abs(loss.item() - le.mean().item()) < 1e-5 * max(1.0, le.mean().item())
    close(lel.cpu(), le.detach(), rel=1e-5, what="per-sample loss")
    close(dl.cpu(), 3.0 * logits.grad, rel=1e-5, what="dlogits")
    lg = logits.detach().to(dev).requires_grad_(True)
    out = CrossEntropyLoss()(lg, tgt.to(dev))
    (2.0 * out).backward()
    assert abs(out.item() - le.mean().item()) < 1e-5 * max(1.0, le.mean().item())
    close(lg.grad.cpu(), 2.0 * logits.grad, rel=1e-5, what="autograd gradient")


def test_unpool_mask_and_affine2(dev):
    from chexpert_amd import ops
    B, H, W, Cn = 2, 8, 6, 256
    db_, d = nhwc_buf(90, B, H // 2, W // 2, Cn, dev)
    xb, x = nhwc_buf(91, B, H, W, Cn, dev)
    sc, sh = rnd(92, (Cn,), -0.3, 1.5), rnd(93, (Cn,), -0.5, 0.5)
    mu, r, es = rnd(94, (Cn,), -0.5, 0.5), rnd(95, (Cn,), 0.5, 2.0), rnd(96, (Cn,), -0.3, 1.5)
    cv = lambda t: t.view(1, -1, 1, 1)
    up = F.interpolate(d, scale_factor=2, mode="nearest") * 0.25
    dz = torch.where((x * cv(sc) + cv(sh)) > 0, up, torch.zeros(()))
    g = torch.zeros(B, H, W, Cn, dtype=torch.bfloat16, device=dev)
    S1, S2 = torch.zeros(Cn, device=dev), torch.zeros(Cn, device=dev)
    ops.unpool2_mask(db_, xb, sc.to(dev), sh.to(dev), mu.to(dev), r.to(dev), es.to(dev), g, S1, S2)
    close(to_nchw(g), cv(es) * dz, rel=6e-3)
    close(S1.cpu(), dz.sum((0, 2, 3)), rel=1e-4)
    close(S2.cpu(), (dz * (x - cv(mu)) * cv(r)).sum((0, 2, 3)), rel=1e-4)
    pa, pb, pc = rnd(97, (Cn,), 0.5, 1.5), rnd(98, (Cn,), -0.3, 0.3), rnd(99, (Cn,), -0.2, 0.2)
    ub, u = nhwc_buf(100, B, H, W, Cn, dev)
    ops.affine2_inplace(ub, xb, pa.to(dev), pb.to(dev), pc.to(dev))
    close(to_nchw(ub), u * cv(pa) + x * cv(pb) + cv(pc), rel=6e-3)


def test_bn_backward_coefficients(dev):
    from chexpert_amd import ops
    Cn, cnt = 64, 500.0
    S1, S2 = rnd(101, (Cn,), -3, 3), rnd(102, (Cn,), -3, 3)
    gamma, mu, r = rnd(103, (Cn,), -0.3, 1.5), rnd(104, (Cn,), -0.5, 0.5), rnd(105, (Cn,), 0.5, 2.0)
    dg0, db0, A0, B0 = rnd(106, (Cn,)), rnd(107, (Cn,)), rnd(108, (Cn,)), rnd(109, (Cn,))
    t = lambda v: v.clone().to(dev)
    dg, db, A, Bc = t(dg0), t(db0), t(A0), t(B0)
    pa, pb, pc = (torch.zeros(Cn, device=dev) for _ in range(3))
    ops.bn_bwd_coef(t(S1), t(S2), cnt, t(gamma), t(mu), t(r), dg, db, A, Bc, pa, pb, pc, Cn)
    close(dg.cpu(), dg0 + S2, rel=1e-6)
    close(db.cpu(), db0 + S1, rel=1e-6)
    close(A.cpu(), A0 + r * gamma * S1 / cnt, rel=1e-6)
    close(Bc.cpu(), B0 + r * gamma * S2 / cnt, rel=1e-6)
    # single-consumer form: dY = gamma*r*(dz - S1/n - xhat*S2/n), xhat = (y-mu)*r
    dz, y = rnd(110, (7, Cn)), rnd(111, (7, Cn))
    want = gamma * r * (dz - S1 / cnt - (y - mu) * r * S2 / cnt)
    close(dz * pa.cpu() + y * pb.cpu() + pc.cpu(), want, rel=1e-5)
    qa, qb, qc = (torch.zeros(Cn, device=dev) for _ in range(3))
    ops.bn_bwd_slice_coef(A, Bc, t(mu), t(r), qa, qb, qc, Cn)
    G, x = rnd(112, (7, Cn)), rnd(113, (7, Cn))
    want = G - A.cpu() - (x - mu) * r * Bc.cpu()
    close(G * qa.cpu() + x * qb.cpu() + qc.cpu(), want, rel=1e-5)
    # the same slice vectors emitted by cx_bn_bwd_coef itself for a sub-range (fused form: one launch per consumer)
    A2, B2 = t(A0), t(B0)
    dg2, db2 = t(dg0), t(db0)
    q2 = [torch.full((24,), 9.0, device=dev) for _ in range(3)]
    ops.bn_bwd_coef(t(S1), t(S2), cnt, t(gamma), t(mu), t(r), dg2, db2, A2, B2, None, None, None, Cn, q=(q2[0], q2[1], q2[2], 8, 24))
    assert torch.equal(A2, A) and torch.equal(B2, Bc)
    for a, b in zip(q2, (qa, qb, qc)):
        assert torch.equal(a, b[8:32])


@pytest.mark.parametrize("kind", ["adam", "sgd_nesterov", "rmsprop"])
def test_fused_optimisers_match_torch_optim(dev, kind):
    """chexpert.py:470 / :479 / :499 wiring, three steps, against torch.optim on the CPU."""
    from chexpert_amd import ops
    from oracle import step as ostep
    n = 10007
    p0 = rnd(120, (n,))
    pc = p0.clone().requires_grad_(True)
    opt, _ = ostep.make_optimizer(kind, [pc], 1e-2)
    p = p0.clone().to(dev)
    st = [torch.zeros(n, device=dev) for _ in range(2)]
    for it in range(3):
        g = rnd(121 + it, (n,))
        pc.grad = g.clone()
        opt.step()
        gd = g.to(dev)
        if kind == "adam":
            ops.adam_step(p, gd, st[0], st[1], 1e-2, 0.9, 0.999, 1e-8, 0.0, it + 1)
        elif kind == "sgd_nesterov":
            ops.sgd_nesterov_step(p, gd, st[0], 1e-2, 0.9, 0.0, it == 0)
        else:
            ops.rmsprop_step(p, gd, st[0], st[1], 1e-2, 0.99, 1e-3, 0.9, 0.0)
    close(p.cpu(), pc.detach(), rel=2e-6, what=kind)


@pytest.mark.parametrize("kind", ["adam", "sgd_nesterov", "rmsprop"])
def test_device_hyper_optimisers_and_schedulers_match_torch(dev, kind):
    """cx_*_step_dev + cx_optim_tick (learning rate / step count in device memory, for hipGraph replay) against torch.optim with
    the reference's scheduler wiring: `scheduler.step()` once per minibatch from `lr_warmup_steps` on (chexpert.py:165),
    MultiStepLR (:480) for SGD, ExponentialLR (:500) for RMSprop, none for Adam."""
    from chexpert_amd import ops
    from oracle import step as ostep
    n, warm, lr0 = 4099, 2, 1e-2
    p0 = rnd(140, (n,))
    pc = p0.clone().requires_grad_(True)
    opt, _ = ostep.make_optimizer(kind, [pc], lr0)
    sched = None
    kindc, gamma, ms = 0, 1.0, (0, 0)
    if kind == "sgd_nesterov":
        sched, kindc, gamma, ms = torch.optim.lr_scheduler.MultiStepLR(opt, [2, 4], 0.1), 2, 0.1, (2, 4)
    if kind == "rmsprop":
        sched, kindc, gamma = torch.optim.lr_scheduler.ExponentialLR(opt, 0.97), 1, 0.97
    hyper = torch.tensor([lr0, 0, kindc, gamma, warm, ms[0], ms[1], lr0], dtype=torch.float32, device=dev)
    p = p0.clone().to(dev)
    st = [torch.zeros(n, device=dev) for _ in range(2)]
    for it in range(8):
        g = rnd(141 + it, (n,))
        pc.grad = g.clone()
        opt.step()
        if sched is not None and it + 1 >= warm:          # chexpert.py:157, :165: the 1-based step count is compared
            sched.step()
        gd = g.to(dev)
        if kind == "adam":
            ops.adam_step_dev(p, gd, st[0], st[1], hyper, 0.9, 0.999, 1e-8, 0.0)
        elif kind == "sgd_nesterov":
            ops.sgd_nesterov_step_dev(p, gd, st[0], hyper, 0.9, 0.0)
        else:
            ops.rmsprop_step_dev(p, gd, st[0], st[1], hyper, 0.99, 1e-3, 0.9, 0.0)
        ops.optim_tick(hyper)
        h = hyper.cpu()
        assert int(h[1]) == it + 1
        assert abs(float(h[0]) - opt.param_groups[0]["lr"]) <= 1e-6 * lr0, (it, float(h[0]), opt.param_groups[0]["lr"])
    close(p.cpu(), pc.detach(), rel=5e-6, what=kind)


@pytest.mark.gpu
@pytest.mark.parametrize("gpro,xpro,K,N", [(2, 1, 264, 128), (0, 1, 320, 128), (2, 0, 264, 128), (2, 1, 264, 384)])
def test_wgrad_1x1_bottleneck_large(dev, gpro, xpro, K, N):
    """The 512-thread bottleneck weight-gradient kernel (taken when pixels x channels >= 2^25): partial last channel tile,
    pixel count not a multiple of the 64-pixel step, with / without the two-tensor dY and the BN-ReLU input prologue."""
    from chexpert_amd import ops
    B, H, W = 8, 127, 129
    gb_, g = nhwc_buf(100, B, H, W, N, dev)
    g2b, g2 = nhwc_buf(101, B, H, W, N, dev)
    xb, x = nhwc_buf(102, B, H, W, K + 8, dev)
    ga, gbv, gc = rnd(103, (N,), 0.5, 1.5), rnd(104, (N,), -0.5, 0.5), rnd(105, (N,), -0.2, 0.2)
    pa, pb = rnd(106, (K,), -0.3, 1.5), rnd(107, (K,), -0.5, 0.5)
    cv = lambda t: t.view(1, -1, 1, 1)
    dy = bf(g * cv(ga) + g2 * cv(gbv) + cv(gc)) if gpro else g
    a = bf(F.relu(x[:, :K] * cv(pa) + cv(pb))) if xpro else x[:, :K]
    want = torch.nn.grad.conv2d_weight(a, (N, K, 1, 1), dy)
    dw = torch.zeros(N, K, 1, 1, device=dev)
    kw = {}
    if gpro:
        kw.update(g_prologue=ops.PRO_AFFINE2, g2=g2b, ga=ga.to(dev), gb=gbv.to(dev), gc=gc.to(dev))
    if xpro:
        kw.update(x_prologue=ops.PRO_AFFINE_RELU, pa=pa.to(dev), pb=pb.to(dev))
    ops.conv_wgrad(gb_, xb[..., :K], dw, **kw)
    close(dw.cpu(), want, rel=2e-3, what="dW 1x1 large")


@pytest.mark.gpu
@pytest.mark.parametrize("Ws", [64, 128])
def test_wgrad_stem_row_strips(dev, Ws):
    """Stem weight gradient on maps whose output width is a multiple of the 32-pixel step: the input rows of a step are staged as
    strips and read at an overlapping 16-B pitch (stem_wgrad_kernel<., true>); image borders on every side, two tensors."""
    from chexpert_amd import ops, _lib
    B, Hs = 2, 12
    xs = bf(rnd(164, (B, 3, Hs, Ws), -2, 2))
    ub, u = nhwc_buf(165, B, Hs // 2, Ws // 2, 64, dev)
    vb, v = nhwc_buf(166, B, Hs // 2, Ws // 2, 64, dev)
    ga, gb_, gc = rnd(167, (64,), 0.5, 1.5), rnd(168, (64,), -0.5, 0.5), rnd(169, (64,), -0.2, 0.2)
    cv = lambda t: t.view(1, -1, 1, 1)
    gs = bf(u * cv(ga) + v * cv(gb_) + cv(gc))
    want = torch.nn.grad.conv2d_weight(xs, (64, 3, 7, 7), gs, stride=2, padding=3)
    dw = torch.zeros(64, 3, 7, 7, device=dev)
    ops.conv_wgrad(ub, ops.nchw3_to_nhwc4(xs.to(dev)), dw, mode=ops.MODE_STEM, g_prologue=ops.PRO_AFFINE2, g2=vb, ga=ga.to(dev),
                   gb=gb_.to(dev), gc=gc.to(dev), splits=3)
    assert "true" in _lib.lib().cx_last_kernel().decode(), _lib.lib().cx_last_kernel().decode()
    close(dw.cpu(), want, rel=2e-3, what="dW stem strips")
    want1 = torch.nn.grad.conv2d_weight(xs, (64, 3, 7, 7), u, stride=2, padding=3)
    dw1 = torch.zeros(64, 3, 7, 7, device=dev)
    ops.conv_wgrad(ub, ops.nchw3_to_nhwc4(xs.to(dev)), dw1, mode=ops.MODE_STEM)
    close(dw1.cpu(), want1, rel=2e-3, what="dW stem strips, one tensor")


@pytest.mark.gpu
def test_wgrad_stem_affine2_odd_pixel_count(dev):
    """Stem weight gradient with the two-tensor BN-backward form of dY (conv0 under norm0) and a pixel count that is not a
    multiple of the 32-pixel step (partial last step, splits > 1)."""
    from chexpert_amd import ops
    B, Hs, Ws = 3, 20, 28                                  # 3*10*14 = 420 output pixels
    xs = bf(rnd(64, (B, 3, Hs, Ws), -2, 2))
    ub, u = nhwc_buf(65, B, Hs // 2, Ws // 2, 64, dev)
    vb, v = nhwc_buf(66, B, Hs // 2, Ws // 2, 64, dev)
    ga, gb_, gc = rnd(67, (64,), 0.5, 1.5), rnd(68, (64,), -0.5, 0.5), rnd(69, (64,), -0.2, 0.2)
    cv = lambda t: t.view(1, -1, 1, 1)
    gs = bf(u * cv(ga) + v * cv(gb_) + cv(gc))
    want = torch.nn.grad.conv2d_weight(xs, (64, 3, 7, 7), gs, stride=2, padding=3)
    dw = torch.zeros(64, 3, 7, 7, device=dev)
    ops.conv_wgrad(ub, ops.nchw3_to_nhwc4(xs.to(dev)), dw, mode=ops.MODE_STEM, g_prologue=ops.PRO_AFFINE2, g2=vb, ga=ga.to(dev),
                   gb=gb_.to(dev), gc=gc.to(dev), splits=5)
    close(dw.cpu(), want, rel=2e-3, what="dW stem affine2")


@pytest.mark.gpu
@pytest.mark.parametrize("pro,acc,B,H,W,N", [(2, True, 2, 9, 10, 96), (0, False, 1, 5, 5, 8), (2, True, 8, 127, 129, 264),
                                              (0, False, 2, 20, 20, 128), (2, False, 3, 13, 7, 136), (0, True, 4, 40, 40, 512),
                                              (2, True, 1, 3, 3, 160), (2, True, 3, 11, 12, 64), (2, False, 2, 10, 10, 72),
                                              (0, True, 2, 8, 9, 56)])
def test_fused_1x1_dgrad_wgrad_equals_separate_kernels(dev, pro, acc, B, H, W, N):
    """cx_conv1x1_dgrad_wgrad = input gradient with the mask epilogue (cx_conv_gemm) + weight gradient with the BN-ReLU input
    prologue (cx_conv_wgrad) of the bottleneck 1x1 convolution, against a torch reference of both."""
    from chexpert_amd import ops
    K = 128
    ub, u = nhwc_buf(120, B, H, W, K, dev)
    vb, v = nhwc_buf(121, B, H, W, K, dev)
    exb, ex = nhwc_buf(122, B, H, W, N + 8, dev)
    oldb, old = nhwc_buf(123, B, H, W, N + 8, dev)
    w = bf(rnd(124, (K, N, 1, 1), -0.1, 0.1))                          # forward weight (O = K = 128, I = N)
    pa, pb, pc = rnd(125, (K,), 0.5, 1.5), rnd(126, (K,), -0.3, 0.3), rnd(127, (K,), -0.2, 0.2)
    e_sc, e_sh = rnd(128, (N,), -0.3, 1.5), rnd(129, (N,), -0.5, 0.5)
    e_mu, e_r, e_scale = rnd(130, (N,), -0.5, 0.5), rnd(131, (N,), 0.5, 2.0), rnd(132, (N,), -0.3, 1.5)
    cv = lambda t: t.view(1, -1, 1, 1)
    dy = bf(u * cv(pa) + v * cv(pb) + cv(pc)) if pro == 2 else u
    acc_ref = F.conv_transpose2d(dy, w)
    exs = ex[:, :N]
    pre = exs * cv(e_sc) + cv(e_sh)
    dz = torch.where(pre > 0, acc_ref, torch.zeros(()))
    want_g = cv(e_scale) * dz + (old[:, :N] if acc else 0)
    S1 = dz.double().sum((0, 2, 3)).float()
    S2 = (dz * (exs - cv(e_mu)) * cv(e_r)).double().sum((0, 2, 3)).float()
    want_dw = torch.nn.grad.conv2d_weight(bf(F.relu(pre)), (K, N, 1, 1), dy)
    R = 4
    st = torch.zeros(2, R, N, device=dev)
    dw = torch.zeros(K, N, 1, 1, device=dev)
    kw = dict(prologue=ops.PRO_AFFINE2, x2=vb, pa=pa.to(dev), pb=pb.to(dev), pc=pc.to(dev)) if pro == 2 else {}
    ops.conv_gemm(ub, ops.pack_weights(w.to(dev), transpose=True), oldb[..., :N], N=N, epilogue=ops.EPI_MASK, ex=exb[..., :N],
                  e_sc=e_sc.to(dev), e_sh=e_sh.to(dev), e_mu=e_mu.to(dev), e_r=e_r.to(dev), e_scale=e_scale.to(dev), stat_sum=st[0],
                  stat_sq=st[1], accumulate=acc, stat_replicas=R, stat_rstride=N, fused_dw=dw, **kw)
    close(to_nchw(oldb[..., :N]), want_g, rel=8e-3, what="g")
    assert torch.equal(to_nchw(oldb[..., N:]), old[:, N:]), "wrote outside the slice"
    close(st[0].sum(0).cpu(), S1, rel=2e-3, what="S1")
    close(st[1].sum(0).cpu(), S2, rel=3e-3, what="S2")
    close(dw.cpu(), want_dw, rel=3e-3, what="dW")


@pytest.mark.gpu
@pytest.mark.parametrize("acc,det,B,H,W,N", [(True, False, 2, 9, 10, 96), (True, True, 8, 40, 40, 256), (False, False, 3, 13, 7, 136),
                                             (True, True, 4, 20, 20, 512), (True, False, 1, 3, 3, 160), (True, True, 16, 40, 40, 160)])
def test_fused_1x1_backward_of_two_layers_equals_two_passes(dev, acc, det, B, H, W, N):
    """cx_conv1x1_dgrad_wgrad_pair_ws (two dense layers per pass over the N channels both read) against the single-layer kernel run
    twice: the gradient buffer bit for bit (the tile carries bf16 between the layers, as the two passes do), weight gradients and
    statistic rows to fp32 summation order; layer a's weight gradient lands in a column range of a wider matrix."""
    from chexpert_amd import ops
    ops.set_det_wgrad(det)
    K, NA = 128, N + 32
    exb, _ = nhwc_buf(500, B, H, W, N + 40, dev)
    oldb, old = nhwc_buf(501, B, H, W, N + 40, dev)
    lay = []
    for i in range(2):
        n_in = NA if i == 0 else N
        ub, _ = nhwc_buf(510 + i, B, H, W, K, dev)
        vb, _ = nhwc_buf(520 + i, B, H, W, K, dev)
        w = bf(rnd(530 + i, (K, n_in, 1, 1), -0.1, 0.1)).to(dev)
        wp = ops.pack_weights(w, transpose=True)[:N * K]                      # rows = input channels: the first N of them
        vec = lambda s, lo, hi: rnd(s + i, (n_in,), lo, hi).to(dev)[:N].contiguous()
        kw = dict(N=N, epilogue=ops.EPI_MASK, ex=exb[..., :N], e_sc=vec(540, -0.3, 1.5), e_sh=vec(550, -0.5, 0.5), e_mu=vec(560, -0.5, 0.5),
                  e_r=vec(570, 0.5, 2.0), e_scale=vec(580, -0.3, 1.5), accumulate=acc, prologue=ops.PRO_AFFINE2, x2=vb,
                  pa=rnd(590 + i, (K,), 0.5, 1.5).to(dev), pb=rnd(600 + i, (K,), -0.3, 0.3).to(dev), pc=rnd(610 + i, (K,), -0.2, 0.2).to(dev))
        lay.append((ub, wp, kw, n_in))
    R = 256 if det else 4

    def run(pair):
        g = oldb.clone()
        sts = [torch.zeros(2, R, N, device=dev) for _ in range(2)]
        dws = [torch.zeros(K, lay[i][3], device=dev) for i in range(2)]
        args = []
        for i in range(2):
            kw = dict(lay[i][2], stat_sum=sts[i][0], stat_sq=sts[i][1], stat_replicas=R, stat_rstride=N, stat_det=det)
            args.append((lay[i][0], lay[i][1], g[..., :N], kw))
        if pair:
            rows = ops.conv1x1_bwd_pair(args[0], args[1], dws[0][:, :N], dws[1])
            assert "pw_bwd2p" in _kernel_name()
        else:
            for i in range(2):
                rows = ops.conv_gemm(args[i][0], args[i][1], args[i][2], fused_dw=dws[i][:, :N], **dict(args[i][3], accumulate=acc or i == 1))
        torch.cuda.synchronize()
        return g, [s_.sum(1) for s_ in sts], dws
    g1, st1, dw1 = run(False)
    g2, st2, dw2 = run(True)
    assert torch.equal(g1, g2), "gradient buffer differs from the two single-layer passes"
    assert torch.equal(g2[..., N:], oldb[..., N:]), "wrote outside the shared channels"
    assert not torch.equal(g2[..., :N], oldb[..., :N])
    for i in range(2):
        close(dw2[i].cpu(), dw1[i].cpu(), rel=1e-5, what="dW layer %d" % i)
        assert float(dw2[0][:, N:].abs().max()) == 0, "wrote outside the column range"
        close(st2[i][0].cpu(), st1[i][0].cpu(), rel=1e-5, what="S1 layer %d" % i)
        close(st2[i][1].cpu(), st1[i][1].cpu(), rel=1e-4, what="S2 layer %d" % i)
    if det:
        g3, st3, dw3 = run(True)
        assert all(torch.equal(a_, b_) for a_, b_ in zip(dw2 + st2, dw3 + st3)), "not deterministic"
    ops.set_det_wgrad(False)


def test_u8_brightness_contrast_jitter_matches_torch_restatement(dev):
    """cx_u8_jitter against a CPU restatement of torchvision's tensor ColorJitter(brightness, contrast) on uint8 (the reference's
    only augmentation code: explore_data.ipynb cell 6, ColorJitter(0.25, 0.25))."""
    from chexpert_amd import ops
    B, S = 5, 96
    u8 = synth.xray_u8(55, B, S)
    bfac = rnd(56, (B,), 0.75, 1.25)
    cfac = rnd(57, (B,), 0.75, 1.25)
    order = torch.tensor([0, 1, 0, 1, 1], dtype=torch.int32)
    want = torch.empty_like(u8)
    for i in range(B):
        img = u8[i].float()

        def bright(t):
            return (t * bfac[i]).clamp(0, 255).to(torch.uint8).float()

        def contrast(t):
            return (cfac[i] * t + (1 - cfac[i]) * t.mean()).clamp(0, 255).to(torch.uint8).float()
        img = contrast(bright(img)) if order[i] == 0 else bright(contrast(img))
        want[i] = img.to(torch.uint8)
    got = ops.u8_jitter(u8.to(dev), bfac.to(dev), cfac.to(dev), order.to(dev)).cpu()
    diff = (got.int() - want.int()).abs()
    # the image mean is an fp32 sum in a different order: a product within 1e-5 of an integer may truncate the other way
    assert diff.max().item() <= 1 and (diff > 0).float().mean().item() < 1e-3, (diff.max().item(), (diff > 0).float().mean().item())
    assert got.float().std() > 10


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,n", [(4, 10, 10, 3), (3, 20, 20, 5), (2, 40, 40, 2)])
def test_conv3x3_wgrad_batch_equals_per_layer(dev, B, H, W, n):
    """cx_conv3x3_wgrad_batch (ABI 8): the 3x3 weight gradients of several dense layers in one launch against torch per layer and
    against cx_conv_wgrad per layer (other pixel-range splits: equal to fp32 summation order); dw is added to."""
    from chexpert_amd import ops
    ops.set_det_wgrad(True)           # the batch stores partial tiles in the slab workspace (no atomic form)
    items, wants, dws0 = [], [], []
    for i in range(n):
        gb_, g = nhwc_buf(300 + i, B, H, W, 32, dev)
        buf = bf(rnd(320 + i, (B, H, W, 160), -1.5, 1.5)).to(torch.bfloat16).to(dev)      # the bottleneck tensor as a slice of a wider buffer
        xb = buf[..., 16:144]
        x = xb.float().cpu().permute(0, 3, 1, 2).contiguous()
        pa, pb = rnd(340 + i, (128,), -0.3, 1.5), rnd(360 + i, (128,), -0.5, 0.5)
        cv = lambda t: t.view(1, -1, 1, 1)
        a = bf(F.relu(x * cv(pa) + cv(pb)))
        wants.append(torch.nn.grad.conv2d_weight(a, (32, 128, 3, 3), g, padding=1))
        dw0 = rnd(380 + i, (32, 128, 3, 3), -1, 1)
        dws0.append(dw0)
        items.append((gb_, xb, pa.to(dev), pb.to(dev), dw0.clone().to(dev)))
    assert ops.conv3x3_wgrad_batch(items), "the library declined a dense-layer shape"
    assert "batch" in _kernel_name()
    for i in range(n):
        close(items[i][4].cpu() - dws0[i], wants[i], rel=2e-3, what="dW layer %d" % i)
        ref = dws0[i].clone().to(dev)
        ops.conv_wgrad(items[i][0], items[i][1], ref, kh=3, kw=3, pad=1, x_prologue=ops.PRO_AFFINE_RELU, pa=items[i][2], pb=items[i][3])
        close(items[i][4].cpu() - dws0[i], ref.cpu() - dws0[i], rel=1e-5, what="batch vs per layer %d" % i)
    ops.set_det_wgrad(False)


def _kernel_name():
    from chexpert_amd import _lib
    return _lib.lib().cx_last_kernel().decode()


@pytest.mark.parametrize("nbytes", [16, 16 * 1023, 16 * 1024, 16 * (4096 * 3 + 17), 1 << 26])
def test_copy_stream_copies_every_byte(dev, nbytes):
    """cx_copy_stream (bench.py's measured stream rate): whole 16 KB pieces and the tail."""
    from chexpert_amd import ops
    src = torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device=dev)
    dst = torch.zeros_like(src)
    ops.copy_stream(src, dst)
    assert torch.equal(src, dst)
    with pytest.raises(RuntimeError):
        ops.copy_stream(src, dst[:nbytes - 16] if nbytes > 16 else torch.zeros(32, dtype=torch.uint8, device=dev))
