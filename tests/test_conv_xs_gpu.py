"""GPU: the activation-stationary 1x1 kernel (csrc/conv1x1_xs.hip: K = 64 | 128 | 256 input channels, N >= 2 K output channels in
chunks of 128 -- the ResNet bottleneck expansion, attn_aug_conv.py:159-211) through cx_conv_gemm against a PyTorch fp32 reference of the
same op and against the tiled kernel (conv_mm.hip) it replaces on these shapes."""
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

    def sel(on, form):
        _ops.KERNEL_HINT = _ops.kernel_hint(on, form)
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

CASES = [
    # B, H, W, K, N, prologue (0 none, 1 BN + ReLU, 2 two-tensor BN-backward form), accumulate
    (2, 9, 11, 64, 256, 0, False),        # one k-step per chunk, two chunks, ragged tiles (198 rows: 64 + 64 + 64 + 6)
    (3, 10, 12, 128, 512, 1, False),      # two k-steps, four chunks
    (2, 20, 20, 256, 1024, 1, False),     # the layer3 shape: four k-steps, eight chunks
    (1, 7, 9, 256, 512, 2, True),         # two-tensor operand, accumulating into the output (63 rows: less than one tile)
    (4, 10, 10, 128, 256, 2, False),
    (9, 20, 20, 256, 1024, 0, True),      # 3600 rows: 57 tiles of 64
    (2, 16, 24, 64, 384, 1, True),        # three chunks (N % 256 != 0)
]


@pytest.mark.parametrize("B,H,W,K,N,pro,acc", CASES)
def test_forward_against_torch(dev, select, B, H, W, K, N, pro, acc):
    from chexpert_amd import ops
    select(1, 5)                                     # form 5: this kernel or an error, never the tiled one
    xb, x = nhwc(5, B, H, W, K + 64, dev)            # the operand is a channel slice of a wider buffer
    xb, x = xb[..., 32:32 + K], x[:, 32:32 + K]
    vb, v = nhwc(9, B, H, W, K, dev)
    w = bf(rnd(6, (N, K, 1, 1), -0.1, 0.1))
    pa, pb, pc = rnd(7, (K,), -0.3, 1.5), rnd(8, (K,), -0.5, 0.5), rnd(10, (K,), -0.2, 0.2)
    if pro == 1:
        a = bf(F.relu(x * cv(pa) + cv(pb)))
        kw = dict(prologue=ops.PRO_AFFINE_RELU, pa=pa.to(dev), pb=pb.to(dev))
    elif pro == 2:
        a = bf(x * cv(pa) + v * cv(pb) + cv(pc))
        kw = dict(prologue=ops.PRO_AFFINE2, x2=vb, pa=pa.to(dev), pb=pb.to(dev), pc=pc.to(dev))
    else:
        a, kw = x, {}
    oldb, old = nhwc(11, B, H, W, 96 + N, dev)
    want = F.conv2d(a, w) + (old[:, 96:] if acc else 0)
    buf = oldb.clone()
    rows_cap = 128
    ssum, ssq = torch.zeros(rows_cap, N, device=dev), torch.zeros(rows_cap, N, device=dev)
    rows = ops.conv_gemm(xb, ops.pack_weights(w.to(dev)), buf[..., 96:], N=N, stat_sum=ssum, stat_sq=ssq, stat_det=True,
                         stat_replicas=rows_cap, stat_rstride=N, accumulate=acc, **kw)
    assert ops.last_kernel().startswith("pw_xs_kernel"), ops.last_kernel()
    M = B * H * W
    assert 1 <= rows <= rows_cap and rows >= (M + 127) // 128
    got = to_nchw(buf[..., 96:])
    close(got, want, what="y")
    assert torch.equal(buf[..., :96], oldb[..., :96]), "wrote outside the slice"
    close(ssum[:rows].sum(0).cpu(), got.sum((0, 2, 3)), rel=1e-4, what="sum")
    close(ssq[:rows].sum(0).cpu(), (got * got).sum((0, 2, 3)), rel=1e-4, what="sum of squares")
    assert (ssum[rows:] == 0).all()
    # the same call on the tiled kernel
    select(1, 1)
    buf2 = oldb.clone()
    ops.conv_gemm(xb, ops.pack_weights(w.to(dev)), buf2[..., 96:], N=N, accumulate=acc, **kw)
    assert ops.last_kernel().startswith("conv_mm_kernel"), ops.last_kernel()
    assert torch.equal(buf2[..., 96:], buf[..., 96:]), "same MFMA shape, same order of the k groups: the two kernels agree bit for bit"


def test_bit_reproducible_and_default_dispatch(dev):
    from chexpert_amd import ops
    B, H, W, K, N = 16, 20, 20, 256, 1024
    xb, _ = nhwc(40, B, H, W, K, dev)
    wp = ops.pack_weights(bf(rnd(42, (N, K, 1, 1), -0.1, 0.1)).to(dev))
    pa, pb = rnd(43, (K,), -1.0, 1.0).to(dev), rnd(44, (K,), -1.0, 1.0).to(dev)
    outs = []
    for _ in range(2):
        y = torch.empty(B, H, W, N, dtype=torch.bfloat16, device=dev)
        s1, s2 = torch.zeros(128, N, device=dev), torch.zeros(128, N, device=dev)
        ops.conv_gemm(xb, wp, y, N=N, prologue=ops.PRO_AFFINE_RELU, pa=pa, pb=pb, stat_sum=s1, stat_sq=s2, stat_det=True, stat_replicas=128,
                      stat_rstride=N)
        assert ops.last_kernel().startswith("pw_xs_kernel"), "the bottleneck expansion takes this kernel by default: " + ops.last_kernel()
        outs.append((y.clone(), s1.clone(), s2.clone()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("B,H,W,K,N", [(2, 9, 10, 64, 256), (3, 10, 12, 128, 512), (1, 20, 20, 256, 1024), (5, 20, 20, 256, 1024)])
def test_join_epilogue_equals_store_plus_relu_bwd_stats(dev, select, B, H, W, K, N):
    """CX_EPI_JOIN on this kernel (the conv1 input gradient of a bottleneck: dz1 K -> N = 4 K, completing the residual join's output
    gradient and running the join's backward): y equals, bit for bit, what the accumulating store on the tiled kernel followed by
    cx_relu_bwd_stats_mask leaves (same MFMA shape, same order of the k groups); the sums agree to fp32 summation order."""
    from chexpert_amd import ops
    gb_, g = nhwc(31, B, H, W, K, dev)                 # dz1 and its second tensor (AFFINE2 prologue: BN1 backward)
    g2b, g2 = nhwc(32, B, H, W, K, dev)
    ga, gbv, gc = rnd(33, (K,), 0.5, 1.5), rnd(34, (K,), -0.5, 0.5), rnd(35, (K,), -0.2, 0.2)
    w = bf(rnd(36, (K, N, 1, 1), -0.1, 0.1))           # forward conv1: N -> K; its input gradient maps K -> N
    wt = w.permute(1, 0, 2, 3).contiguous()
    idb, idg = nhwc(37, B, H, W, N, dev)               # gradient that arrived through the identity path
    y3b, y3 = nhwc(38, B, H, W, N, dev)                # the join BatchNorm's input
    outb, out = nhwc(39, B, H, W, N, dev, -1.0, 1.0)   # join output before the ReLU (sign decides the mask)
    mu, r = rnd(40, (N,), -0.5, 0.5), rnd(41, (N,), 0.5, 2.0)
    mask = torch.zeros(B * H * W * N // 8, dtype=torch.uint8, device=dev)
    jo = torch.empty_like(outb)
    ones, zeros = torch.ones(N, device=dev), torch.zeros(N, device=dev)
    ops.affine2_relu(outb, outb, ones, zeros, zeros, jo, mask)
    rows_cap = 128
    wp = ops.pack_weights(wt.to(dev))
    kw = dict(N=N, prologue=ops.PRO_AFFINE2, x2=g2b, pa=ga.to(dev), pb=gbv.to(dev), pc=gc.to(dev), accumulate=True)
    select(1, 1)                                       # reference path on the tiled kernel: store (accumulating) then the join pass
    y_ref = idb.clone()
    ops.conv_gemm(gb_, wp, y_ref, **kw)
    assert ops.last_kernel().startswith("conv_mm_kernel")
    S = torch.zeros(3, rows_cap, N, device=dev)
    rows_r = ops.relu_bwd_stats(y_ref, jo, y3b, mu.to(dev), r.to(dev), None, None, None, y_ref, S[0], S[1], None, stat_rows=rows_cap, mask=mask)
    select(1, 5)
    y = idb.clone()
    T = torch.zeros(2, rows_cap, N, device=dev)
    rows = ops.conv_gemm(gb_, wp, y, epilogue=ops.EPI_JOIN, ex=y3b, e_mu=mu.to(dev), e_r=r.to(dev), emask=mask, stat_sum=T[0], stat_sq=T[1],
                         stat_det=True, stat_replicas=rows_cap, stat_rstride=N, **kw)
    assert ops.last_kernel().startswith("pw_xs_kernel"), ops.last_kernel()
    assert torch.equal(y, y_ref), "fused join differs from store + relu_bwd_stats"
    s1, s2 = T[0, :rows].sum(0).cpu(), T[1, :rows].sum(0).cpu()
    close(s1, S[0, :rows_r].sum(0).cpu(), rel=1e-5, what="S1")
    close(s2, S[1, :rows_r].sum(0).cpu(), rel=1e-4, what="S2")
    a = bf(g * cv(ga) + g2 * cv(gbv) + cv(gc))
    t = bf(idg + F.conv2d(a, wt))
    dz = torch.where(out > 0, t, torch.zeros(()))
    close(to_nchw(y), dz, what="dz")
    close(s1, dz.sum((0, 2, 3)), rel=2e-3, what="S1 vs torch")
    close(s2, (dz * (y3 - cv(mu)) * cv(r)).sum((0, 2, 3)), rel=2e-3, what="S2 vs torch")
