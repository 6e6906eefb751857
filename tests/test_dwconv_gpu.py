"""GPU: the depthwise-convolution entry points (cx_dwconv_fwd / _dgrad / _wgrad, dwconv.hip tiled kernels and the effnet.hip
fallback) against a torch fp32 restatement of the MBConv depthwise stage (/root/reference/models/efficientnet.py:53-64, 93-95):
BatchNorm+Swish applied on load and rounded to bf16, bf16 outputs, fp32 statistics and weight gradients."""
import pytest
import torch
import torch.nn.functional as F

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


def rnd(seed, shape, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g) * (hi - lo) + lo


def nhwc(t, dev):                      # NCHW fp32 (bf16-representable) -> NHWC bf16 on the device
    return t.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev)


def nchw(t):
    return t.float().cpu().permute(0, 3, 1, 2)


def close(got, want, rel, what):
    err = (got - want).abs().max().item() / (want.abs().max().item() + 1e-12)
    assert err < rel, "%s: rel err %.3e" % (what, err)


def swish(z):
    return z * torch.sigmoid(z)


def dswish(z):
    s = torch.sigmoid(z)
    return s * (1 + z * (1 - s))


CASES = [  # k, stride, B, H, W, C, act   (tile edges, odd sizes, channel counts that do not fill a 32/64-channel block)
    (3, 1, 2, 20, 37, 40, True), (3, 2, 3, 21, 40, 96, True), (5, 1, 2, 17, 18, 144, True), (5, 2, 2, 23, 33, 40, True),
    (3, 1, 5, 10, 10, 72, False), (5, 2, 4, 9, 11, 128, True), (5, 1, 3, 7, 7, 64, True), (3, 2, 2, 40, 40, 32, False),
    (3, 3, 2, 12, 12, 16, True),       # stride 3: not tiled, the effnet.hip kernels
]


@pytest.mark.parametrize("k,s,B,H,W,C,act", CASES)
def test_dwconv_forward(dev, k, s, B, H, W, C, act):
    from chexpert_amd._lib import lib, ptr, check, stream_ptr
    pad = k // 2
    x = bf(rnd(1, (B, C, H, W), -2, 2))
    w = rnd(2, (C, 1, k, k), -0.5, 0.5)
    sc, sh = rnd(3, (C,), 0.5, 1.5), rnd(4, (C,), -0.5, 0.5)
    cv = lambda t: t.view(1, -1, 1, 1)
    a = bf(swish(x * cv(sc) + cv(sh))) if act else x
    want = F.conv2d(a, w, stride=s, padding=pad, groups=C)
    Ho, Wo = want.shape[2:]
    xb = nhwc(x, dev)
    y = torch.full((B, Ho, Wo, C), 7.0, device=dev, dtype=torch.bfloat16)
    st = torch.zeros(2, C, device=dev)
    wd, scd, shd = w.to(dev), sc.to(dev), sh.to(dev)
    check(lib().cx_dwconv_fwd(ptr(xb), ptr(wd), ptr(scd) if act else None, ptr(shd) if act else None, ptr(y), ptr(st[0]), ptr(st[1]),
                              B, H, W, C, k, s, pad, 0, stream_ptr()), "cx_dwconv_fwd")
    got = nchw(y)
    close(got, want, 6e-3, "y")
    close(st[0].cpu(), got.double().sum((0, 2, 3)).float(), 2e-3, "sum")
    close(st[1].cpu(), (got.double() ** 2).sum((0, 2, 3)).float(), 2e-3, "sumsq")
    # deterministic statistic rows: the rows add up to the same sums, and two launches give the same bits
    outs = []
    for _ in range(2):
        rows = torch.full((2, 512, C), 7.0, device=dev)
        check(lib().cx_dwconv_fwd(ptr(xb), ptr(wd), ptr(scd) if act else None, ptr(shd) if act else None, ptr(y), ptr(rows[0]), ptr(rows[1]),
                                  B, H, W, C, k, s, pad, 512, stream_ptr()), "cx_dwconv_fwd")
        n = lib().cx_last_stat_rows()
        assert 0 < n <= 512
        outs.append(rows[:, :n].clone())
    assert torch.equal(outs[0], outs[1])
    close(outs[0][0].sum(0).cpu(), st[0].cpu(), 1e-4, "row sums")
    close(outs[0][1].sum(0).cpu(), st[1].cpu(), 1e-4, "row sums of squares")


@pytest.mark.parametrize("k,s,B,H,W,C,act", CASES)
@pytest.mark.parametrize("accumulate", [0, 1])
def test_dwconv_input_gradient(dev, k, s, B, H, W, C, act, accumulate):
    from chexpert_amd._lib import lib, ptr, check, stream_ptr
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    x = bf(rnd(11, (B, C, H, W), -2, 2))
    g, g2 = bf(rnd(12, (B, C, Ho, Wo))), bf(rnd(13, (B, C, Ho, Wo)))
    ga, gb, gc = rnd(14, (C,), 0.5, 1.5), rnd(15, (C,), -0.3, 0.3), rnd(16, (C,), -0.2, 0.2)
    w = rnd(17, (C, 1, k, k), -0.5, 0.5)
    sc, sh, mu, r = rnd(18, (C,), 0.5, 1.5), rnd(19, (C,), -0.5, 0.5), rnd(20, (C,), -0.5, 0.5), rnd(21, (C,), 0.5, 2.0)
    old = bf(rnd(22, (B, C, H, W)))
    cv = lambda t: t.view(1, -1, 1, 1)
    dY = bf(g * cv(ga) + g2 * cv(gb) + cv(gc))
    da = torch.nn.grad.conv2d_input((B, C, H, W), w, dY, stride=s, padding=pad, groups=C)
    dz = da * dswish(x * cv(sc) + cv(sh)) if act else da
    want = dz + (old if accumulate else 0)
    S1 = dz.double().sum((0, 2, 3)).float()
    S2 = (dz * (x - cv(mu)) * cv(r)).double().sum((0, 2, 3)).float()
    out = nhwc(old, dev) if accumulate else torch.full((B, H, W, C), 7.0, device=dev, dtype=torch.bfloat16)
    st = torch.zeros(2, C, device=dev)
    t = [v.to(dev) for v in (ga, gb, gc, w, sc, sh, mu, r)]
    gq, g2q, xq = nhwc(g, dev), nhwc(g2, dev), nhwc(x, dev)
    opt = lambda v: ptr(v) if act else None
    check(lib().cx_dwconv_dgrad(ptr(gq), ptr(g2q), ptr(t[0]), ptr(t[1]), ptr(t[2]), ptr(t[3]), ptr(xq), opt(t[4]), opt(t[5]), opt(t[6]),
                                opt(t[7]), ptr(out), ptr(st[0]), ptr(st[1]), B, H, W, C, k, s, pad, accumulate, 0, stream_ptr()),
          "cx_dwconv_dgrad")
    close(nchw(out), want, 8e-3, "dz")
    close(st[0].cpu(), S1, 2e-3, "S1")
    if act:
        close(st[1].cpu(), S2, 3e-3, "S2")
        outs = []                            # deterministic statistic rows (the engine asks for them where a BatchNorm precedes)
        for _ in range(2):
            out2 = nhwc(old, dev) if accumulate else torch.full((B, H, W, C), 7.0, device=dev, dtype=torch.bfloat16)
            rows = torch.full((2, 512, C), 7.0, device=dev)
            check(lib().cx_dwconv_dgrad(ptr(gq), ptr(g2q), ptr(t[0]), ptr(t[1]), ptr(t[2]), ptr(t[3]), ptr(xq), opt(t[4]), opt(t[5]), opt(t[6]),
                                        opt(t[7]), ptr(out2), ptr(rows[0]), ptr(rows[1]), B, H, W, C, k, s, pad, accumulate, 512, stream_ptr()),
                  "cx_dwconv_dgrad")
            n = lib().cx_last_stat_rows()
            outs.append(rows[:, :n].clone())
        assert torch.equal(outs[0], outs[1]) and torch.equal(out2, out)
        close(outs[0][0].sum(0).cpu(), st[0].cpu(), 1e-4, "S1 rows")
        close(outs[0][1].sum(0).cpu(), st[1].cpu(), 1e-4, "S2 rows")


@pytest.mark.parametrize("k,s,B,H,W,C,act", CASES)
def test_dwconv_weight_gradient(dev, k, s, B, H, W, C, act):
    from chexpert_amd._lib import lib, ptr, check, stream_ptr
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    x = bf(rnd(31, (B, C, H, W), -2, 2))
    g, g2 = bf(rnd(32, (B, C, Ho, Wo))), bf(rnd(33, (B, C, Ho, Wo)))
    ga, gb, gc = rnd(34, (C,), 0.5, 1.5), rnd(35, (C,), -0.3, 0.3), rnd(36, (C,), -0.2, 0.2)
    sc, sh = rnd(37, (C,), 0.5, 1.5), rnd(38, (C,), -0.5, 0.5)
    cv = lambda t: t.view(1, -1, 1, 1)
    dY = bf(g * cv(ga) + g2 * cv(gb) + cv(gc))
    a = bf(swish(x * cv(sc) + cv(sh))) if act else x
    want = torch.nn.grad.conv2d_weight(a, (C, 1, k, k), dY, stride=s, padding=pad, groups=C)
    dw0 = rnd(39, (C, 1, k, k))
    dw = dw0.clone().to(dev)
    t = [v.to(dev) for v in (ga, gb, gc, sc, sh)]
    gq, g2q, xq = nhwc(g, dev), nhwc(g2, dev), nhwc(x, dev)
    check(lib().cx_dwconv_wgrad(ptr(gq), ptr(g2q), ptr(t[0]), ptr(t[1]), ptr(t[2]), ptr(xq), ptr(t[3]) if act else None,
                                ptr(t[4]) if act else None, ptr(dw), B, H, W, C, k, s, pad, None, 0, stream_ptr()), "cx_dwconv_wgrad")
    close(dw.cpu() - dw0, want, 2e-3, "dW")
    scratch = torch.empty(16 << 20, device=dev)          # slab workspace: reproducible sums
    outs = []
    for _ in range(2):
        dws = dw0.clone().to(dev)
        check(lib().cx_dwconv_wgrad(ptr(gq), ptr(g2q), ptr(t[0]), ptr(t[1]), ptr(t[2]), ptr(xq), ptr(t[3]) if act else None,
                                    ptr(t[4]) if act else None, ptr(dws), B, H, W, C, k, s, pad, ptr(scratch), scratch.numel(), stream_ptr()),
              "cx_dwconv_wgrad")
        assert lib().cx_last_slab_floats() > 0
        outs.append(dws)
    assert torch.equal(outs[0], outs[1])
    close(outs[0].cpu() - dw0, (dw.cpu() - dw0), 1e-4, "dW through slabs")


@pytest.mark.parametrize("B,C,R", [(5, 96, 4), (37, 240, 10), (128, 672, 28), (3, 2688, 112)])
def test_se_backward(dev, B, C, R):
    """cx_se_bwd (SELayer FCs, efficientnet.py:70-73) against torch autograd; gradients accumulate into the given buffers."""
    from chexpert_amd._lib import lib, ptr, check, stream_ptr
    pooled = rnd(51, (B, C), -1, 1).requires_grad_(True)
    w1, b1 = rnd(52, (R, C), -0.2, 0.2).requires_grad_(True), rnd(53, (R,), -0.2, 0.2).requires_grad_(True)
    w2, b2 = rnd(54, (C, R), -0.5, 0.5).requires_grad_(True), rnd(55, (C,), -0.2, 0.2).requires_grad_(True)
    ds = rnd(56, (B, C), -1, 1)
    h1 = pooled @ w1.t() + b1
    s = torch.sigmoid(swish(h1) @ w2.t() + b2)
    s.backward(ds)
    init = [rnd(60 + i, t.shape) for i, t in enumerate((w1, b1, w2, b2))]
    d = [t.clone().to(dev) for t in init]
    dpooled = torch.full((B, C), 7.0, device=dev)
    args = [t.detach().to(dev).contiguous() for t in (ds, s, h1, pooled, w1, w2)]
    check(lib().cx_se_bwd(*[ptr(t) for t in args], ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), ptr(dpooled), B, C, R, None, 0, stream_ptr()),
          "cx_se_bwd")
    for got, i0, ref, what in zip(d, init, (w1, b1, w2, b2), ("dW1", "db1", "dW2", "db2")):
        close(got.cpu() - i0, ref.grad, 1e-4, what)
    close(dpooled.cpu(), pooled.grad, 1e-4, "dpooled")
    scratch = torch.empty(8 << 20, device=dev)            # slab workspace: image groups added in order, the same bits every time
    outs = []
    for _ in range(2):
        d2 = [t.clone().to(dev) for t in init]
        check(lib().cx_se_bwd(*[ptr(t) for t in args], ptr(d2[0]), ptr(d2[1]), ptr(d2[2]), ptr(d2[3]), ptr(dpooled), B, C, R, ptr(scratch),
                              scratch.numel(), stream_ptr()), "cx_se_bwd")
        outs.append(d2)
    assert all(torch.equal(a_, b_) for a_, b_ in zip(outs[0], outs[1]))
    for got, i0, ref, what in zip(outs[0], init, (w1, b1, w2, b2), ("dW1", "db1", "dW2", "db2")):
        close(got.cpu() - i0, ref.grad, 1e-4, what + " through slabs")
