"""Thin tensor-level wrappers over the C ABI (one call = one kernel launch on the current stream).

Tensors: activations are torch.bfloat16 NHWC views (B,H,W,C) whose channel pitch may exceed C (a
slice of a dense-block buffer); statistics / coefficient vectors / parameter gradients are fp32.
"""
import os

import torch

from . import _lib as L
from ._lib import (EPI_JOIN, EPI_MASK, EPI_STORE, MODE_CONV, MODE_POOL2, MODE_STEM, PRO_AFFINE2, PRO_AFFINE_RELU, PRO_JOIN, PRO_NONE,
                   CxConv, CxWgrad, check, lib, ptr, require_cuda, stream_ptr)
import ctypes as C


def _nhwc(t):
    """(B,H,W,C) view -> (B,H,W,C,pitch).  bf16 is the fast path; fp32 tensors select the fp32 storage mode (CX_DT_F32)."""
    assert t.dtype in (torch.bfloat16, torch.float32) and t.dim() == 4, "expected a bf16 (or fp32-mode) NHWC tensor"
    B, H, W, Cc = t.shape
    sb, sh, sw, sc = t.stride()
    assert sc == 1 and sh == W * sw and sb == H * sh, "NHWC slice must be dense in (B,H,W) with a channel pitch"
    return B, H, W, Cc, sw


# Workspace of the reproducible weight-gradient sums (CxWgrad.scratch): one slab buffer per (device, stream) -- kernels on one stream
# are serialised, the weight-gradient kernels of the side stream run beside those of the main stream.  The engines whose statistics
# are deterministic (plain DenseNet / ResNet) switch it on for their backward pass (set_det_wgrad), which makes the whole training
# step bit-reproducible; measured cost +1.3 % on DenseNet121 bs=256 (1.3 GB of slab traffic per step), none on ResNet152.
# CHEXPERT_DET_WGRAD=0 keeps the fp32 atomics everywhere.  A launch whose splits x |dW| exceed the buffer falls back to atomics.
WGRAD_SCRATCH_DEFAULT = 0 if os.environ.get("CHEXPERT_DET_WGRAD", "1") == "0" else 16 << 20
WGRAD_SCRATCH_FLOATS = 0


def set_det_wgrad(on):
    global WGRAD_SCRATCH_FLOATS
    WGRAD_SCRATCH_FLOATS = WGRAD_SCRATCH_DEFAULT if on else 0


_wgrad_scratch = {}


def wgrad_scratch(device):
    if WGRAD_SCRATCH_FLOATS <= 0:
        return None
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    t = _wgrad_scratch.get(key)
    if t is None or t.numel() != WGRAD_SCRATCH_FLOATS:
        t = _wgrad_scratch[key] = torch.empty(WGRAD_SCRATCH_FLOATS, dtype=torch.float32, device=device)
    return t


# ---- deferred slab sums (cx_wgrad_defer): one table-driven launch per backward pass instead of one small launch per weight gradient
class _WgradArena:
    """Slab memory of one backward pass: every weight-gradient launch between defer_begin() and defer_flush() gets its own
    region (bump allocation in launch order, so the addresses -- and with them the descriptor table -- repeat from step to step),
    and the ordered sums into dw run in ONE launch at the flush.  The table lives on the device, keyed by its content: a repeated
    schedule uploads nothing (which also makes the flush capturable in a hipGraph after a warm-up step).  The arena grows to what
    a pass needed (DenseNet121 at 256 images: ~2 GB of the 288): a launch that does not fit runs its sum at once, as without
    deferral, and the next pass finds a larger arena."""
    START = int(os.environ.get("CHEXPERT_WGRAD_ARENA_MB", "256")) * (1 << 20) // 4
    MIN_FREE = 16 << 20            # a launch is only deferred while this much is left (what covers every layer, see above)

    def __init__(self, device):
        self.device = device
        self.buf = torch.empty(self.START, dtype=torch.float32, device=device)
        self.off = 0
        self.need = 0
        self.tables = {}
        self.active = False


_arenas = {}


def _arena(device):
    a = _arenas.get(device.index)
    if a is None:
        a = _arenas[device.index] = _WgradArena(device)
    return a


def wgrad_defer_begin(device):
    """Start deferring the slab sums of this thread's weight-gradient launches (no-op without the slab workspace)."""
    if WGRAD_SCRATCH_FLOATS <= 0:
        return False
    a = _arena(device)
    if a.need > a.buf.numel():                       # the previous pass did not fit
        a.buf = None
        a.buf = torch.empty(int(a.need * 1.05) + a.MIN_FREE, dtype=torch.float32, device=device)
        a.tables.clear()
    a.off, a.need, a.active = 0, 0, True
    lib().cx_wgrad_defer(1)
    return True


def _wgrad_ws(device):
    """(tensor to hand to the launch as scratch, arena or None, deferred?)."""
    a = _arenas.get(device.index)
    if a is not None and a.active:
        if a.buf.numel() - a.off >= a.MIN_FREE:
            return a.buf[a.off:], a, True
        lib().cx_wgrad_defer(0)                      # this launch sums at once from the per-stream scratch
        return wgrad_scratch(device), a, False
    return wgrad_scratch(device), None, False


def _wgrad_used(a, deferred):
    if a is not None:
        n = (lib().cx_last_slab_floats() + 63) // 64 * 64
        a.need += n
        if deferred:
            a.off += n
        else:
            lib().cx_wgrad_defer(1)


def wgrad_defer_flush(device, keep=False):
    """Add every deferred slab to its dw (one launch on the current stream, which must have been joined with the producers) and
    return to immediate sums -- or, with `keep`, go on deferring (a data-parallel backward flushes before each bucket's
    all-reduce: the gradients of the bucket are then final, the slabs of the later layers keep collecting)."""
    a = _arenas.get(device.index)
    if a is None or not a.active:
        return
    a.active = bool(keep)
    cap = 1024
    arr = (L.CxReduceDesc * cap)()
    blocks = C.c_int64(0)
    n = lib().cx_wgrad_defer_take(arr, cap, C.byref(blocks))
    if not keep:
        lib().cx_wgrad_defer(-1)
    if n < 0:
        raise RuntimeError("more than %d deferred weight-gradient sums" % cap)
    if n == 0:
        return
    key = bytes(memoryview(arr).cast("B")[:n * C.sizeof(L.CxReduceDesc)])
    t = a.tables.get(key)
    if t is None:
        if len(a.tables) > 256:
            a.tables.clear()
        # (an upload from pageable memory: not legal while a hipGraph is being captured -- the tables of a captured step are those
        # of the warm-up steps, every address in them is a persistent buffer)
        t = a.tables[key] = torch.frombuffer(bytearray(key), dtype=torch.uint8).to(device)
    check(lib().cx_dw_reduce_table(ptr(t), n, blocks.value, stream_ptr()), "cx_dw_reduce_table")


def wgrad_defer_abort(device):
    """Leave deferral without running the sums (error paths; no-op after a flush)."""
    a = _arenas.get(device.index)
    if a is not None and a.active:
        a.active = False
        lib().cx_wgrad_defer(-1)


def conv_gemm(x, w_packed, y, *, fused_dw=None, **kw):
    """cx_conv_gemm; with `fused_dw` (fp32 OIHW gradient of the forward 1x1 weight) cx_conv1x1_dgrad_wgrad instead: the input
    gradient with the mask epilogue AND the weight gradient of the same bottleneck convolution in one pass."""
    p = _conv_params(x, w_packed, y, **kw)
    if fused_dw is None:
        check(lib().cx_conv_gemm(C.byref(p), stream_ptr()), "cx_conv_gemm")
    else:
        require_cuda(fused_dw)
        assert x.dtype == torch.bfloat16, "the fused 1x1 input + weight gradient is a bf16 kernel"
        assert fused_dw.dtype == torch.float32
        ws, arena, dfr = _wgrad_ws(fused_dw.device)
        check(lib().cx_conv1x1_dgrad_wgrad_ld_ws(C.byref(p), ptr(fused_dw), _dw_pitch(fused_dw, p.N), ptr(ws),
                                                 0 if ws is None else ws.numel(), stream_ptr()), "cx_conv1x1_dgrad_wgrad_ld_ws")
        _wgrad_used(arena, dfr)
    return lib().cx_last_stat_rows() if p.stat_det else None      # stat_det: rows the consumer has to sum


def _dw_pitch(dw, n):
    """Row pitch of a (128, n[, 1, 1]) fp32 weight gradient that may be a column range of a wider matrix."""
    if dw.dim() == 1:
        assert dw.numel() == 128 * n and dw.is_contiguous()
        return n
    assert dw.shape[0] == 128 and dw.shape[1] == n and dw.numel() == 128 * n and dw.stride(1) == 1 and dw.stride(0) >= n, \
        (tuple(dw.shape), dw.stride())
    return dw.stride(0)


def conv1x1_bwd_pair(a, b, dw_a, dw_b):
    """cx_conv1x1_dgrad_wgrad_pair_ws: the fused 1x1 backward of TWO dense layers in one pass over the channels both read.  `a`
    (the later layer) and `b` are (x, w_packed, y, keywords) as `conv_gemm` takes them; dw_a may be a column range of the later
    layer's wider weight gradient.  Returns the statistic rows each layer wrote (stat_det)."""
    pa = _conv_params(a[0], a[1], a[2], **a[3])
    pb = _conv_params(b[0], b[1], b[2], **b[3])
    require_cuda(dw_a, dw_b)
    assert dw_a.dtype == torch.float32 and dw_b.dtype == torch.float32 and dw_b.is_contiguous()
    ws, arena, dfr = _wgrad_ws(dw_a.device)
    check(lib().cx_conv1x1_dgrad_wgrad_pair_ws(C.byref(pa), C.byref(pb), ptr(dw_a), _dw_pitch(dw_a, pa.N), ptr(dw_b), ptr(ws),
                                               0 if ws is None else ws.numel(), stream_ptr()), "cx_conv1x1_dgrad_wgrad_pair_ws")
    _wgrad_used(arena, dfr)
    return lib().cx_last_stat_rows() if pa.stat_det else None


def last_stat_rows():
    return lib().cx_last_stat_rows()


def last_kernel():
    """The kernel instantiation the most recent conv / weight-gradient call of this thread dispatched to, as rocprofv3 spells it."""
    return lib().cx_last_kernel().decode()


def last_pro_out():
    """True when the last conv_gemm of this thread wrote its `pro_out` side tensor (the selected kernel supports it)."""
    return bool(lib().cx_last_pro_out())


def kernel_hint(on=-1, form=-1):
    """CxConv.kernel_hint / CxWgrad.kernel_hint (ABI 10; include/chexpert_hip.h CX_KERNEL_HINT): pins the kernel family (on = 0 the
    generic kernels, 1 the tiled ones) and tile form of ONE call -- tests and micro-benchmarks only; 0 = the library picks."""
    return (0 if on < 0 else on + 1) | ((0 if form < 0 else form + 1) << 8)


KERNEL_HINT = 0          # default hint of the calls made while it is set (tests / scratch benchmarks: `ops.KERNEL_HINT = kernel_hint(1, 3)`)


def _conv_params(x, w_packed, y, *, N, kh=1, kw=1, stride=1, pad=0, mode=MODE_CONV, prologue=PRO_NONE, pa=None, pb=None,
                 pc=None, x2=None, epilogue=EPI_STORE, stat_sum=None, stat_sq=None, ex=None, e_sc=None, e_sh=None,
                 e_mu=None, e_r=None, e_scale=None, accumulate=False, K=None, tstride=1, stat_replicas=1, stat_rstride=0,
                 stat_det=False, pro_out=None, emask=None, x3=None, po_lo=None, po_mask=None, dil=1, hint=None):
    require_cuda(x, w_packed, y)
    p = CxConv()
    p.kernel_hint = KERNEL_HINT if hint is None else hint
    B, H, W, Cx, ldx = _nhwc(x)
    By, Ho, Wo, Cy, ldy = _nhwc(y)
    assert By == B and Cy == N
    p.x, p.w, p.y = ptr(x), ptr(w_packed), ptr(y)
    p.B, p.H, p.W, p.Ho, p.Wo = B, H, W, Ho, Wo
    p.K = (32 if mode == MODE_STEM else Cx) if K is None else K
    p.N = N
    p.ldx, p.ldy = ldx, ldy
    p.kh, p.kw, p.stride, p.pad = kh, kw, stride, pad
    p.prologue, p.mode, p.epilogue, p.accumulate = prologue, mode, epilogue, int(accumulate)
    p.dil = dil
    p.tstride = tstride
    p.pa, p.pb, p.pc = ptr(pa), ptr(pb), ptr(pc)
    if x2 is not None:
        assert x2.shape == x.shape
        p.x2, p.ldx2 = ptr(x2), _nhwc(x2)[4]
    p.stat_sum, p.stat_sq = ptr(stat_sum), ptr(stat_sq)
    p.stat_replicas, p.stat_rstride, p.stat_det = stat_replicas, stat_rstride, int(bool(stat_det))
    p.dtype = 1 if x.dtype == torch.float32 else 0
    for t_ in (w_packed, y, x2, ex):
        assert t_ is None or t_.dtype == x.dtype, "all tensors of one convolution share the storage type"
    if ex is not None:
        assert ex.shape == y.shape
        p.ex, p.ldex = ptr(ex), _nhwc(ex)[4]
    p.e_sc, p.e_sh, p.e_mu, p.e_r, p.e_scale = ptr(e_sc), ptr(e_sh), ptr(e_mu), ptr(e_r), ptr(e_scale)
    if pro_out is not None:              # dense side output of the prologue (CxConv.pro_out); last_pro_out() says whether it was written
        require_cuda(pro_out)
        assert pro_out.dtype == x.dtype and tuple(pro_out.shape) == tuple(x.shape)
        p.pro_out, p.ldpo = ptr(pro_out), _nhwc(pro_out)[4]
    if emask is not None:                # CX_EPI_JOIN: sign bits of the forward join (affine2_relu's mask)
        require_cuda(emask)
        assert emask.dtype == torch.uint8 and emask.is_contiguous() and emask.numel() * 8 == B * Ho * Wo * N
        p.emask = ptr(emask)
    if prologue == PRO_JOIN:             # the residual join of the block below as this conv1's prologue (CxConv.x3 / po_lo / po_mask)
        assert x2 is not None and pro_out is not None and x.is_contiguous() and x2.is_contiguous()
        for t_, n_ in ((x3, x.numel()), (po_lo, x.numel()), (po_mask, x.numel() // 8)):
            if t_ is not None:
                require_cuda(t_)
                assert t_.dtype in (torch.int8, torch.uint8) and t_.is_contiguous() and t_.numel() == n_
        p.x3, p.po_lo, p.po_mask = ptr(x3), ptr(po_lo), ptr(po_mask)
    p._keep = (x, w_packed, y, pa, pb, pc, x2, stat_sum, stat_sq, ex, e_sc, e_sh, e_mu, e_r, e_scale, pro_out, emask, x3, po_lo, po_mask)      # keep the views alive
    return p


def conv_wgrad(g, x, dw, *, kh=1, kw=1, stride=1, pad=0, mode=MODE_CONV, g_prologue=PRO_NONE, g2=None, ga=None, gb=None,
               gc=None, x_prologue=PRO_NONE, pa=None, pb=None, splits=0, K=None, dil=1, hint=None):
    require_cuda(g, x, dw)
    p = CxWgrad()
    p.kernel_hint = KERNEL_HINT if hint is None else hint
    p.dil = dil
    B, Ho, Wo, N, ldg = _nhwc(g)
    Bx, H, W, Cx, ldx = _nhwc(x)
    assert Bx == B and dw.dtype == torch.float32 and dw.is_contiguous()
    p.g, p.x, p.dw = ptr(g), ptr(x), ptr(dw)
    p.B, p.H, p.W, p.Ho, p.Wo = B, H, W, Ho, Wo
    p.K = (32 if mode == MODE_STEM else Cx) if K is None else K
    p.N = N
    p.ldg, p.ldx = ldg, ldx
    if g2 is not None:
        assert g2.shape == g.shape
        p.g2, p.ldg2 = ptr(g2), _nhwc(g2)[4]
    p.ga, p.gb, p.gc, p.pa, p.pb = ptr(ga), ptr(gb), ptr(gc), ptr(pa), ptr(pb)
    p.kh, p.kw, p.stride, p.pad = kh, kw, stride, pad
    p.g_prologue, p.x_prologue, p.mode, p.splits = g_prologue, x_prologue, mode, splits
    p.dtype = 1 if g.dtype == torch.float32 else 0
    assert x.dtype == g.dtype and (g2 is None or g2.dtype == g.dtype)
    ws, arena, dfr = _wgrad_ws(dw.device)
    if ws is not None:
        p.scratch, p.scratch_floats = ptr(ws), ws.numel()
    check(lib().cx_conv_wgrad(C.byref(p), stream_ptr()), "cx_conv_wgrad")
    _wgrad_used(arena, dfr)


def conv3x3_wgrad_batch(items):
    """cx_conv3x3_wgrad_batch: the 3x3 weight gradients of several dense layers of one block in ONE launch.
    items: [(g dense (B,H,W,32) gradient slice, x saved bottleneck tensor (B,H,W,128), pa, pb norm2 scale / shift, dw fp32 OIHW)].
    Returns False when the library declines the shape / workspace (the caller then launches cx_conv_wgrad per layer)."""
    n = len(items)
    if n == 0:
        return True
    g0, x0 = items[0][0], items[0][1]
    if n > L.WGRAD_BATCH_MAX or g0.dtype != torch.bfloat16:
        return False
    p, bt = CxWgrad(), L.CxWgradBatch()
    B, H, W, N, ldg = _nhwc(g0)
    _, _, _, K, ldx = _nhwc(x0)
    p.B, p.H, p.W, p.Ho, p.Wo, p.K, p.N = B, H, W, H, W, K, N
    p.ldg, p.ldx = ldg, ldx
    p.kh, p.kw, p.stride, p.pad = 3, 3, 1, 1
    p.g_prologue, p.x_prologue, p.mode, p.dtype = PRO_NONE, PRO_AFFINE_RELU, MODE_CONV, 0
    for i, (g, x, pa, pb, dw) in enumerate(items):
        require_cuda(g, x, dw)
        assert _nhwc(g) == (B, H, W, N, ldg) and _nhwc(x) == (B, H, W, K, ldx) and g.dtype == x.dtype == torch.bfloat16
        assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.numel() == N * K * 9
        bt.g[i], bt.x[i], bt.pa[i], bt.pb[i], bt.dw[i] = ptr(g), ptr(x), ptr(pa), ptr(pb), ptr(dw)
    bt.n = n
    ws, arena, dfr = _wgrad_ws(g0.device)
    if ws is None:
        return False
    p.scratch, p.scratch_floats = ptr(ws), ws.numel()
    rc = lib().cx_conv3x3_wgrad_batch(C.byref(p), C.byref(bt), stream_ptr())
    if rc == -4:                               # CX_EUNSUPPORTED: nothing was launched
        if arena is not None and not dfr:
            lib().cx_wgrad_defer(1)
        return False
    check(rc, "cx_conv3x3_wgrad_batch")
    _wgrad_used(arena, dfr)
    return True


def pack_weights(w, transpose=False, stem=False, out=None):
    """OIHW fp32 -> packed bf16 (see cx_pack_weights)."""
    require_cuda(w)
    O, I, kh, kw = w.shape
    n = 7 * O * 32 if stem else kh * kw * O * I
    if out is None:
        out = torch.empty(n, dtype=torch.bfloat16, device=w.device)
    assert w.is_contiguous() and w.dtype == torch.float32 and out.numel() >= n
    check(lib().cx_pack_weights(ptr(w), ptr(out), O, I, kh, kw, int(transpose), int(stem), stream_ptr()), "cx_pack_weights")
    return out


def _fn(name, t):
    """C entry point for the storage type of tensor t (bf16 or the fp32 mode)."""
    return getattr(lib(), name + "_f32" if t.dtype == torch.float32 else name)


def pack_weights_table(flat, packed, desc_dev, n_desc):
    if packed.dtype == torch.float32:
        check(lib().cx_pack_weights_table_f32(ptr(flat), ptr(packed), ptr(desc_dev), n_desc, stream_ptr()), "cx_pack_weights_table_f32")
    else:
        check(lib().cx_pack_weights_table(ptr(flat), ptr(packed), ptr(desc_dev), n_desc, stream_ptr()), "cx_pack_weights_table")


def u8_to_nhwc4(x, out=None, mean=0.5330, std=0.0349):
    """uint8 grey images (B,1,H,W) or (B,H,W) -> whitened, channel-expanded (B,H,W,4) bf16 (chexpert.py:70-72 on the GPU)."""
    require_cuda(x)
    assert x.dtype == torch.uint8 and x.is_contiguous()
    if x.dim() == 4:
        assert x.shape[1] == 1
        B, _, H, W = x.shape
    else:
        B, H, W = x.shape
    if out is None:
        out = torch.empty(B, H, W, 4, dtype=torch.bfloat16, device=x.device)
    check(_fn("cx_u8_to_nhwc4", out)(ptr(x), ptr(out), B * H * W, mean, std, stream_ptr()), "cx_u8_to_nhwc4")
    return out


def u8_jitter(x, brightness, contrast, order, out=None):
    """ColorJitter(brightness, contrast) of explore_data.ipynb cell 6 on decoded grey bytes (B,1,H,W) / (B,H,W) uint8, on the GPU;
    brightness / contrast: fp32 (B,) factors, order: int32 (B,), 0 = brightness first."""
    require_cuda(x, brightness, contrast, order)
    assert x.dtype == torch.uint8 and x.is_contiguous() and order.dtype == torch.int32
    B = x.shape[0]
    HW = x.numel() // B
    if out is None:
        out = torch.empty_like(x)
    check(lib().cx_u8_jitter(ptr(x), ptr(out), B, HW, ptr(brightness), ptr(contrast), ptr(order), stream_ptr()), "cx_u8_jitter")
    return out


def nchw3_to_nhwc4(x, out=None):
    require_cuda(x)
    B, Cc, H, W = x.shape
    assert Cc == 3 and x.dtype == torch.float32 and x.is_contiguous()
    if out is None:
        out = torch.empty(B, H, W, 4, dtype=torch.bfloat16, device=x.device)
    check(_fn("cx_nchw3_to_nhwc4", out)(ptr(x), ptr(out), B, H, W, stream_ptr()), "cx_nchw3_to_nhwc4")
    return out


def bn_coef(s, q, count, gamma, beta, eps, momentum, rmean, rvar, scale, shift, mean, rstd, Cn=None, replicas=1, rstride=0):
    Cn = Cn if Cn is not None else s.numel()
    check(lib().cx_bn_coef(ptr(s), ptr(q), float(count), ptr(gamma), ptr(beta), eps, momentum, ptr(rmean), ptr(rvar),
                           ptr(scale), ptr(shift), ptr(mean), ptr(rstd), Cn, replicas, rstride, stream_ptr()), "cx_bn_coef")


def bn_coef_moments(mean, rstd, count, gamma, beta, eps, momentum, rmean, rvar, scale, shift, Cn, fresh=None):
    """cx_bn_coef_moments; fresh = (sum, sq, rows, rstride, c_lo, c_n) reduces those channels from deterministic statistic rows first."""
    fs, fq, rows, rstride, c_lo, c_n = fresh if fresh is not None else (None, None, 0, 0, 0, 0)
    check(lib().cx_bn_coef_moments(ptr(mean), ptr(rstd), float(count), ptr(gamma), ptr(beta), eps, momentum, ptr(rmean), ptr(rvar),
                                   ptr(scale), ptr(shift), Cn, ptr(fs), ptr(fq), rows, rstride, c_lo, c_n, stream_ptr()),
          "cx_bn_coef_moments")


def bn_coef_eval(rmean, rvar, gamma, beta, eps, scale, shift, mean, rstd, Cn=None):
    Cn = Cn if Cn is not None else rmean.numel()
    check(lib().cx_bn_coef_eval(ptr(rmean), ptr(rvar), ptr(gamma), ptr(beta), eps, ptr(scale), ptr(shift), ptr(mean),
                                ptr(rstd), Cn, stream_ptr()), "cx_bn_coef_eval")


def bn_bwd_coef(S1, S2, count, gamma, mean, rstd, dgamma, dbeta, A, Bc, pa, pb, pc, Cn, replicas=1, rstride=0, q=None):
    """q = (qa, qb, qc, q_lo, q_n): also emit the slice coefficients (cx_bn_bwd_slice_coef) of channels [q_lo, q_lo + q_n)."""
    qa, qb, qc, q_lo, q_n = q if q is not None else (None, None, None, 0, 0)
    check(lib().cx_bn_bwd_coef(ptr(S1), ptr(S2), float(count), ptr(gamma), ptr(mean), ptr(rstd), ptr(dgamma), ptr(dbeta),
                               ptr(A), ptr(Bc), ptr(pa), ptr(pb), ptr(pc), Cn, replicas, rstride, ptr(qa), ptr(qb), ptr(qc), q_lo, q_n,
                               stream_ptr()), "cx_bn_bwd_coef")


def bn_bwd_slice_coef(A, Bc, mean, rstd, pa, pb, pc, Cn):
    check(lib().cx_bn_bwd_slice_coef(ptr(A), ptr(Bc), ptr(mean), ptr(rstd), ptr(pa), ptr(pb), ptr(pc), Cn, stream_ptr()),
          "cx_bn_bwd_slice_coef")


def dropout_slice_fwd(y, p, seed, uid, S1=None, S2=None, stat_rows=0):
    """cx_dropout_slice_fwd: dropout in place on the channel slice y (a (B,H,W,C) view of a wider NHWC buffer); returns the statistic
    rows written (S1 / S2 given)."""
    B, H, W, Cc, ld = _nhwc(y)
    require_cuda(y, seed)
    assert seed.dtype == torch.int64
    check(_fn("cx_dropout_slice_fwd", y)(ptr(y), ld, B * H * W, Cc, float(p), ptr(seed), int(uid), ptr(S1), ptr(S2), stat_rows,
                                         stream_ptr()), "cx_dropout_slice_fwd")
    return lib().cx_last_stat_rows() if S1 is not None else None


def dropout_slice_bwd(g, x, qa, qb, qc, p, seed, uid):
    """cx_dropout_slice_bwd: g <- keep ? (qa g + qb x + qc) / (1 - p) : 0 in place on the gradient slice."""
    B, H, W, Cc, ldg = _nhwc(g)
    assert x.shape == g.shape and seed.dtype == torch.int64
    require_cuda(g, x, seed)
    check(_fn("cx_dropout_slice_bwd", g)(ptr(g), ldg, ptr(x), _nhwc(x)[4], ptr(qa), ptr(qb), ptr(qc), B * H * W, Cc, float(p), ptr(seed),
                                         int(uid), stream_ptr()), "cx_dropout_slice_bwd")


def bnrelu_maxpool_fwd(x, scale, shift, y, argmax, stat_sum, stat_sq, stat_rows=0):
    B, H, W, Cc, ldx = _nhwc(x)
    assert ldx == Cc
    ldy = _nhwc(y)[4]
    check(_fn("cx_bnrelu_maxpool_fwd", x)(ptr(x), ptr(scale), ptr(shift), ptr(y), ptr(argmax), ptr(stat_sum), ptr(stat_sq),
                                      B, H, W, Cc, ldy, stat_rows, stream_ptr()), "cx_bnrelu_maxpool_fwd")
    return lib().cx_last_stat_rows() if stat_rows else None


def bnrelu_maxpool_bwd(x, scale, shift, mean, rstd, argmax, g, gx, ga, gb, gc, dz, S1, S2, stat_rows=0):
    B, H, W, Cc, ldx = _nhwc(x)
    assert ldx == Cc and _nhwc(dz)[4] == Cc
    check(_fn("cx_bnrelu_maxpool_bwd", x)(ptr(x), ptr(scale), ptr(shift), ptr(mean), ptr(rstd), ptr(argmax), ptr(g), ptr(gx),
                                      ptr(ga), ptr(gb), ptr(gc), ptr(dz), ptr(S1), ptr(S2), B, H, W, Cc, _nhwc(g)[4],
                                      _nhwc(gx)[4], stat_rows, stream_ptr()), "cx_bnrelu_maxpool_bwd")
    return lib().cx_last_stat_rows() if stat_rows else None


def head_fwd(x, scale, shift, w, bias, pooled, logits):
    B, H, W, Cc, ldx = _nhwc(x)
    check(_fn("cx_head_fwd", x)(ptr(x), ptr(scale), ptr(shift), ptr(w), ptr(bias), ptr(pooled), ptr(logits), B, H * W, Cc, ldx,
                            logits.shape[1], stream_ptr()), "cx_head_fwd")


def bce_fwd_bwd(logits, target, loss, loss_elem, dlogits, grad_scale=1.0):
    B, n = logits.shape
    check(lib().cx_bce_fwd_bwd(ptr(logits), ptr(target), ptr(loss), ptr(loss_elem), ptr(dlogits), grad_scale, B, n,
                               stream_ptr()), "cx_bce_fwd_bwd")


def softmax_ce_fwd_bwd(logits, target, loss, loss_elem, dlogits, grad_scale=1.0):
    """CrossEntropyLoss forward + gradient in one launch (fp32 logits [B, n], int64 class indices [B])."""
    B, n = logits.shape
    check(lib().cx_softmax_ce_fwd_bwd(ptr(logits), ptr(target), ptr(loss), ptr(loss_elem), ptr(dlogits), grad_scale, B, n,
                                      stream_ptr()), "cx_softmax_ce_fwd_bwd")


def head_bwd(dlogits, pooled, w, dw, db, dpooled):
    B, n = dlogits.shape
    check(lib().cx_head_bwd(ptr(dlogits), ptr(pooled), ptr(w), ptr(dw), ptr(db), ptr(dpooled), B, pooled.shape[1], n,
                            stream_ptr()), "cx_head_bwd")


def gap_relu_bn_bwd(dpooled, x, scale, shift, mean, rstd, e_scale, g, S1, S2, stat_rows=0):
    B, H, W, Cc, ldx = _nhwc(x)
    check(_fn("cx_gap_relu_bn_bwd", x)(ptr(dpooled), ptr(x), ptr(scale), ptr(shift), ptr(mean), ptr(rstd), ptr(e_scale), ptr(g),
                                   ptr(S1), ptr(S2), B, H * W, Cc, ldx, _nhwc(g)[4], stat_rows, stream_ptr()), "cx_gap_relu_bn_bwd")
    return lib().cx_last_stat_rows() if stat_rows else None


def unpool2_mask(d, x, sc, sh, mean, rstd, e_scale, g, S1, S2, stat_rows=0):
    B, H, W, Cc, ldx = _nhwc(x)
    check(_fn("cx_unpool2_mask", x)(ptr(d), ptr(x), ptr(sc), ptr(sh), ptr(mean), ptr(rstd), ptr(e_scale), ptr(g), ptr(S1), ptr(S2),
                                B, H, W, Cc, _nhwc(d)[4], ldx, _nhwc(g)[4], stat_rows, stream_ptr()), "cx_unpool2_mask")
    return lib().cx_last_stat_rows() if stat_rows else None


def affine2_inplace(dz, x, pa, pb, pc):
    B, H, W, Cc, ld = _nhwc(dz)
    assert ld == Cc and _nhwc(x)[4] == Cc
    check(lib().cx_affine2_inplace(ptr(dz), ptr(x), ptr(pa), ptr(pb), ptr(pc), B * H * W, Cc, stream_ptr()),
          "cx_affine2_inplace")


def affine2_relu(a, b, pa, pb, pc, out, mask=None):
    """mask (optional, uint8 [rows * C / 8]): the sign bits of `out` for relu_bwd_stats (cx_affine2_relu_mask)."""
    B, H, W, Cc, ld = _nhwc(a)
    assert ld == Cc and _nhwc(b)[4] == Cc and _nhwc(out)[4] == Cc
    assert mask is None or (mask.dtype == torch.uint8 and mask.numel() == B * H * W * Cc // 8)
    check(_fn("cx_affine2_relu_mask", a)(ptr(a), ptr(b), ptr(pa), ptr(pb), ptr(pc), ptr(out), ptr(mask), B * H * W, Cc, stream_ptr()),
          "cx_affine2_relu_mask")


def join_fwd(a, b, b_lo, pa, pb, pc, out, out_lo, mask=None):
    """cx_join_fwd: out = relu(a*pa + (b [+ b_lo])*pb + pc) as hi (bf16 `out`) + lo (int8 `out_lo`) planes + sign bits `mask`."""
    require_cuda(a, b, out)
    B, H, W, Cc, ld = _nhwc(a)
    assert a.dtype == torch.bfloat16 and ld == Cc and _nhwc(b)[4] == Cc and _nhwc(out)[4] == Cc
    n = B * H * W * Cc
    for t_ in (b_lo, out_lo):
        assert t_ is None or (t_.dtype in (torch.int8, torch.uint8) and t_.is_contiguous() and t_.numel() == n)
    assert mask is None or (mask.dtype == torch.uint8 and mask.numel() == n // 8)
    check(lib().cx_join_fwd(ptr(a), ptr(b), ptr(b_lo), ptr(pa), ptr(pb), ptr(pc), ptr(out), ptr(out_lo), ptr(mask), B * H * W, Cc,
                            stream_ptr()), "cx_join_fwd")


def relu_bwd_stats(dout, out, a, mu_a, r_a, b, mu_b, r_b, dz, S1, S2a, S2b, stat_rows=0, mask=None):
    """mask (optional): sign bits written by affine2_relu, read instead of `out`."""
    B, H, W, Cc, ld = _nhwc(dout)
    assert ld == Cc
    check(_fn("cx_relu_bwd_stats_mask", dout)(ptr(dout), ptr(out), ptr(mask), ptr(a), ptr(mu_a), ptr(r_a), ptr(b), ptr(mu_b), ptr(r_b),
                                              ptr(dz), ptr(S1), ptr(S2a), ptr(S2b), B * H * W, Cc, stat_rows, stream_ptr()),
          "cx_relu_bwd_stats_mask")
    return lib().cx_last_stat_rows() if stat_rows else None


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    check(lib().cx_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, weight_decay, step, grad_scale,
                             stream_ptr()), "cx_adam_step")


def sgd_nesterov_step(p, g, buf, lr, momentum, weight_decay, first_step, grad_scale=1.0):
    check(lib().cx_sgd_nesterov_step(ptr(p), ptr(g), ptr(buf), p.numel(), lr, momentum, weight_decay, int(first_step),
                                     grad_scale, stream_ptr()), "cx_sgd_nesterov_step")


def rmsprop_step(p, g, sq, buf, lr, alpha, eps, momentum, weight_decay, grad_scale=1.0):
    check(lib().cx_rmsprop_step(ptr(p), ptr(g), ptr(sq), ptr(buf), p.numel(), lr, alpha, eps, momentum, weight_decay,
                                grad_scale, stream_ptr()), "cx_rmsprop_step")


def adam_step_dev(p, g, m, v, hyper, beta1, beta2, eps, weight_decay, grad_scale=1.0):
    check(lib().cx_adam_step_dev(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), ptr(hyper), beta1, beta2, eps, weight_decay, grad_scale,
                                 stream_ptr()), "cx_adam_step_dev")


def sgd_nesterov_step_dev(p, g, buf, hyper, momentum, weight_decay, grad_scale=1.0):
    check(lib().cx_sgd_nesterov_step_dev(ptr(p), ptr(g), ptr(buf), p.numel(), ptr(hyper), momentum, weight_decay, grad_scale,
                                         stream_ptr()), "cx_sgd_nesterov_step_dev")


def rmsprop_step_dev(p, g, sq, buf, hyper, alpha, eps, momentum, weight_decay, grad_scale=1.0):
    check(lib().cx_rmsprop_step_dev(ptr(p), ptr(g), ptr(sq), ptr(buf), p.numel(), ptr(hyper), alpha, eps, momentum, weight_decay,
                                    grad_scale, stream_ptr()), "cx_rmsprop_step_dev")


def optim_tick(hyper):
    check(lib().cx_optim_tick(ptr(hyper), stream_ptr()), "cx_optim_tick")


def bf16_to_f32_nchw(x, out=None):
    B, H, W, Cc, ldx = _nhwc(x)
    if out is None:
        out = torch.empty(B, Cc, H, W, dtype=torch.float32, device=x.device)
    check(lib().cx_bf16_to_f32_nchw(ptr(x), ptr(out), B, H, W, Cc, ldx, stream_ptr()), "cx_bf16_to_f32_nchw")
    return out


# ---- attention-augmented convolution pieces (csrc/aaconv.hip)
def aa_attention_fwd(qkv, key_rel_h, key_rel_w, o, lse, nh, dk, dv):
    B, H, W, Cq, ldq = _nhwc(qkv)
    check(_fn("cx_aa_attention_fwd", qkv)(ptr(qkv), ptr(key_rel_h), ptr(key_rel_w), ptr(o), ptr(lse), B, H, W, nh, dk, dv, ldq, stream_ptr()),
          "cx_aa_attention_fwd")


def aa_attention_weights(qkv, key_rel_h, key_rel_w, lse, nh, dk, dv):
    """softmax(logits) (B, nh, HW, HW) fp32 of the forward that produced `qkv` / `lse` (AAConv2d.weights, attn_aug_conv.py:87)."""
    B, H, W, Cq, ldq = _nhwc(qkv)
    out = torch.empty(B, nh, H * W, H * W, dtype=torch.float32, device=qkv.device)
    check(_fn("cx_aa_attention_weights", qkv)(ptr(qkv), ptr(key_rel_h), ptr(key_rel_w), ptr(lse), ptr(out), B, H, W, nh, dk, dv, ldq,
                                        stream_ptr()), "cx_aa_attention_weights")
    return out


def aa_attention_bwd(qkv, key_rel_h, key_rel_w, o, d_o, lse, dqkv, d_rel_h, d_rel_w, nh, dk, dv):
    """With the slab workspace on (set_det_wgrad) the relative-table gradients are summed in workgroup order (reproducible)."""
    B, H, W, Cq, ldq = _nhwc(qkv)
    ws = wgrad_scratch(qkv.device)                 # partial tables are consumed inside the call: the per-stream scratch is enough
    need = ((H * W + 127) // 128) * B * nh * (dk // nh) * (2 * H - 1 + 2 * W - 1)
    if ws is not None and ws.numel() < need:
        ws = _big_scratch(qkv.device, need)
    check(_fn("cx_aa_attention_bwd", qkv)(ptr(qkv), ptr(key_rel_h), ptr(key_rel_w), ptr(o), ptr(d_o), ptr(lse), ptr(dqkv), ptr(d_rel_h),
                                    ptr(d_rel_w), B, H, W, nh, dk, dv, ldq, ptr(ws), 0 if ws is None else ws.numel(), stream_ptr()),
          "cx_aa_attention_bwd")


_big = {}


def _big_scratch(device, floats):
    """A larger per-(device, stream) workspace for the few calls whose partial results exceed the default slab buffer."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    t = _big.get(key)
    if t is None or t.numel() < floats:
        t = _big[key] = torch.empty(int(floats), dtype=torch.float32, device=device)
    return t


def aa_outproj_fwd(o, w, y, stat_sum, stat_sq, stat_rows=0, stat_rstride=0):
    """stat_rows > 0: deterministic statistic rows (returns the number written), else atomics into stat_sum / stat_sq."""
    B, H, W, dv, ldy = _nhwc(y)
    check(_fn("cx_aa_outproj_fwd", y)(ptr(o), ptr(w), ptr(y), ldy, ptr(stat_sum), ptr(stat_sq), B * H * W, dv, stat_rows, stat_rstride,
                                  stream_ptr()), "cx_aa_outproj_fwd")
    return lib().cx_last_stat_rows() if stat_rows > 0 else None


def aa_outproj_bwd(g, gx, ga, gb, gc, o, w, d_o, dw):
    B, H, W, dv, ldg = _nhwc(g)
    ws, arena, dfr = _wgrad_ws(dw.device)
    check(_fn("cx_aa_outproj_bwd", g)(ptr(g), ldg, ptr(gx), _nhwc(gx)[4], ptr(ga), ptr(gb), ptr(gc), ptr(o), ptr(w), ptr(d_o), ptr(dw),
                                  B * H * W, dv, ptr(ws), 0 if ws is None else ws.numel(), stream_ptr()), "cx_aa_outproj_bwd")
    _wgrad_used(arena, dfr)


def rows_reduce(dst, rows, n_rows, C, rstride, accumulate=True):
    check(lib().cx_rows_reduce(ptr(dst), ptr(rows), n_rows, C, rstride, int(accumulate), stream_ptr()), "cx_rows_reduce")


def copy_stream(src, dst):
    """dst = src through the library's own 16-byte-per-lane copy kernel (the measured stream rate bench.py reports)."""
    require_cuda(src, dst)
    n = src.numel() * src.element_size()
    if dst.numel() * dst.element_size() != n or not (src.is_contiguous() and dst.is_contiguous()):
        raise RuntimeError("copy_stream needs two contiguous buffers of equal size")
    check(lib().cx_copy_stream(ptr(src), ptr(dst), n, stream_ptr()), "cx_copy_stream")


def stats_bc(x, s, q):
    B, H, W, Cc, ldx = _nhwc(x)
    check(_fn("cx_stats_bc", x)(ptr(x), ptr(s), ptr(q), B, H * W, Cc, ldx, stream_ptr()), "cx_stats_bc")


def affine_relu_bc(x, sc, sh, y):
    B, H, W, Cc, ldx = _nhwc(x)
    assert _nhwc(y)[4] == Cc
    check(_fn("cx_affine_relu_bc", x)(ptr(x), ptr(sc), ptr(sh), ptr(y), B, H * W, Cc, ldx, stream_ptr()), "cx_affine_relu_bc")


def in_relu_bwd(da, x, sc, sh, S1, S2, gout):
    B, H, W, Cc, ldx = _nhwc(x)
    assert _nhwc(da)[4] == Cc
    check(_fn("cx_in_relu_bwd", da)(ptr(da), ptr(x), ptr(sc), ptr(sh), ptr(S1), ptr(S2), ptr(gout), B, H * W, Cc, ldx, _nhwc(gout)[4],
                               stream_ptr()), "cx_in_relu_bwd")


def f32_to_bf16(x, y):
    check(lib().cx_f32_to_bf16(ptr(x), ptr(y), x.numel(), stream_ptr()), "cx_f32_to_bf16")
