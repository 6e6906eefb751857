"""EfficientNet B0-B7 (compound-scaled MBConv nets) on the gfx950 kernels.

Drop-in surface of /root/reference/models/efficientnet.py: `construct_model(model_name, n_classes)`
(:188-228), the `nn.Sequential` index layout and therefore the `state_dict` keys (`stem.0/1`,
`blocks.S.B.{0..8}`, SE at `.6.1/.6.3` (or `.3.1/.3.3` when expand_ratio == 1), `head.0/1/6`), `model.head[1]`,
`model.head[-1]`, BatchNorm eps 1e-3 / momentum 0.01 (:140, :174-176), class `__name__` = model name (:226).
Reproduced quirks: symmetric `ceil(total/2)` "same" padding (:53-64), SE width from the block INPUT channels
(:82), skip whenever shapes match (:109).

Schedule per MBConv block (NHWC bf16):
  expand 1x1 (implicit GEMM, raw + stats) -> depthwise k x k with bn+Swish applied on load (raw + stats)
  -> SE: global pool of swish(bn(.)), two tiny FCs -> u = swish(bn(y_d)) * s[b][c] (one pass)
  -> project 1x1 (implicit GEMM) -> x_out = bn(y_p) (+ x_in).
DropConnect (efficientnet.py:44-51, :100-101) and the classifier Dropout (:169-171) are applied in train mode: per-image /
per-element keep masks drawn on the GPU by `cx_dropout_mask_dev` from (model.drop_seed, device-side count of training forwards, block) -- reproducible and
independent of torch's RNG; the parity tests feed the drawn masks (`engine.last_masks`) to the oracle.
"""
import math
import os
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib as L
from .. import ops
from .._lib import CxPackDesc, check, lib, ptr, stream_ptr
from .densenet import BatchNorm2dParams, Conv2dParams, PoolMarker, _FusedOnly

SCALING_PARAMS = {  # width, depth, resolution, dropout (efficientnet.py:13-21)
    "efficientnet-b0": (1.0, 1.0, 224, 0.2), "efficientnet-b1": (1.0, 1.1, 240, 0.2), "efficientnet-b2": (1.1, 1.2, 260, 0.3),
    "efficientnet-b3": (1.2, 1.4, 300, 0.3), "efficientnet-b4": (1.4, 1.8, 380, 0.4), "efficientnet-b5": (1.6, 2.2, 456, 0.4),
    "efficientnet-b6": (1.8, 2.6, 528, 0.5), "efficientnet-b7": (2.0, 3.1, 600, 0.5)}
_BASE = [(1, 32, 16, 3, 1, 1), (2, 16, 24, 3, 2, 6), (2, 24, 40, 5, 2, 6), (3, 40, 80, 3, 2, 6), (3, 80, 112, 5, 1, 6),
         (4, 112, 192, 5, 2, 6), (1, 192, 320, 3, 1, 6)]     # repeats, in, out, k, stride, expand (:149-155)


class Marker(_FusedOnly, nn.Module):
    pass


class DropMarker(_FusedOnly, nn.Module):
    def __init__(self, p):
        super().__init__()
        self.p = p


class SELayer(nn.Sequential):
    def __init__(self, c, r):
        super().__init__(PoolMarker(), Conv2dParams(c, r, 1), Marker(), Conv2dParams(r, c, 1), Marker())


class MBConvBlock(nn.Sequential):
    def __init__(self, cin, cout, k, stride, expand, se_ratio, drop_rate):
        ce = int(cin * expand)
        mods = []
        if expand != 1:
            mods += [Conv2dParams(cin, ce, 1, bias=False), BatchNorm2dParams(ce), Marker()]
        mods += [Conv2dParams(ce, ce, k, stride, groups=ce, bias=False), BatchNorm2dParams(ce), Marker(),
                 SELayer(ce, max(1, int(cin * se_ratio))), Conv2dParams(ce, cout, 1, bias=False), BatchNorm2dParams(cout)]
        if cin == cout and stride == 1:
            mods += [DropMarker(drop_rate)]
        super().__init__(*mods)
        self.cfg = dict(cin=cin, cout=cout, ce=ce, k=k, stride=stride, expand=expand, skip=(cin == cout and stride == 1))


class MBConvBlockRepeat(nn.Sequential):
    def __init__(self, n, cin, cout, k, stride, expand, se_ratio, drop):
        mods = []
        for i in range(n):
            mods.append(MBConvBlock(cin, cout, k, stride, expand, se_ratio, drop * i / n))
            cin, stride = cout, 1
        super().__init__(*mods)


def _round_filters(f, width, div=8):
    new = max(div, int(f * width + div / 2) // div * div)
    if new < 0.9 * f * width:
        new += div
    return int(new)


def same_pad(h_in, k, stride):
    """PaddedConv2d (:53-64): symmetric ceil(total/2)."""
    h_out = math.ceil(h_in / stride)
    return math.ceil(max((h_out - 1) * stride - h_in + (k - 1) + 1, 0) / 2)


class _BN:
    def __init__(self, V, C, fz, bz):
        self.C = C
        self.sum, self.sq = fz.take(C), fz.take(C)
        self.S1, self.S2 = bz.take(C), bz.take(C)
        self.sc, self.sh, self.mean, self.rstd = (V.take(C) for _ in range(4))
        self.pa, self.pb, self.pc = (V.take(C) for _ in range(3))


class _Region:
    def __init__(self, base=0):
        self.n = base

    def take(self, n):
        off = self.n
        self.n += (n + 3) // 4 * 4
        return (off, n)


class _LibF32:
    """The C ABI with the `_f32` twin of an entry point where one exists (the fp32 storage mode)."""

    def __init__(self, l):
        self._l = l

    def __getattr__(self, name):
        return getattr(self._l, name + "_f32", None) or getattr(self._l, name)


class _Engine:
    def __init__(self, model):
        self.model = model
        # activation storage type: bf16, or fp32 = north_star's "1e-3 fp32" parity mode (generic f32-MFMA convolutions, the
        # storage-typed depthwise / squeeze-excite / Swish kernels of csrc/effnet.hip; no tiled fast paths)
        self.dtype = getattr(model, "_storage_dtype", torch.bfloat16)
        # deterministic statistics and weight gradients (statistic rows summed in row order, slab sums, one owner per squeeze-excite
        # sum): two steps on the same batch give the same bits; CHEXPERT_DET=0 keeps the fp32 atomics

        self.det = os.environ.get("CHEXPERT_DET", "1") != "0"
        self.flat = None
        self.device = None
        self.pool = {}
        self.reducer = None
        self.mb = [b for rep in model.blocks for b in rep]
        self.mb_names = ["blocks.%d.%d" % (si, bi) for si, rep in enumerate(model.blocks) for bi, _ in enumerate(rep)]
        self.last_masks = {}         # name -> mask / keep of the most recent train-mode forward (tests, reproducibility)
        self.bns = [model.stem[1]] + [m for b in self.mb for m in b if isinstance(m, nn.BatchNorm2d)] + [model.head[1]]
        tmp = [_Region(0) for _ in range(3)]
        for bn in self.bns:
            _BN(tmp[2], bn.num_features, tmp[0], tmp[1])
        nf, nb = tmp[0].n, tmp[1].n
        self.fz, self.bz, self.rest = _Region(0), _Region(nf), _Region(nf + nb)
        self.bn = {id(bn): _BN(self.rest, bn.num_features, self.fz, self.bz) for bn in self.bns}
        self.fwd_zero, self.bwd_zero = (0, nf), (nf, nb)
        cmax = max(bn.num_features for bn in self.bns)
        self.ones = self.rest.take(cmax)
        self.vec_size = self.rest.n

    # ---- binding
    def bind(self, dev):
        m = self.model
        params = [p for _, p in m.named_parameters()]
        ok = (self.flat is not None and self.device == dev and len(params) == len(self.offsets)
              and all(p.data_ptr() == self.flat.data_ptr() + 4 * off for p, off in zip(params, self.offsets)))
        if ok:
            return
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, off in zip(params, offs):
            flat[off:off + p.numel()].copy_(p.data.reshape(-1))
            p.data = flat[off:off + p.numel()].view(p.shape)
        for b in m.buffers():
            if b.device != dev:
                raise RuntimeError("module buffers are on %s, input on %s -- call model.to(device)" % (b.device, dev))
        self.flat, self.offsets, self.params = flat, offs, params
        self.flat_grad = torch.zeros_like(flat)
        self.grad_views = [self.flat_grad[off:off + p.numel()].view(p.shape) for p, off in zip(params, offs)]
        self.off_of = {id(p): off for p, off in zip(params, offs)}
        self.device = dev
        self.n_classes = m.head[6].out_features
        self.pool = {}
        descs, cur = [], 0
        self.wf, self.wb = {}, {}

        def add(conv, transpose=False):
            nonlocal cur
            O, I, kh, kw = conv.weight.shape
            descs.append(CxPackDesc(self.off_of[id(conv.weight)], cur, O, I, kh, kw, int(transpose), 0))
            off, n = cur, O * I * kh * kw
            cur += (n + 7) // 8 * 8
            return (off, n)
        for b in self.mb + [m.head]:
            for mod in b:
                if isinstance(mod, nn.Conv2d) and mod.groups == 1:
                    self.wf[id(mod)] = add(mod)
                    self.wb[id(mod)] = add(mod, transpose=True)
        self.stem_off = cur
        cur += 9 * m.stem[0].out_channels * 8
        self.packed = torch.empty(cur, dtype=self.dtype, device=dev)
        arr = (CxPackDesc * len(descs))(*descs)
        self.desc_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self.n_desc = len(descs)

    def pack(self):
        ops.pack_weights_table(self.flat, self.packed, self.desc_dev, self.n_desc)
        w8 = F.pad(self.model.stem[0].weight.detach(), (0, 0, 0, 0, 0, 5)).contiguous()       # (O,3,3,3) -> (O,8,3,3)
        if self.dtype == torch.float32:         # [tap][O][I] fp32: a 3 K-element layout copy
            self.packed[self.stem_off:self.stem_off + w8.numel()].copy_(w8.permute(2, 3, 0, 1).reshape(-1))
        else:
            ops.pack_weights(w8, out=self.packed[self.stem_off:])

    def w_fwd(self, conv):
        off, n = self.wf[id(conv)]
        return self.packed[off:off + n]

    def w_bwd(self, conv):
        off, n = self.wb[id(conv)]
        return self.packed[off:off + n]

    def G(self, p):
        off = self.off_of[id(p)]
        return self.flat_grad[off:off + p.numel()]

    # ---- workspace
    def acquire(self, B, H, W):
        lst = self.pool.setdefault((B, H, W), [])
        if lst:
            return lst.pop()
        dev, bf, f32 = self.device, self.dtype, torch.float32
        e = lambda *s, dtype=bf: torch.empty(*s, dtype=dtype, device=dev)
        m = self.model
        ws = type("WS", (), {})()
        ws.key, ws.B, ws.H, ws.W = (B, H, W), B, H, W
        ws.x8 = e(B, H, W, 8)
        p0 = same_pad(H, 3, 2)
        h, w = (H + 2 * p0 - 3) // 2 + 1, (W + 2 * same_pad(W, 3, 2) - 3) // 2 + 1
        c0 = m.stem[0].out_channels
        ws.stem_pad = p0
        ws.ys, ws.x0 = e(B, h, w, c0), e(B, h, w, c0)
        ws.blk = []
        for b in self.mb:
            c = b.cfg
            pd = same_pad(h, c["k"], c["stride"])
            ho, wo = (h + 2 * pd - c["k"]) // c["stride"] + 1, (w + 2 * pd - c["k"]) // c["stride"] + 1
            se = [mod for mod in b if isinstance(mod, SELayer)][0]
            R = se[1].out_channels
            t = dict(hin=(h, w), hout=(ho, wo), pad=pd, R=R,
                     ye=e(B, h, w, c["ce"]) if c["expand"] != 1 else None, yd=e(B, ho, wo, c["ce"]), u=e(B, ho, wo, c["ce"]),
                     yp=e(B, ho, wo, c["cout"]), out=e(B, ho, wo, c["cout"]),
                     pooled=e(B, c["ce"], dtype=f32), h1=e(B, R, dtype=f32), s=e(B, c["ce"], dtype=f32))
            ws.blk.append(t)
            h, w = ho, wo
        ws.yh = e(B, h, w, 1280)
        ws.hw_last = (h, w)
        ws.pooled = e(B, 1280, dtype=f32)
        ws.logits = e(B, self.n_classes, dtype=f32)
        ws.vec = torch.zeros(self.vec_size, dtype=f32, device=dev)
        ws.slab = torch.empty(2, self.SLAB, dtype=f32, device=dev) if self.det else None
        o, n = self.ones
        ws.vec[o:o + n].fill_(1.0)
        ws.bwd = None
        return ws

    def release(self, ws):
        lst = self.pool.setdefault(ws.key, [])
        if len(lst) < 2:
            lst.append(ws)

    @staticmethod
    def _v(ws, slot, n=None):
        off, m = slot
        return ws.vec[off:off + (m if n is None else n)]

    SLAB = 1 << 22               # floats per half of the statistic-row scratch
    ROWS = 2048                  # most statistic rows an element-wise / depthwise producer writes

    def _rows_cap(self, C):
        return min(self.ROWS, self.SLAB // C)

    def _bn_coef(self, ws, bn, count, train, rows=None):
        S, v = self.bn[id(bn)], self._v
        if train:
            ssum, ssq, reps, rstride = (ws.slab[0], ws.slab[1], rows, S.C) if self.det else (v(ws, S.sum), v(ws, S.sq), 1, 0)
            ops.bn_coef(ssum, ssq, count, bn.weight, bn.bias, bn.eps, bn.momentum, bn.running_mean, bn.running_var,
                        v(ws, S.sc), v(ws, S.sh), v(ws, S.mean), v(ws, S.rstd), S.C, replicas=reps, rstride=rstride)
        else:
            ops.bn_coef_eval(bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps, v(ws, S.sc), v(ws, S.sh), v(ws, S.mean),
                             v(ws, S.rstd), S.C)

    @staticmethod
    def _parts(b):
        mods = list(b)
        i = 0
        conv_e = bn_e = None
        if b.cfg["expand"] != 1:
            conv_e, bn_e = mods[0], mods[1]
            i = 3
        return conv_e, bn_e, mods[i], mods[i + 1], mods[i + 3], mods[i + 4], mods[i + 5]      # dw, bn_d, se, conv_p, bn_p

    # ---- forward
    def forward(self, x, train):
        m, v, lb = self.model, self._v, (_LibF32(lib()) if self.dtype == torch.float32 else lib())
        u8 = x.dtype == torch.uint8             # decoded grey bytes (B,1,H,W): whitened + expanded on the GPU (cx_u8_to_nhwc8)
        if x.dim() != 4 or x.shape[1] != (1 if u8 else 3):
            raise RuntimeError("expected a (B,3,H,W) float input or a (B,1,H,W) uint8 image")
        B, _, H, W = x.shape
        self.bind(x.device)
        self.pack()
        ws = self.acquire(B, H, W)
        self.last_masks = {}
        self.n_forward = getattr(self, "n_forward", 0) + 1
        ws.step = self.n_forward
        if train:
            # the masks' step counter lives in device memory and is bumped by a kernel: a captured step (graph.py) draws new
            # Dropout / DropConnect masks at every replay, and the eager step draws the same ones
            if getattr(self, "step_dev", None) is None or self.step_dev.device != x.device:
                self.step_dev = torch.zeros(1, dtype=torch.int64, device=x.device)
            check(lb.cx_counter_add(ptr(self.step_dev), 1, stream_ptr()), "cx_counter_add")
        det = self.det and train
        if train and not det:
            z0, zn = self.fwd_zero
            ws.vec[z0:z0 + zn].zero_()
        st = (lambda s: v(ws, s)) if train else (lambda s: None)
        # statistics arguments of a convolution feeding BatchNorm S_, of a depthwise producer (sum, sq, stat_rows)
        csp = (lambda S_: dict(stat_sum=ws.slab[0], stat_sq=ws.slab[1], stat_det=True, stat_replicas=self.SLAB // S_.C, stat_rstride=S_.C)) \
            if det else (lambda S_: dict(stat_sum=st(S_.sum), stat_sq=st(S_.sq)))
        dsp = (lambda S_: (ptr(ws.slab[0]), ptr(ws.slab[1]), self._rows_cap(S_.C))) if det else \
            (lambda S_: (ptr(st(S_.sum)), ptr(st(S_.sq)), 0))
        sp = stream_ptr()
        # row scratch of the per-(image, channel) sums (free between a coefficient launch and the next statistics producer)
        rsc = (ptr(ws.slab[0]), self.SLAB) if self.det else (None, 0)
        S0 = self.bn[id(m.stem[1])]
        if u8:
            check(lb.cx_u8_to_nhwc8(ptr(x.contiguous()), ptr(ws.x8), B * H * W, 0.5330, 0.0349, sp), "cx_u8_to_nhwc8")
        else:
            check(lb.cx_nchw3_to_nhwc8(ptr(x.contiguous().float()), ptr(ws.x8), B, H, W, sp), "cx_nchw3_to_nhwc8")
        c0 = m.stem[0].out_channels
        rows = ops.conv_gemm(ws.x8, self.packed[self.stem_off:], ws.ys, N=c0, kh=3, kw=3, stride=2, pad=ws.stem_pad, **csp(S0))
        hs, wsz = ws.ys.shape[1:3]
        self._bn_coef(ws, m.stem[1], B * hs * wsz, train, rows)
        check(lb.cx_scale_act_bc(ptr(ws.ys), ptr(v(ws, S0.sc)), ptr(v(ws, S0.sh)), None, ptr(ws.x0), B, hs * wsz, c0, sp), "cx_scale_act_bc")
        xin = ws.x0
        for bi, b in enumerate(self.mb):
            c, t = b.cfg, ws.blk[bi]
            conv_e, bn_e, dw, bn_d, se, conv_p, bn_p = self._parts(b)
            (hi, wi), (ho, wo) = t["hin"], t["hout"]
            Sd, Sp = self.bn[id(bn_d)], self.bn[id(bn_p)]
            if conv_e is not None:
                Se = self.bn[id(bn_e)]
                rows = ops.conv_gemm(xin, self.w_fwd(conv_e), t["ye"], N=c["ce"], **csp(Se))
                self._bn_coef(ws, bn_e, B * hi * wi, train, rows)
                xdw, sc, sh = t["ye"], v(ws, Se.sc), v(ws, Se.sh)
            else:
                xdw, sc, sh = xin, None, None
            d1, d2, dcap = dsp(Sd)
            check(lb.cx_dwconv_fwd(ptr(xdw), ptr(dw.weight), ptr(sc), ptr(sh), ptr(t["yd"]), d1, d2, B, hi, wi,
                                   c["ce"], c["k"], c["stride"], t["pad"], dcap, sp), "cx_dwconv_fwd")
            self._bn_coef(ws, bn_d, B * ho * wo, train, lib().cx_last_stat_rows() if det else None)
            # squeeze + excite (efficientnet.py:69-73) in two launches: the excitation kernel adds the pool's split rows itself
            # (CHEXPERT_SE_FUSED=0: the separate pool / reduce / excitation launches, for A/B runs)
            if os.environ.get("CHEXPERT_SE_FUSED", "1") == "0":
                check(lb.cx_gap_affine_act(ptr(t["yd"]), ptr(v(ws, Sd.sc)), ptr(v(ws, Sd.sh)), ptr(t["pooled"]), B, ho * wo, c["ce"], 2,
                                           *rsc, sp), "cx_gap_affine_act")
                check(lb.cx_se_fwd(ptr(t["pooled"]), ptr(se[1].weight), ptr(se[1].bias), ptr(se[3].weight), ptr(se[3].bias), ptr(t["h1"]),
                                   ptr(t["s"]), B, c["ce"], t["R"], sp), "cx_se_fwd")
            else:
              check(lb.cx_gap_se_fwd(ptr(t["yd"]), ptr(v(ws, Sd.sc)), ptr(v(ws, Sd.sh)), ptr(t["pooled"]), ptr(se[1].weight), ptr(se[1].bias),
                                     ptr(se[3].weight), ptr(se[3].bias), ptr(t["h1"]), ptr(t["s"]), B, ho * wo, c["ce"], t["R"], 2, *rsc, sp),
                    "cx_gap_se_fwd")
            check(lb.cx_scale_act_bc(ptr(t["yd"]), ptr(v(ws, Sd.sc)), ptr(v(ws, Sd.sh)), ptr(t["s"]), ptr(t["u"]), B, ho * wo, c["ce"], sp),
                  "cx_scale_act_bc")
            rows = ops.conv_gemm(t["u"], self.w_fwd(conv_p), t["yp"], N=c["cout"], **csp(Sp))
            self._bn_coef(ws, bn_p, B * ho * wo, train, rows)
            # DropConnect (efficientnet.py:44-51, :100-101): train mode only, on blocks with a skip; the per-image mask / keep
            # probability is drawn by cx_dropout_mask from the model's step counter (reproducible, independent of torch's RNG)
            p_dc = list(b)[-1].p if (train and c["skip"] and isinstance(list(b)[-1], DropMarker)) else 0.0
            t["dc"] = None
            if p_dc > 0.0:
                t["dc"] = torch.empty(B, dtype=torch.float32, device=self.device)
                check(lb.cx_dropout_mask_dev(ptr(t["dc"]), B, 1.0 - p_dc, self._seed_base(bi), ptr(self.step_dev), sp), "cx_dropout_mask_dev")
                self.last_masks[self.mb_names[bi]] = t["dc"]
            check(lb.cx_affine2_out(ptr(t["yp"]), ptr(xin) if c["skip"] else None, ptr(v(ws, Sp.sc)), ptr(v(ws, self.ones)),
                                    ptr(v(ws, Sp.sh)), ptr(t["dc"]), ho * wo, ptr(t["out"]), B * ho * wo, c["cout"], sp), "cx_affine2_out")
            xin = t["out"]
        Sh = self.bn[id(m.head[1])]
        hl, wl = ws.hw_last
        rows = ops.conv_gemm(xin, self.w_fwd(m.head[0]), ws.yh, N=1280, **csp(Sh))
        self._bn_coef(ws, m.head[1], B * hl * wl, train, rows)
        check(lb.cx_gap_affine_act(ptr(ws.yh), ptr(v(ws, Sh.sc)), ptr(v(ws, Sh.sh)), ptr(ws.pooled), B, hl * wl, 1280, 2, *rsc, sp),
              "cx_gap_affine_act")
        # Dropout in front of the classifier (efficientnet.py:169-171), train mode only
        p_do = m.head[5].p if train else 0.0
        ws.drop = None
        fc_in = ws.pooled
        if p_do > 0.0:
            ws.drop = torch.empty(B, 1280, dtype=torch.float32, device=self.device)
            check(lb.cx_dropout_mask_dev(ptr(ws.drop), B * 1280, 1.0 - p_do, self._seed_base(len(self.mb)), ptr(self.step_dev), sp),
                  "cx_dropout_mask_dev")
            ws.pooled_d = torch.empty_like(ws.pooled)
            check(lb.cx_mul_f32(ptr(ws.pooled), ptr(ws.drop), ptr(ws.pooled_d), B * 1280, sp), "cx_mul_f32")
            fc_in = ws.pooled_d
            self.last_masks["head"] = ws.drop
        ws.fc_in = fc_in
        check(lb.cx_linear_fwd(ptr(fc_in), ptr(m.head[6].weight), ptr(m.head[6].bias), ptr(ws.logits), B, 1280, self.n_classes, sp),
              "cx_linear_fwd")
        if train:
            m._nbt_pending += 1
        return ws

    def _seed_base(self, idx):
        """Host part of the 64-bit seed of the mask of block `idx`: (model seed, block); the kernel adds the device-side count of
        training forwards * 1000003."""
        return (int(self.model.drop_seed) * 0x9E3779B1 + idx * 7919 + 12345) & 0xFFFFFFFFFFFFFFFF

    # ---- backward
    def _alloc_bwd(self, ws):
        if ws.bwd is not None:
            return
        dev, bf, B = self.device, self.dtype, ws.B
        bw = {"g": []}
        shapes = {}
        for t in ws.blk:
            sh = tuple(t["out"].shape)
            if sh not in shapes:
                shapes[sh] = torch.empty(sh, dtype=bf, device=dev)
            bw["g"].append(shapes[sh])
        bw["g0"] = torch.empty_like(ws.x0)
        bw["du"] = torch.empty(max(t["u"].numel() for t in ws.blk), dtype=bf, device=dev)
        bw["dzd"] = torch.empty_like(bw["du"])
        bw["dze"] = torch.empty(max([t["ye"].numel() for t in ws.blk if t["ye"] is not None] + [ws.yh.numel(), ws.ys.numel()]),
                                dtype=bf, device=dev)
        bw["gdc"] = torch.empty(max(t["out"].numel() for t in ws.blk), dtype=bf, device=dev)     # DropConnect-masked gradient
        ws.bwd = bw

    def backward(self, ws, dlogits):
        ops.set_det_wgrad(self.det)            # reproducible weight-gradient sums with the deterministic statistics
        deferred = self.det and ops.wgrad_defer_begin(self.device)
        try:
            self._backward(ws, dlogits)
            if deferred:
                ops.wgrad_defer_flush(self.device)
        finally:
            if deferred:
                ops.wgrad_defer_abort(self.device)

    def _backward(self, ws, dlogits):
        m, v, G, lb = self.model, self._v, self.G, (_LibF32(lib()) if self.dtype == torch.float32 else lib())
        det = self.det
        B = ws.B
        sp = stream_ptr()
        self._alloc_bwd(ws)
        bw = ws.bwd
        if not det:
            z0, zn = self.bwd_zero
            ws.vec[z0:z0 + zn].zero_()
        fresh = any(p.grad is None for p in self.params)
        if fresh:
            self.flat_grad.zero_()
        elif not all(p.grad.data_ptr() == gv.data_ptr() for p, gv in zip(self.params, self.grad_views)):
            raise RuntimeError("parameter .grad tensors were replaced; call zero_grad(set_to_none=True) first")
        red = self.reducer
        if red is not None:
            red.begin()
        done = (lambda p: red.ready(self.off_of[id(p)])) if red is not None else (lambda p: None)

        def bn_bwd(S, bn, count):
            """BatchNorm backward coefficients of `bn` from the sums its producer has just written (deterministic mode: the rows of
            the scratch pair, counted by cx_last_stat_rows)."""
            if det:
                ops.bn_bwd_coef(ws.slab[0], ws.slab[1], count, bn.weight, v(ws, S.mean), v(ws, S.rstd), G(bn.weight), G(bn.bias), None, None,
                                v(ws, S.pa), v(ws, S.pb), v(ws, S.pc), S.C, replicas=lib().cx_last_stat_rows(), rstride=S.C)
            else:
                ops.bn_bwd_coef(v(ws, S.S1), v(ws, S.S2), count, bn.weight, v(ws, S.mean), v(ws, S.rstd), G(bn.weight), G(bn.bias), None, None,
                                v(ws, S.pa), v(ws, S.pb), v(ws, S.pc), S.C)

        def ssp(S):      # (S1, S2, stat_rows) of an element-wise / depthwise producer of S's backward sums
            return (ptr(ws.slab[0]), ptr(ws.slab[1]), self._rows_cap(S.C)) if det else (ptr(v(ws, S.S1)), ptr(v(ws, S.S2)), 0)

        def dw_wgrad(*args):
            """cx_dwconv_wgrad with the slab workspace of ops (deferred sums when backward defers them)."""
            wsb, arena, dfr = ops._wgrad_ws(self.device)
            check(lb.cx_dwconv_wgrad(*args, ptr(wsb), 0 if wsb is None else wsb.numel(), sp), "cx_dwconv_wgrad")
            ops._wgrad_used(arena, dfr)
        # ---- head
        fc, Sh = m.head[6], self.bn[id(m.head[1])]
        hl, wl = ws.hw_last
        dpool = torch.empty(B, 1280, dtype=torch.float32, device=self.device)
        ops.head_bwd(dlogits, ws.fc_in, fc.weight, G(fc.weight), G(fc.bias), dpool)
        if ws.drop is not None:
            check(lb.cx_mul_f32(ptr(dpool), ptr(ws.drop), ptr(dpool), B * 1280, sp), "cx_mul_f32")
        dzh = bw["dze"][:ws.yh.numel()].view(ws.yh.shape)
        check(lb.cx_se_act_bwd(None, ptr(ws.yh), ptr(v(ws, Sh.sc)), ptr(v(ws, Sh.sh)), ptr(v(ws, Sh.mean)), ptr(v(ws, Sh.rstd)), None,
                               ptr(dpool), ptr(dzh), *ssp(Sh)[:2], B, hl * wl, 1280, ssp(Sh)[2], sp), "cx_se_act_bwd")
        bn_bwd(Sh, m.head[1], B * hl * wl)
        g = bw["g"][-1]
        xlast = ws.blk[-1]["out"]
        ops.conv_gemm(dzh, self.w_bwd(m.head[0]), g, N=xlast.shape[3], prologue=ops.PRO_AFFINE2, x2=ws.yh, pa=v(ws, Sh.pa), pb=v(ws, Sh.pb),
                      pc=v(ws, Sh.pc))
        ops.conv_wgrad(dzh, xlast, G(m.head[0].weight), g_prologue=ops.PRO_AFFINE2, g2=ws.yh, ga=v(ws, Sh.pa), gb=v(ws, Sh.pb),
                       gc=v(ws, Sh.pc))
        done(m.head[0].weight)
        # ---- blocks
        for bi in range(len(self.mb) - 1, -1, -1):
            b, t = self.mb[bi], ws.blk[bi]
            c = b.cfg
            conv_e, bn_e, dw, bn_d, se, conv_p, bn_p = self._parts(b)
            (hi, wi), (ho, wo) = t["hin"], t["hout"]
            Sd, Sp = self.bn[id(bn_d)], self.bn[id(bn_p)]
            xin = ws.blk[bi - 1]["out"] if bi > 0 else ws.x0
            g = bw["g"][bi]
            gin = g if c["skip"] else (bw["g"][bi - 1] if bi > 0 else bw["g0"])       # skip: dx accumulates into g itself
            ce, rows_o = c["ce"], B * ho * wo
            gsk = g                                   # gradient of the block output: the skip path takes it as it is
            if t.get("dc") is not None:               # the branch sees it through the DropConnect mask
                gb = bw["gdc"][:g.numel()].view(g.shape)
                check(lb.cx_scale_rows(ptr(g), ptr(t["dc"]), ho * wo, ptr(gb), rows_o, c["cout"], sp), "cx_scale_rows")
                g = gb
            check(lb.cx_bn_lin_bwd_stats(ptr(g), ptr(t["yp"]), ptr(v(ws, Sp.mean)), ptr(v(ws, Sp.rstd)), *ssp(Sp)[:2],
                                         rows_o, c["cout"], ssp(Sp)[2], sp), "cx_bn_lin_bwd_stats")
            bn_bwd(Sp, bn_p, rows_o)
            du = bw["du"][:rows_o * ce].view(B, ho, wo, ce)
            dzd = bw["dzd"][:rows_o * ce].view(B, ho, wo, ce)
            ops.conv_gemm(g, self.w_bwd(conv_p), du, N=ce, prologue=ops.PRO_AFFINE2, x2=t["yp"], pa=v(ws, Sp.pa), pb=v(ws, Sp.pb),
                          pc=v(ws, Sp.pc))
            ops.conv_wgrad(g, t["u"], G(conv_p.weight), g_prologue=ops.PRO_AFFINE2, g2=t["yp"], ga=v(ws, Sp.pa), gb=v(ws, Sp.pb),
                           gc=v(ws, Sp.pc))
            ds = torch.empty(B, ce, dtype=torch.float32, device=self.device)
            dpl = torch.empty(B, ce, dtype=torch.float32, device=self.device)
            # ds[b][c] = sum_hw du * swish(bn(yd)) and the two FCs' backward (efficientnet.py:69-73): the reduce kernel's split rows go
            # straight into the first FC pass (cx_se_bwd_fused: no launch of their own)
            wsb, arena, dfr = ops._wgrad_ws(self.device)
            if os.environ.get("CHEXPERT_SE_FUSED", "1") == "0":
                check(lb.cx_se_bwd_reduce(ptr(du), ptr(t["yd"]), ptr(v(ws, Sd.sc)), ptr(v(ws, Sd.sh)), ptr(ds), B, ho * wo, ce,
                                          *((ptr(ws.slab[0]), self.SLAB) if det else (None, 0)), sp), "cx_se_bwd_reduce")
                check(lb.cx_se_bwd(ptr(ds), ptr(t["s"]), ptr(t["h1"]), ptr(t["pooled"]), ptr(se[1].weight), ptr(se[3].weight),
                                   ptr(G(se[1].weight)), ptr(G(se[1].bias)), ptr(G(se[3].weight)), ptr(G(se[3].bias)), ptr(dpl), B, ce, t["R"],
                                   ptr(wsb), 0 if wsb is None else wsb.numel(), sp), "cx_se_bwd")
            else:
              check(lb.cx_se_bwd_fused(ptr(du), ptr(t["yd"]), ptr(v(ws, Sd.sc)), ptr(v(ws, Sd.sh)), ptr(ds), ptr(t["s"]), ptr(t["h1"]),
                                     ptr(t["pooled"]), ptr(se[1].weight), ptr(se[3].weight), ptr(G(se[1].weight)), ptr(G(se[1].bias)),
                                       ptr(G(se[3].weight)), ptr(G(se[3].bias)), ptr(dpl), B, ho * wo, ce, t["R"],
                                       *((ptr(ws.slab[0]), self.SLAB) if det else (None, 0)), ptr(wsb), 0 if wsb is None else wsb.numel(), sp),
                    "cx_se_bwd_fused")
            ops._wgrad_used(arena, dfr)
            check(lb.cx_se_act_bwd(ptr(du), ptr(t["yd"]), ptr(v(ws, Sd.sc)), ptr(v(ws, Sd.sh)), ptr(v(ws, Sd.mean)), ptr(v(ws, Sd.rstd)),
                                   ptr(t["s"]), ptr(dpl), ptr(dzd), *ssp(Sd)[:2], B, ho * wo, ce, ssp(Sd)[2], sp), "cx_se_act_bwd")
            bn_bwd(Sd, bn_d, rows_o)
            dargs = (ptr(dzd), ptr(t["yd"]), ptr(v(ws, Sd.pa)), ptr(v(ws, Sd.pb)), ptr(v(ws, Sd.pc)))
            if conv_e is not None:
                Se = self.bn[id(bn_e)]
                dze = bw["dze"][:B * hi * wi * ce].view(B, hi, wi, ce)
                check(lb.cx_dwconv_dgrad(*dargs, ptr(dw.weight), ptr(t["ye"]), ptr(v(ws, Se.sc)), ptr(v(ws, Se.sh)), ptr(v(ws, Se.mean)),
                                         ptr(v(ws, Se.rstd)), ptr(dze), *ssp(Se)[:2], B, hi, wi, ce, c["k"],
                                         c["stride"], t["pad"], 0, ssp(Se)[2], sp), "cx_dwconv_dgrad")
                bn_bwd(Se, bn_e, B * hi * wi)           # (before the next producer re-uses the statistic rows)
                dw_wgrad(*dargs, ptr(t["ye"]), ptr(v(ws, Se.sc)), ptr(v(ws, Se.sh)), ptr(G(dw.weight)), B, hi, wi, ce, c["k"],
                         c["stride"], t["pad"])
                ops.conv_gemm(dze, self.w_bwd(conv_e), gin, N=c["cin"], prologue=ops.PRO_AFFINE2, x2=t["ye"], pa=v(ws, Se.pa),
                              pb=v(ws, Se.pb), pc=v(ws, Se.pc), accumulate=c["skip"])
                ops.conv_wgrad(dze, xin, G(conv_e.weight), g_prologue=ops.PRO_AFFINE2, g2=t["ye"], ga=v(ws, Se.pa), gb=v(ws, Se.pb),
                               gc=v(ws, Se.pc))
            else:
                check(lb.cx_dwconv_dgrad(*dargs, ptr(dw.weight), ptr(xin), None, None, None, None, ptr(gin), None, None, B, hi, wi, ce,
                                         c["k"], c["stride"], t["pad"], int(c["skip"]), 0, sp), "cx_dwconv_dgrad")
                dw_wgrad(*dargs, ptr(xin), None, None, ptr(G(dw.weight)), B, hi, wi, ce, c["k"], c["stride"], t["pad"])
            done(list(b.parameters())[0])
        # ---- stem: x0 = swish(bn(ys))
        S0 = self.bn[id(m.stem[1])]
        hs, wsz = ws.ys.shape[1:3]
        c0 = m.stem[0].out_channels
        dzs = bw["dze"][:ws.ys.numel()].view(ws.ys.shape)
        check(lb.cx_se_act_bwd(ptr(bw["g0"]), ptr(ws.ys), ptr(v(ws, S0.sc)), ptr(v(ws, S0.sh)), ptr(v(ws, S0.mean)), ptr(v(ws, S0.rstd)), None,
                               None, ptr(dzs), *ssp(S0)[:2], B, hs * wsz, c0, ssp(S0)[2], sp), "cx_se_act_bwd")
        bn_bwd(S0, m.stem[1], B * hs * wsz)
        # (persistent: its address is part of the deferred slab-sum table, which must not change from step to step)
        if getattr(ws, "dw8", None) is None:
            ws.dw8 = torch.empty(c0, 8, 3, 3, dtype=torch.float32, device=self.device)
        dw8 = ws.dw8.zero_()
        ops.conv_wgrad(dzs, ws.x8, dw8, kh=3, kw=3, stride=2, pad=ws.stem_pad, g_prologue=ops.PRO_AFFINE2, g2=ws.ys, ga=v(ws, S0.pa),
                       gb=v(ws, S0.pb), gc=v(ws, S0.pc))
        ops.wgrad_defer_flush(self.device)       # the stem gradient is read back right here: run the deferred slab sums now
        G(m.stem[0].weight).view(c0, 3, 3, 3).add_(dw8[:, :3])
        if red is not None:
            red.finish()
        if fresh:
            for p, gv in zip(self.params, self.grad_views):
                p.grad = gv

    def enable_data_parallel(self, bucket_bytes=16 << 20, group=None):
        from ..parallel import GradReducer
        if self.flat_grad is None:
            raise RuntimeError("bind the engine first (run one forward)")
        self.reducer = GradReducer(self.flat_grad, bucket_bytes, group)
        # the deferred weight-gradient slab sums (ops.wgrad_defer_*) run before each bucket leaves, so that the bucket is final
        self.reducer.pre_launch = lambda: ops.wgrad_defer_flush(self.device, keep=True)


class _Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, model):
        if not model.training:
            raise NotImplementedError("autograd through the fused EfficientNet needs train() mode")
        ws = model._eng().forward(x, True)
        ctx.model, ctx.ws = model, ws
        return ws.logits.clone()

    @staticmethod
    def backward(ctx, dlogits):
        eng, ws = ctx.model._eng(), ctx.ws
        if ws is None:
            raise RuntimeError("backward through the fused EfficientNet can only run once per forward")
        eng.backward(ws, dlogits.contiguous().float())
        eng.release(ws)
        ctx.ws = None
        return None, None, None


class EfficientNet(nn.Module):
    def __init__(self, model_name, n_classes):
        super().__init__()
        assert model_name in SCALING_PARAMS.keys(), "Invalid model name."
        width, depth, _, dropout = SCALING_PARAMS[model_name]
        c_stem = _round_filters(32, width)
        self.stem = nn.Sequential(Conv2dParams(3, c_stem, 3, 2, bias=False), BatchNorm2dParams(c_stem), Marker())
        reps = []
        for (n, cin, cout, k, s, e) in _BASE:
            reps.append(MBConvBlockRepeat(int(math.ceil(depth * n)), _round_filters(cin, width), _round_filters(cout, width), k, s, e,
                                          0.25, 0.2))
        self.blocks = nn.Sequential(*reps)
        self.head = nn.Sequential(Conv2dParams(_round_filters(320, width), 1280, 1, bias=False), BatchNorm2dParams(1280), Marker(),
                                  PoolMarker(), Marker(), DropMarker(dropout), nn.Linear(1280, n_classes))
        for mod in self.modules():                                   # reset_parameters (:171-183)
            if isinstance(mod, nn.BatchNorm2d):
                mod.eps, mod.momentum = 1e-3, 0.01
            if isinstance(mod, nn.Conv2d):
                nn.init.kaiming_normal_(mod.weight, mode="fan_out", nonlinearity="conv2d")
                if mod.bias is not None:
                    nn.init.constant_(mod.bias, 0)
            if isinstance(mod, nn.Linear):
                nn.init.kaiming_uniform_(mod.weight, a=math.sqrt(5), mode="fan_in", nonlinearity="linear")
                nn.init.constant_(mod.bias, 0)
        self._nbt_pending = 0
        self._engine = None
        self.drop_seed = 0           # seed of the Dropout / DropConnect masks (with the forward counter and the block index)

    def _eng(self):
        if self._engine is None or self._engine.dtype != getattr(self, "_storage_dtype", torch.bfloat16):
            object.__setattr__(self, "_engine", _Engine(self))
        return self._engine

    def storage_dtype(self, dtype):
        """torch.bfloat16 (default) or torch.float32: the fp32 parity mode (same method as DenseNet.storage_dtype)."""
        dtype = {"bf16": torch.bfloat16, "fp32": torch.float32}.get(dtype, dtype)
        if dtype not in (torch.bfloat16, torch.float32):
            raise ValueError("storage dtype must be bf16 or fp32")
        object.__setattr__(self, "_storage_dtype", dtype)
        return self

    def state_dict(self, *args, **kwargs):
        if self._nbt_pending:
            for mod in self.modules():
                if isinstance(mod, nn.BatchNorm2d):
                    mod.num_batches_tracked += self._nbt_pending
            self._nbt_pending = 0
        return super().state_dict(*args, **kwargs)

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("chexpert_amd EfficientNet runs on the GPU only (hand-written HIP kernels); there is no CPU fallback")
        eng = self._eng()
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _Fn.apply(x, self.head[6].weight, self)
        if not self.training:
            from ..gradcam import hooked_eval_forward, hooks_registered
            if hooks_registered(self):                     # Grad-CAM hook protocol of the reference (chexpert.py:271-272)
                return hooked_eval_forward(self, x)
        ws = eng.forward(x, self.training)
        out = ws.logits.clone()
        eng.release(ws)
        return out

    def forward_backward(self, x, target):
        eng = self._eng()
        ws = eng.forward(x, self.training)
        B, n = ws.logits.shape
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        dl = torch.empty(B, n, dtype=torch.float32, device=x.device)
        ops.bce_fwd_bwd(ws.logits, target, loss, None, dl)
        eng.backward(ws, dl)
        logits = ws.logits.clone()
        eng.release(ws)
        return loss, logits


def construct_model(model_name, n_classes):
    """efficientnet.py:188-228: compound scaling of the B0 definition; the instance's class is named after the model."""
    assert model_name in SCALING_PARAMS.keys(), "Invalid model name."
    cls = type(model_name, (EfficientNet,), {})
    return cls(model_name, n_classes)
