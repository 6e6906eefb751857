"""Grad-CAM for the fused models, computing what /root/reference/chexpert.py:260-303 computes.

As executed by the reference: the legacy backward hook on the classifier
returns the gradient w.r.t. W^T, so after `.mean(1)` the channel weights are
`w[f] = (1/n_cls) * sum_b pooled[b,f]` -- independent of `cls_idx` and shared by the minibatch -- and the
hooked feature map is post-ReLU.  No backward pass is therefore needed: one eval-mode forward through the
HIP engine yields the block-4 buffer, norm5 scale/shift and the pooled features; two small kernels produce
the normalised, bilinearly up-sampled maps.  `hooks` / `cls_idx` are accepted for signature compatibility.
"""
import torch

from . import _lib as L


def _targets(model, eng, ws, dev):
    """(hooked module, final Linear, feature buffer, scale, shift, ReLU'd-in-place-after-the-hook?) of the reference's hook pairs."""
    if hasattr(model, "features"):                               # DenseNet: norm5 output, ReLU'd in place by the model (:514)
        nt = eng.slots["nt"][len(eng.blocks) - 1]
        return model.features.norm5, model.classifier, ws.buf[-1], ws.v(nt[0]), ws.v(nt[1]), True
    if hasattr(model, "_stages"):                                # ResNet / WideResNet: output of the last stage (already post-ReLU)
        return model._stages()[-1], model.fc, ws.blk[-1]["out"], None, None, False
    if hasattr(model, "head"):                                   # EfficientNet: head[1] BatchNorm output, before Swish
        S = eng.bn[id(model.head[1])]
        return model.head[1], model.head[-1], ws.yh, eng._v(ws, S.sc), eng._v(ws, S.sh), False
    raise RuntimeError("unknown model family")


def hooks_registered(model):
    try:
        eng = model._eng()
    except NotImplementedError:
        return False
    mods = [model.features.norm5, model.classifier] if hasattr(model, "features") else \
        ([model._stages()[-1], model.fc] if hasattr(model, "_stages") else [model.head[1], model.head[-1]])
    return any(len(m._forward_hooks) or len(m._backward_hooks) for m in mods)


class _HookedLinear(torch.autograd.Function):
    """Gives the fused eval forward an autograd edge at the final Linear so that hooks registered on it fire the way the
    reference's grad_cam expects (chexpert.py:268-283): a legacy `register_backward_hook` on nn.Linear receives
    grad_input = (grad_bias, grad_x, grad_W^T)."""

    @staticmethod
    def forward(ctx, anchor, logits, pooled, linear):
        ctx.pooled, ctx.linear = pooled, linear
        return logits.clone()

    @staticmethod
    def backward(ctx, dlogits):
        from . import ops
        lin, pooled = ctx.linear, ctx.pooled
        dl = dlogits.contiguous().float()
        dw = torch.zeros_like(lin.weight)
        db = torch.zeros(lin.out_features, device=dl.device)
        dp = torch.empty_like(pooled)
        ops.head_bwd(dl, pooled, lin.weight.detach(), dw, db, dp)
        for hook in list(lin._backward_hooks.values()):
            hook(lin, (db, dp, dw.t()), (dl,))
        return None, None, None, None


def hooked_eval_forward(model, x):
    """Eval forward through the HIP engine that honours `register_forward_hook` on the reference's Grad-CAM targets and
    `register_backward_hook` on the final Linear (chexpert.py:271-272 with the hook pairs of :468, :484, :498), so the
    reference's own `grad_cam(model, x, hooks)` works against the drop-in."""
    eng = model._eng()
    ws = eng.forward(x, False)
    try:
        hooked, lin, buf, sc, sh, relu_after = _targets(model, eng, ws, x.device)
        B, h, w, C = buf.shape
        feat = torch.empty(B, C, h, w, dtype=torch.float32, device=x.device)
        if buf.dtype != torch.bfloat16:
            raise NotImplementedError("hooks are served from the bf16 engine")

        def fill(relu):
            L.check(L.lib().cx_affine_to_f32_nchw(L.ptr(buf), L.ptr(sc), L.ptr(sh), int(relu), L.ptr(feat), B, h, w, C, buf.stride(2),
                                                  L.stream_ptr()), "cx_affine_to_f32_nchw")
        fill(False)
        for hook in list(hooked._forward_hooks.values()):
            hook(hooked, (None,), feat)
        if relu_after:
            fill(True)                       # F.relu(features, inplace=True) of the reference mutates the hooked tensor (:514)
        logits, pooled = ws.logits.clone(), ws.pooled.clone()
    finally:
        eng.release(ws)
    out = _HookedLinear.apply(lin.weight, logits, pooled, lin) if torch.is_grad_enabled() else logits
    for hook in list(lin._forward_hooks.values()):
        hook(lin, (pooled,), out)
    return out


@torch.no_grad()
def grad_cam(model, x, hooks=None, cls_idx=None):
    """Hook targets of the reference: DenseNet `features.norm5` / `classifier` (chexpert.py:468), ResNet `layer4` / `fc`
    (:484, :490), EfficientNet `head[1]` / `head[-1]` (:498).  The map tensor and the pooled input of the final Linear both
    exist in the engine's workspace after one eval forward."""
    if not x.is_cuda:
        raise RuntimeError("grad_cam runs on the GPU only")
    was_training = model.training
    model.eval()
    eng = model._eng()
    ws = eng.forward(x, False)
    try:
        dev = x.device
        if hasattr(model, "features"):                               # DenseNet: relu(norm5(block-4 buffer))
            buf = ws.buf[-1]
            nt = eng.slots["nt"][len(eng.blocks) - 1]
            sc, sh, inner = ws.v(nt[0]), ws.v(nt[1]), 1
        elif hasattr(model, "_stages"):                              # ResNet: output of the last stage (post-ReLU, no BN in between)
            buf = ws.blk[-1]["out"]
            C_ = buf.shape[3]
            sc, sh, inner = torch.ones(C_, device=dev), torch.zeros(C_, device=dev), 0
        elif hasattr(model, "head"):                                 # EfficientNet: head[1] BatchNorm output, before Swish
            buf = ws.yh
            S = eng.bn[id(model.head[1])]
            sc, sh, inner = eng._v(ws, S.sc), eng._v(ws, S.sh), 0
        else:
            raise RuntimeError("grad_cam: unknown model family")
        B, h, w, C = buf.shape
        n_cls = ws.logits.shape[1]
        wts = (ws.pooled.sum(0) / n_cls).contiguous()
        cam = torch.empty(B, h * w, dtype=torch.float32, device=dev)
        L.check(L.lib().cx_gradcam_map(L.ptr(buf), L.ptr(sc), L.ptr(sh), L.ptr(wts), L.ptr(cam), B, h * w, C, buf.stride(2), inner,
                                        L.stream_ptr()), "cx_gradcam_map")
        out = torch.empty(B, 1, x.shape[2], x.shape[3], dtype=torch.float32, device=dev)
        L.check(L.lib().cx_cam_norm_upsample(L.ptr(cam), L.ptr(out), B, h, w, x.shape[2], x.shape[3], L.stream_ptr()),
                "cx_cam_norm_upsample")
    finally:
        eng.release(ws)
        model.train(was_training)
    return out
