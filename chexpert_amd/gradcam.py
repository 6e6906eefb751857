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
        elif hasattr(model, "layer4"):                               # ResNet: output of layer4 (post-ReLU, no BN in between)
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
