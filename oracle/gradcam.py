"""Oracle (test infrastructure): what /root/reference/chexpert.py:260-303 (`grad_cam`) computes.

As executed: the legacy backward hook on the classifier Linear yields
grad wrt W^T, so after `.mean(1)` the channel weights are class-independent and shared by the batch:
    w[f] = (1/n_cls) * sum_b pooled[b, f]        pooled = GAP(post-activation feature)
the hooked feature map is post-ReLU for DenseNet / ResNet (in-place relu on the hooked tensor),
    cam = relu(sum_f w[f] * feat[b, f]) -> per image (t - min) / (max - min + 1e-5)
    -> bilinear(align_corners=True) to the input size.
"""
import torch
import torch.nn.functional as F


def grad_cam_from_features(feat_post_act, n_classes, out_hw, pooled=None):
    """feat_post_act (B,C,h,w): the tensor the forward hook ends up holding.  `pooled` (B,C): the input of the hooked Linear
    when it is not the spatial mean of that tensor (EfficientNet, chexpert.py:498: the hook sits on head[1], the BatchNorm in
    front of Swish / pool, so the map is the pre-activation BN output and the weights come from GAP(swish(.)))."""
    if pooled is None:
        pooled = feat_post_act.mean((2, 3))                  # (B,C)
    w = pooled.sum(0) / n_classes                            # (C,)
    cam = F.relu((feat_post_act * w.view(1, -1, 1, 1)).sum(1, keepdim=True))
    mn = cam.amin((1, 2, 3), keepdim=True)
    mx = cam.amax((1, 2, 3), keepdim=True)
    cam = (cam - mn) / (mx - mn + 1e-5)
    return F.interpolate(cam, out_hw, mode="bilinear", align_corners=True)
