"""`models.attn_aug_conv` of the reference (AAConv2d, BasicBlock, Bottleneck, ResNet, WideResNet, _Transition, DenseNet) backed by
chexpert_amd.models -- same constructor signatures and state_dict keys."""
from chexpert_amd.models import AAConv2d, BasicBlock, Bottleneck, DenseNet, ResNet, WideResNet, _Transition  # noqa: F401
