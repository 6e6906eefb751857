"""Import path of the reference (`from models.attn_aug_conv import DenseNet, ResNet, Bottleneck`, /root/reference/chexpert.py:26)
resolved to the MI355X implementation, so the reference's own model zoo (chexpert.py:461-502) builds the HIP-backed networks without
an edit.  (`models/` deliberately has no __init__.py: the reference's `models/` is a namespace package too, and tests/golden/
make_golden.py must still reach the REAL reference when /root/reference is first on sys.path.)

`models.attn_aug_conv` of the reference (AAConv2d, BasicBlock, Bottleneck, ResNet, WideResNet, _Transition, DenseNet) backed by
chexpert_amd.models -- same constructor signatures and state_dict keys."""
from chexpert_amd.models import AAConv2d, BasicBlock, Bottleneck, DenseNet, ResNet, WideResNet, _Transition  # noqa: F401
