from .densenet import AAConv2d, DenseNet, _Transition, densenet121  # noqa: F401
from .resnet import BasicBlock, Bottleneck, ResNet, WideResNet, resnet152  # noqa: F401
from .efficientnet import construct_model  # noqa: F401
