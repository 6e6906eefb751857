from .densenet import DenseNet, densenet121  # noqa: F401
from .resnet import Bottleneck, ResNet, resnet152  # noqa: F401
from .efficientnet import construct_model  # noqa: F401
