from .densenet import DenseNet, densenet121  # noqa: F401
