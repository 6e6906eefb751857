"""`models.efficientnet` of the reference (`construct_model(model_name, n_classes)`, SCALING_PARAMS) backed by chexpert_amd.models."""
from chexpert_amd.models.efficientnet import SCALING_PARAMS, construct_model  # noqa: F401
