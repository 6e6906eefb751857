"""Import paths of the reference (`from models.efficientnet import construct_model`, `from models.attn_aug_conv import DenseNet,
ResNet, Bottleneck`: /root/reference/chexpert.py:25-26) resolved to the MI355X implementation, so that the reference's own
`chexpert.py` model zoo (:461-502) builds the HIP-backed networks without an edit."""
