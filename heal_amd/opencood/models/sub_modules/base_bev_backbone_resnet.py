"""Import-path alias (reference: opencood/models/sub_modules/base_bev_backbone_resnet.py); the implementation lives in bev_blocks."""
from .bev_blocks import ResNetBEVBackbone  # noqa: F401
