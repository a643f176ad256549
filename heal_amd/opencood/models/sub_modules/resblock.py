"""Import-path alias (reference: opencood/models/sub_modules/resblock.py); the implementation lives in bev_blocks."""
from .bev_blocks import ResNetModified, Bottleneck, BasicBlock  # noqa: F401
