"""Import-path alias (reference: opencood/models/sub_modules/downsample_conv.py); the implementation lives in bev_blocks."""
from .bev_blocks import DownsampleConv, DoubleConv  # noqa: F401
