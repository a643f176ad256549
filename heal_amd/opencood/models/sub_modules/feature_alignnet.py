"""Import-path alias (reference: opencood/models/sub_modules/feature_alignnet.py); the implementation lives in bev_blocks."""
from .bev_blocks import AlignNet  # noqa: F401
