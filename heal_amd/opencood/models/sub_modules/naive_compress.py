"""Import-path alias (reference: opencood/models/sub_modules/naive_compress.py); the implementation lives in bev_blocks."""
from .bev_blocks import NaiveCompressor  # noqa: F401
