"""Stub: the reference imports torchvision at module scope only (save_image, transforms)."""
from . import utils, transforms, models  # noqa: F401
