"""Reference import path of the text-line net (train_settings/models/geotr/unet_model.py:4-37)."""
from dvd_amd.prestage import UNet  # noqa: F401
