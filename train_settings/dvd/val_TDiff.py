"""Sampling plug-in (reference train_settings/dvd/val_TDiff.py:40-116): exports run(settings)."""
from dvd_amd.val_TDiff import run  # noqa: F401
