"""Reference import path of the pre-stage document-mask nets (train_settings/models/geotr/geotr_core.py:745-1112)."""
from dvd_amd.prestage import GeoTr_Seg_Inf, Seg, U2NETP, reload_model, reload_segmodel  # noqa: F401
