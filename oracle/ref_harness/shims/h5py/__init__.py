"""Stub: h5py is imported by the reference datasets package, unused on the path."""
