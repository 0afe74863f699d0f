"""Stub: cv2 is imported at module scope by the reference but unused on the path."""
