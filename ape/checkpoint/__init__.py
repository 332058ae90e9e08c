"""ape/checkpoint/__init__.py:1-7 (hot-path part)"""
from ape_amd.checkpoint import DetectionCheckpointer  # noqa: F401
