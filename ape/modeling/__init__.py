"""ape/modeling/__init__.py (hot-path part)"""
from . import ape_deta, backbone, text  # noqa: F401
