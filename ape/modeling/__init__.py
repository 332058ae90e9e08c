"""ape/modeling/__init__.py (hot-path part)"""
from . import ape_deta, backbone, text  # noqa: F401

from .. import _overlay as _ov  # noqa: E402

_ov.extend(__path__, "modeling")
