"""ape/checkpoint/__init__.py:1-7 (hot-path part)"""
from ape_amd.checkpoint import DetectionCheckpointer  # noqa: F401

from .. import _overlay as _ov  # noqa: E402

_ov.extend(__path__, "checkpoint")
__getattr__ = _ov.lazy(globals(), {"FSDPDetectionCheckpointer": ".detection_checkpoint"})
