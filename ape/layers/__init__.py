"""ape/layers/__init__.py:1-8"""
from ape_amd.layers import (BiAttentionBlock, BiMultiHeadAttention, MultiScaleDeformableAttention, StillClassifier,  # noqa: F401
                            VisionLanguageAlign, VisionLanguageFusion, multi_scale_deformable_attn_pytorch)

from .. import _overlay as _ov  # noqa: E402

_ov.extend(__path__, "layers")
__getattr__ = _ov.lazy(globals(), {"ZeroShotFC": ".zero_shot_fc"})
