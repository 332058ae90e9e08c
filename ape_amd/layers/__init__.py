"""Mirror of ape/layers/__init__.py:1-8 (the operator/layer API of the hot path) on the HIP kernels."""
from .fuse_helper import BiAttentionBlock, BiMultiHeadAttention  # noqa: F401
from .multi_scale_deform_attn import (MultiScaleDeformableAttention,  # noqa: F401
                                      multi_scale_deformable_attn_pytorch)
from .vision_language_align import StillClassifier, VisionLanguageAlign  # noqa: F401
from .vision_language_fusion import VisionLanguageFusion  # noqa: F401
