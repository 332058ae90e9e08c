"""ape/layers/__init__.py:1-8"""
from ape_amd.layers import (BiAttentionBlock, BiMultiHeadAttention, MultiScaleDeformableAttention, StillClassifier,  # noqa: F401
                            VisionLanguageAlign, VisionLanguageFusion, multi_scale_deformable_attn_pytorch)
