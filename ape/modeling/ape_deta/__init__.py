"""ape/modeling/ape_deta/__init__.py:1-16 (the VL model family)"""
from ape_amd.modeling.ape_deta import (DeformableDETRSegmVL, DeformableDetrTransformerDecoderVL,  # noqa: F401
                                       DeformableDetrTransformerEncoderVL, DeformableDetrTransformerVL, SomeThing)
