"""ape/modeling/ape_deta/__init__.py:1-16 (the VL model family of APE-*_D and the plain one of APE-L_A/B/C)"""
from ape_amd.modeling.ape_deta import (DeformableDETRSegm, DeformableDETRSegmVL, DeformableDetrTransformer,  # noqa: F401
                                       DeformableDetrTransformerDecoder, DeformableDetrTransformerDecoderVL,
                                       DeformableDetrTransformerEncoder, DeformableDetrTransformerEncoderVL,
                                       DeformableDetrTransformerVL, SomeThing)

from ... import _overlay as _ov  # noqa: E402

_ov.extend(__path__, "modeling", "ape_deta")
# training-side names of ape/modeling/ape_deta/__init__.py:1-16 (instantiated by the LazyConfigs even for inference): the reference's own
__getattr__ = _ov.lazy(globals(), {
    "DeformableCriterion": ".deformable_criterion", "Stage1Assigner": ".assigner", "Stage2Assigner": ".assigner",
    "DeformableDETR": ".deformable_detr"})
