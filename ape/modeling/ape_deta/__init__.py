"""ape/modeling/ape_deta/__init__.py:1-16 (the VL model family)"""
from ape_amd.modeling.ape_deta import (DeformableDETRSegmVL, DeformableDetrTransformerDecoderVL,  # noqa: F401
                                       DeformableDetrTransformerEncoderVL, DeformableDetrTransformerVL, SomeThing)

from ... import _overlay as _ov  # noqa: E402

_ov.extend(__path__, "modeling", "ape_deta")
# training-side names of ape/modeling/ape_deta/__init__.py:1-16 (instantiated by the LazyConfigs even for inference): the reference's own
__getattr__ = _ov.lazy(globals(), {
    "DeformableCriterion": ".deformable_criterion", "Stage1Assigner": ".assigner", "Stage2Assigner": ".assigner",
    "DeformableDETR": ".deformable_detr", "DeformableDETRSegm": ".deformable_detr_segm", "DeformableDetrTransformer": ".deformable_transformer",
    "DeformableDetrTransformerDecoder": ".deformable_transformer", "DeformableDetrTransformerEncoder": ".deformable_transformer"})
