"""Mirror of ape/modeling/ape_deta/__init__.py:1-16: the VL (APE-*_D) model family and the plain one (APE-L_A/B/C)."""
from .ape_deta import SomeThing  # noqa: F401
from .deformable_detr_segm import (DeformableDETRSegm, DeformableDetrTransformer, DeformableDetrTransformerDecoder,  # noqa: F401
                                   DeformableDetrTransformerEncoder)
from .deformable_detr_segm_vl import DeformableDETRSegmVL  # noqa: F401
from .deformable_transformer_vl import (DeformableDetrTransformerDecoderVL,  # noqa: F401
                                        DeformableDetrTransformerEncoderVL, DeformableDetrTransformerVL)
