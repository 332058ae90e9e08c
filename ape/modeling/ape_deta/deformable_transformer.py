"""ape/modeling/ape_deta/deformable_transformer.py"""
from ape_amd.modeling.ape_deta.deformable_detr_segm import (DeformableDetrTransformer, DeformableDetrTransformerDecoder,  # noqa: F401
                                                             DeformableDetrTransformerEncoder)
