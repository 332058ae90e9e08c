"""ape/modeling/ape_deta/deformable_detr_segm.py"""
from ape_amd.modeling.ape_deta.deformable_detr_segm import DeformableDETRSegm  # noqa: F401
