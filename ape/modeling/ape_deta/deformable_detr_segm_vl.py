from ape_amd.modeling.ape_deta.deformable_detr_segm_vl import DeformableDETRSegmVL  # noqa: F401
