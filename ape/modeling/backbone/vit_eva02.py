from ape_amd.modeling.backbone.vit_eva02 import SimpleFeaturePyramid, ViT  # noqa: F401
