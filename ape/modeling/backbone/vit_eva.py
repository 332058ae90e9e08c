from ape_amd.modeling.backbone.vit_eva import SimpleFeaturePyramid, ViT  # noqa: F401
