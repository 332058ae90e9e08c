from ape_amd.modeling.backbone.vit_eva_clip import SimpleFeaturePyramid, ViT  # noqa: F401
