from .vit_eva_clip import SimpleFeaturePyramid, ViT  # noqa: F401
