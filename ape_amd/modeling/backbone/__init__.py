from . import vit_eva02, vit_eva_clip  # noqa: F401
from .vit_eva_clip import SimpleFeaturePyramid, ViT  # noqa: F401
