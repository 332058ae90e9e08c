from . import vit_eva02, vit_eva_clip  # noqa: F401
