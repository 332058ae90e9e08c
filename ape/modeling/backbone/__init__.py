from . import vit_eva, vit_eva02, vit_eva_clip  # noqa: F401

from ... import _overlay as _ov  # noqa: E402

_ov.extend(__path__, "modeling", "backbone")      # vit_eva.py (ViT-e / ViT-g) etc. stay the reference's
