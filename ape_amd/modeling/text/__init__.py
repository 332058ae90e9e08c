"""ape/modeling/text (hot-path part): the EVA-02-CLIP text tower the APE configs instantiate as `model_language`"""
from .clip_wrapper_eva02 import EVA02CLIP  # noqa: F401
