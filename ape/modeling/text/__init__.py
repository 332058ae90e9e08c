from ape_amd.modeling.text import EVA02CLIP  # noqa: F401

from ... import _overlay as _ov  # noqa: E402

_ov.extend(__path__, "modeling", "text")
# the other text towers of ape/modeling/text/__init__.py:1-8 (not on the APE-*_D / APE-Ti path)
__getattr__ = _ov.lazy(globals(), {
    "Bert": ".bert_wrapper", "build_clip_text_encoder": ".clip_wrapper", "get_clip_embeddings": ".clip_wrapper", "EVA01CLIP": ".clip_wrapper_eva01",
    "build_openclip_text_encoder": ".clip_wrapper_open", "get_openclip_embeddings": ".clip_wrapper_open", "Llama2": ".llama2_wrapper",
    "T5_warpper": ".t5_wrapper", "TextModel": ".text_encoder"})
