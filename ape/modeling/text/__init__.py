from ape_amd.modeling.text import EVA02CLIP  # noqa: F401
