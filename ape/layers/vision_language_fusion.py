from ape_amd.layers.vision_language_fusion import VisionLanguageFusion  # noqa: F401
