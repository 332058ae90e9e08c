from ape_amd.layers.vision_language_align import StillClassifier, VisionLanguageAlign  # noqa: F401
