from .defaults import DefaultPredictor  # noqa: F401
