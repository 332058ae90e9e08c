from ape_amd.engine import DefaultPredictor  # noqa: F401
