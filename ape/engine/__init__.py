from .defaults import DefaultPredictor  # noqa: F401

from .. import _overlay as _ov  # noqa: E402

_ov.extend(__path__, "engine")                    # ape.engine.train_loop stays the reference's
