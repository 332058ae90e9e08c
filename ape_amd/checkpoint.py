"""DetectionCheckpointer -- mirror of ape/checkpoint/detection_checkpoint.py:15-48 (the class demo/demo_lazy.py, tools/train_net.py
--eval-only and ape/engine/defaults.py:199 load `model_final.pth` with).  The reference subclasses detectron2's checkpointer and
changes one thing: checkpoint entries that are neither tensors nor numpy arrays are dropped with a warning instead of raising
(:21-48).  With detectron2 importable this is that subclass; without it, a minimal loader with the same call pattern
(`DetectionCheckpointer(model).load(path)`, `.save(name)`): torch.load, `{"model": state_dict, ...}` or a bare state dict,
numpy -> tensor, `module.` prefixes stripped, non-strict `load_state_dict` with the incompatible keys logged and returned.

The HIP-backed model keeps the reference's parameter names (state-dict contract tests), so the public APE checkpoints load
unchanged; weights are repacked to the kernels' layouts lazily after `load_state_dict` (ape_amd/packing.py)."""
import logging
import os

import numpy as np
import torch

try:                                                       # full environment
    from detectron2.checkpoint import DetectionCheckpointer as _D2Checkpointer
    if not hasattr(_D2Checkpointer, "_convert_ndarray_to_tensor"):
        raise ImportError("partial detectron2 stand-in")
except ImportError:
    _D2Checkpointer = None


def _convert(state_dict, logger):
    """numpy -> tensor in place; unsupported entries are dropped with a warning (detection_checkpoint.py:36-48)"""
    for k in list(state_dict.keys()):
        v = state_dict[k]
        if not isinstance(v, np.ndarray) and not isinstance(v, torch.Tensor):
            logger.warning("Unsupported type found in checkpoint! {}: {}".format(k, type(v)))
            state_dict.pop(k)
            continue
        if not isinstance(v, torch.Tensor):
            state_dict[k] = torch.from_numpy(v)


if _D2Checkpointer is not None:

    class DetectionCheckpointer(_D2Checkpointer):
        def _convert_ndarray_to_tensor(self, state_dict):
            _convert(state_dict, logging.getLogger(__name__))

else:

    class DetectionCheckpointer:
        def __init__(self, model, save_dir="", *, save_to_disk=None, **checkpointables):
            self.model = model
            self.save_dir = save_dir
            self.save_to_disk = bool(save_dir) if save_to_disk is None else save_to_disk
            self.checkpointables = dict(checkpointables)
            self.logger = logging.getLogger(__name__)

        def load(self, path, checkpointables=None):
            if not path:
                self.logger.info("No checkpoint found. Initializing model from scratch")
                return {}
            if not os.path.isfile(path):
                raise FileNotFoundError(f"Checkpoint {path} not found!")
            ckpt = torch.load(path, map_location="cpu", weights_only=False)
            if not (isinstance(ckpt, dict) and "model" in ckpt):
                ckpt = {"model": ckpt}
            sd = dict(ckpt["model"])
            _convert(sd, self.logger)
            sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
            own = self.model.state_dict()
            for k in list(sd):                                    # shape mismatches are skipped like fvcore's checkpointer does
                if k in own and tuple(own[k].shape) != tuple(sd[k].shape):
                    self.logger.warning(f"Skip loading parameter '{k}': checkpoint shape {tuple(sd[k].shape)} vs model {tuple(own[k].shape)}")
                    sd.pop(k)
            incompatible = self.model.load_state_dict(sd, strict=False)
            if incompatible.missing_keys:
                self.logger.warning("Some model parameters are not found in the checkpoint: " + ", ".join(incompatible.missing_keys[:20]))
            if incompatible.unexpected_keys:
                self.logger.warning("The checkpoint contains keys the model does not use: " + ", ".join(incompatible.unexpected_keys[:20]))
            for key in (self.checkpointables if checkpointables is None else checkpointables):
                if key in ckpt and key in self.checkpointables:
                    self.checkpointables[key].load_state_dict(ckpt.pop(key))
            ckpt["__incompatible__"] = incompatible
            return ckpt

        def save(self, name, **kwargs):
            if not self.save_dir or not self.save_to_disk:
                return
            data = {"model": self.model.state_dict()}
            for key, obj in self.checkpointables.items():
                data[key] = obj.state_dict()
            data.update(kwargs)
            basename = "{}.pth".format(name)
            os.makedirs(self.save_dir, exist_ok=True)
            torch.save(data, os.path.join(self.save_dir, basename))
            with open(os.path.join(self.save_dir, "last_checkpoint"), "w") as fh:
                fh.write(basename)

        def has_checkpoint(self):
            return os.path.exists(os.path.join(self.save_dir, "last_checkpoint"))

        def get_checkpoint_file(self):
            with open(os.path.join(self.save_dir, "last_checkpoint")) as fh:
                return os.path.join(self.save_dir, fh.read().strip())
