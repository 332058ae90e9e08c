"""Result containers of the reference entry point.  `model([...])[i]["instances"]` is a detectron2 `Instances`
(ape/modeling/ape_deta/deformable_detr_segm_vl.py:857-872 via detector_postprocess) with fields pred_boxes (`Boxes`),
scores, pred_classes, pred_masks, moved to the CPU.  When detectron2 is importable its own classes are used; otherwise
these stand-ins offer the part of their API that the reference's consumers touch (demo/demo_lazy.py:189-198,
ape/evaluation/*: `.pred_boxes.tensor`, `.scores`, `.pred_classes`, `.pred_masks`, `.has()`, `.to()`, `len()`, indexing)."""
import torch

try:                                                     # full environment: the real thing
    from detectron2.structures import Boxes, Instances   # noqa: F401
    if not (hasattr(Instances, "set") and hasattr(Instances, "has") and hasattr(Boxes, "nonempty")):
        raise ImportError("a partial detectron2 stand-in is registered (test shims): use the local containers")
except ImportError:

    class Boxes:
        def __init__(self, tensor):
            self.tensor = tensor.reshape(-1, 4).float()

        def __len__(self):
            return self.tensor.shape[0]

        def __getitem__(self, item):
            t = self.tensor[item]
            return Boxes(t.reshape(-1, 4))

        def to(self, *args, **kw):
            return Boxes(self.tensor.to(*args, **kw))

        def clone(self):
            return Boxes(self.tensor.clone())

        def area(self):
            b = self.tensor
            return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

        def clip(self, box_size):
            h, w = box_size
            self.tensor[:, 0::2].clamp_(0, w)
            self.tensor[:, 1::2].clamp_(0, h)

        def nonempty(self, threshold=0.0):
            b = self.tensor
            return ((b[:, 2] - b[:, 0]) > threshold) & ((b[:, 3] - b[:, 1]) > threshold)

        @property
        def device(self):
            return self.tensor.device

        def __repr__(self):
            return f"Boxes({self.tensor})"

    class Instances:
        def __init__(self, image_size, **fields):
            object.__setattr__(self, "_image_size", tuple(image_size))
            object.__setattr__(self, "_fields", {})
            for k, v in fields.items():
                self.set(k, v)

        @property
        def image_size(self):
            return self._image_size

        def __setattr__(self, name, val):
            if name.startswith("_"):
                object.__setattr__(self, name, val)
            else:
                self.set(name, val)

        def __getattr__(self, name):
            if name == "_fields" or name not in self._fields:
                raise AttributeError(f"Cannot find field '{name}' in the given Instances!")
            return self._fields[name]

        def set(self, name, value):
            if len(self._fields):
                assert len(self) == len(value), f"Adding a field of length {len(value)} to a Instances of length {len(self)}"
            self._fields[name] = value

        def has(self, name):
            return name in self._fields

        def remove(self, name):
            del self._fields[name]

        def get(self, name):
            return self._fields[name]

        def get_fields(self):
            return self._fields

        def to(self, *args, **kw):
            ret = Instances(self._image_size)
            for k, v in self._fields.items():
                ret.set(k, v.to(*args, **kw) if hasattr(v, "to") else v)
            return ret

        def __getitem__(self, item):
            if isinstance(item, int):
                item = slice(item, None, len(self)) if item >= 0 else slice(item + len(self), None, len(self))
            ret = Instances(self._image_size)
            for k, v in self._fields.items():
                ret.set(k, v[item])
            return ret

        def __len__(self):
            for v in self._fields.values():
                return len(v)
            return 0

        def __repr__(self):
            return f"Instances(num_instances={len(self)}, image_size={self._image_size}, fields={list(self._fields)})"


def make_instances(image_size, boxes, scores, classes, masks=None, **extra):
    inst = Instances(tuple(image_size))
    inst.pred_boxes = Boxes(boxes)
    inst.scores = scores
    inst.pred_classes = classes
    if masks is not None:
        inst.pred_masks = masks
    for k, v in extra.items():
        inst.set(k, v)
    return inst
