"""`ape` -- the import paths of shenyunhang/APE's hot path, served by the MI355X implementation (`ape_amd`).

The reference's LazyConfigs name their classes by import path (`from ape.layers import VisionLanguageFusion`,
`from ape.modeling.ape_deta import DeformableDETRSegmVL, ...`, `from ape.modeling.backbone.vit_eva_clip import ViT`:
configs/LVISCOCOCOCOSTUFF_.../ape_deta_vitl_eva02_clip_vlf_lsj1024_cp_16x4_1080k.py:10-16,
configs/COCO_InstanceSegmentation/ape_deta/models/ape_deta_r50.py:9-19, configs/common/backbone/vitl_eva02_clip.py:7,
configs/common/backbone/vitt_eva02.py:7), and `demo/demo_lazy.py` / `tools/train_net.py --eval-only` instantiate whatever
those paths resolve to.  With this directory ahead of the reference checkout on sys.path they resolve to the HIP-backed
classes (same names, constructor kwargs, forward signatures, state-dict keys).  Only the inference hot path is mirrored;
everything else of the reference's `ape` package (data, evaluation, text towers, training) stays with the reference.
"""
import ape_amd as _impl

__version__ = _impl.__version__

from . import _overlay as _ov  # noqa: E402

_ov.extend(__path__)           # ape.data, ape.evaluation, ape.utils, ape.model_zoo, ... of a reference checkout stay importable
