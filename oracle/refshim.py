"""Run the REFERENCE's own source files (never copied) from /root/reference under third-party shims.

TEST INFRASTRUCTURE (oracle), usable only where /root/reference exists (this container): it validates the
restatement in oracle/ape_oracle.py and generates tests/golden/*.  detectron2 / detrex / torchvision / timm /
fvcore / cv2 are not installable here, so minimal stand-ins built from oracle/thirdparty.py are registered in
sys.modules, then the reference files are loaded by path with importlib (the `ape` package __init__ files are
not executed -- they import the whole training stack).
"""
import copy
import importlib.util
import os
import sys
import types
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import thirdparty as tp

REF = os.environ.get("APE_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF, "ape"))


# --------------------------------------------------------------------------------- detectron2 stand-ins
class ShapeSpec:
    def __init__(self, channels=None, height=None, width=None, stride=None):
        self.channels, self.height, self.width, self.stride = channels, height, width, stride


class D2LayerNorm(nn.Module):
    def __init__(self, normalized_shape, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps = eps

    def forward(self, x):
        return tp.layer_norm_2d(x, self.weight, self.bias, self.eps)


def get_norm(norm, out_channels):
    if norm is None:
        return None
    if isinstance(norm, str):
        if len(norm) == 0:
            return None
        norm = {"GN": lambda c: nn.GroupNorm(32, c), "LN": lambda c: D2LayerNorm(c)}[norm]
    return norm(out_channels)


class Conv2d(nn.Conv2d):
    def __init__(self, *args, **kwargs):
        norm = kwargs.pop("norm", None)
        activation = kwargs.pop("activation", None)
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation

    def forward(self, x):
        x = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


class CNNBlockBase(nn.Module):
    def __init__(self, in_channels, out_channels, stride):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride


class Backbone(nn.Module):
    def forward(self):
        raise NotImplementedError

    @property
    def size_divisibility(self):
        return 0

    @property
    def padding_constraints(self):
        return {}

    def output_shape(self):
        return {name: ShapeSpec(channels=self._out_feature_channels[name], stride=self._out_feature_strides[name])
                for name in self._out_features}


class LastLevelMaxPool(nn.Module):
    def __init__(self):
        super().__init__()
        self.num_levels = 1
        self.in_feature = "p5"

    def forward(self, x):
        return [tp.last_level_max_pool(x)]


def _assert_strides_are_log2_contiguous(strides):
    for i, stride in enumerate(strides[1:], 1):
        assert stride == 2 * strides[i - 1]


class ImageList:
    def __init__(self, tensor, image_sizes):
        self.tensor, self.image_sizes = tensor, image_sizes

    @staticmethod
    def from_tensors(tensors, size_divisibility=0, pad_value=0.0, padding_constraints=None):
        sq = (padding_constraints or {}).get("square_size", 0)
        assert len(tensors) == 1, "oracle runs batch 1 like the reference's evaluation"
        t, hw = tp.pad_to_square(tensors[0], sq, pad_value)
        return ImageList(t[None], [hw])


class Boxes:
    def __init__(self, tensor):
        self.tensor = tensor.reshape(-1, 4).float() if tensor.numel() == 0 else tensor.float()

    def clip(self, box_size):
        h, w = box_size
        x1 = self.tensor[:, 0].clamp(min=0, max=w)
        y1 = self.tensor[:, 1].clamp(min=0, max=h)
        x2 = self.tensor[:, 2].clamp(min=0, max=w)
        y2 = self.tensor[:, 3].clamp(min=0, max=h)
        self.tensor = torch.stack((x1, y1, x2, y2), dim=-1)

    def __len__(self):
        return self.tensor.shape[0]

    def to(self, *a, **k):
        return Boxes(self.tensor.to(*a, **k))


class BitMasks:
    def __init__(self, tensor):
        self.tensor = tensor.to(torch.bool)

    def crop_and_resize(self, boxes, mask_size):
        return tp.bitmasks_crop_and_resize(self.tensor, boxes, mask_size)


class Instances:
    def __init__(self, image_size, **kwargs):
        object.__setattr__(self, "_image_size", image_size)
        object.__setattr__(self, "_fields", {})
        for k, v in kwargs.items():
            self._fields[k] = v

    @property
    def image_size(self):
        return self._image_size

    def __setattr__(self, name, val):
        if name.startswith("_"):
            object.__setattr__(self, name, val)
        else:
            self._fields[name] = val

    def __getattr__(self, name):
        if name == "_fields" or name not in self._fields:
            raise AttributeError(name)
        return self._fields[name]

    def has(self, name):
        return name in self._fields

    def get_fields(self):
        return self._fields

    def to(self, *a, **k):
        return self


def detector_postprocess(results, output_height, output_width, mask_threshold=0.5):
    masks = results.pred_masks[:, 0] if results.has("pred_masks") else None
    b, s, c, m, _ = tp.detector_postprocess(results.pred_boxes.tensor, results.scores, results.pred_classes, masks,
                                            results.image_size, output_height, output_width, mask_threshold)
    out = Instances((output_height, output_width))
    out.pred_boxes, out.scores, out.pred_classes = Boxes(b), s, c
    if m is not None:
        out.pred_masks = m
    return out


def sem_seg_postprocess(result, img_size, output_height, output_width):
    result = result[:, : img_size[0], : img_size[1]].expand(1, -1, -1, -1)
    return F.interpolate(result, size=(output_height, output_width), mode="bilinear", align_corners=False)[0]


def move_device_like(src, dst):
    return src.to(dst.device)


def retry_if_cuda_oom(func):
    return func


class _Metadata(dict):
    def __init__(self, name):
        super().__init__()
        self.name = name

    def get(self, k, d=None):
        return dict.get(self, k, d)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


METADATA = {}     # name -> dict(thing_classes=..., stuff_classes=...): what MetadataCatalog.get(name) carries in a run


class _MetadataCatalog:
    def get(self, name):
        m = _Metadata(name)
        m.update(METADATA.get(name, {}))
        return m


# --------------------------------------------------------------------------------- detrex stand-ins
class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = F.relu(layer(x)) if i < self.num_layers - 1 else layer(x)
        return x


class FFN(nn.Module):
    def __init__(self, embed_dim=256, feedforward_dim=1024, output_dim=None, num_fcs=2, activation=None,
                 ffn_drop=0.0, fc_bias=True, add_identity=True):
        super().__init__()
        output_dim = embed_dim if output_dim is None else output_dim
        act = activation if activation is not None else nn.ReLU(inplace=True)
        layers = []
        in_channels = embed_dim
        for _ in range(num_fcs - 1):
            layers.append(nn.Sequential(nn.Linear(in_channels, feedforward_dim, bias=fc_bias), act, nn.Dropout(ffn_drop)))
            in_channels = feedforward_dim
        layers.append(nn.Linear(feedforward_dim, output_dim, bias=fc_bias))
        layers.append(nn.Dropout(ffn_drop))
        self.layers = nn.Sequential(*layers)
        self.add_identity = add_identity
        self.embed_dim = embed_dim

    def forward(self, x, identity=None):
        out = self.layers(x)
        if not self.add_identity:
            return out
        if identity is None:
            identity = x
        return identity + out


class MultiheadAttention(nn.Module):
    def __init__(self, embed_dim, num_heads, attn_drop=0.0, proj_drop=0.0, batch_first=False, **kwargs):
        super().__init__()
        self.embed_dim, self.num_heads, self.batch_first = embed_dim, num_heads, batch_first
        self.attn = nn.MultiheadAttention(embed_dim=embed_dim, num_heads=num_heads, dropout=attn_drop,
                                          batch_first=batch_first, **kwargs)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None, attn_mask=None,
                key_padding_mask=None, **kwargs):
        if key is None:
            key = query
        if value is None:
            value = key
        if identity is None:
            identity = query
        if key_pos is None and query_pos is not None:
            if query_pos.shape == key.shape:
                key_pos = query_pos
            else:
                warnings.warn("position encoding of key is missing in MultiheadAttention.")
        if query_pos is not None:
            query = query + query_pos
        if key_pos is not None:
            key = key + key_pos
        out = self.attn(query=query, key=key, value=value, attn_mask=attn_mask, key_padding_mask=key_padding_mask)[0]
        return identity + self.proj_drop(out)


class BaseTransformerLayer(nn.Module):
    def __init__(self, attn, ffn, norm, operation_order=None):
        super().__init__()
        assert set(operation_order).issubset({"self_attn", "norm", "cross_attn", "ffn"})
        num_attn = operation_order.count("self_attn") + operation_order.count("cross_attn")
        if isinstance(attn, nn.Module):
            attn = [copy.deepcopy(attn) for _ in range(num_attn)]
        assert len(attn) == num_attn
        self.num_attn = num_attn
        self.operation_order = operation_order
        self.pre_norm = operation_order[0] == "norm"
        self.attentions = nn.ModuleList()
        index = 0
        for op in operation_order:
            if op in ("self_attn", "cross_attn"):
                self.attentions.append(attn[index])
                index += 1
        self.embed_dim = self.attentions[0].embed_dim
        self.ffns = nn.ModuleList(copy.deepcopy(ffn) for _ in range(operation_order.count("ffn")))
        self.norms = nn.ModuleList(copy.deepcopy(norm) for _ in range(operation_order.count("norm")))

    def forward(self, query, key=None, value=None, query_pos=None, key_pos=None, attn_masks=None,
                query_key_padding_mask=None, key_padding_mask=None, **kwargs):
        norm_index = attn_index = ffn_index = 0
        identity = query
        if attn_masks is None:
            attn_masks = [None for _ in range(self.num_attn)]
        elif isinstance(attn_masks, torch.Tensor):
            attn_masks = [copy.deepcopy(attn_masks) for _ in range(self.num_attn)]
        for layer in self.operation_order:
            if layer == "self_attn":
                temp_key = temp_value = query
                query = self.attentions[attn_index](
                    query, temp_key, temp_value, identity if self.pre_norm else None, query_pos=query_pos,
                    key_pos=query_pos, attn_mask=attn_masks[attn_index], key_padding_mask=query_key_padding_mask, **kwargs)
                attn_index += 1
                identity = query
            elif layer == "norm":
                query = self.norms[norm_index](query)
                norm_index += 1
            elif layer == "cross_attn":
                query = self.attentions[attn_index](
                    query, key, value, identity if self.pre_norm else None, query_pos=query_pos, key_pos=key_pos,
                    attn_mask=attn_masks[attn_index], key_padding_mask=key_padding_mask, **kwargs)
                attn_index += 1
                identity = query
            elif layer == "ffn":
                query = self.ffns[ffn_index](query, identity if self.pre_norm else None)
                ffn_index += 1
        return query


class TransformerLayerSequence(nn.Module):
    def __init__(self, transformer_layers=None, num_layers=None):
        super().__init__()
        self.num_layers = num_layers
        self.layers = nn.ModuleList()
        if isinstance(transformer_layers, nn.Module):
            for _ in range(num_layers):
                self.layers.append(copy.deepcopy(transformer_layers))
        else:
            assert isinstance(transformer_layers, list) and len(transformer_layers) == num_layers
            for layer in transformer_layers:
                self.layers.append(layer)


class PositionEmbeddingSine(nn.Module):
    def __init__(self, num_pos_feats=64, temperature=10000, scale=2 * 3.141592653589793, eps=1e-6, offset=0.0,
                 normalize=False):
        super().__init__()
        self.kw = dict(num_pos_feats=num_pos_feats, temperature=temperature, scale=scale, eps=eps, offset=offset,
                       normalize=normalize)

    def forward(self, mask):
        return tp.position_embedding_sine(mask, **self.kw)


class ConvNormAct(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 norm_layer=None, activation=None, **kwargs):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                              dilation=dilation, groups=groups, bias=bias)
        self.norm = norm_layer
        self.activation = activation

    def forward(self, x):
        x = self.conv(x)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


class ChannelMapper(nn.Module):
    def __init__(self, input_shapes, in_features, out_channels, kernel_size=3, stride=1, bias=True, groups=1,
                 dilation=1, norm_layer=None, activation=None, num_outs=None, **kwargs):
        super().__init__()
        self.extra_convs = None
        chans = [input_shapes[f].channels for f in in_features]
        if num_outs is None:
            num_outs = len(input_shapes)
        self.convs = nn.ModuleList(
            ConvNormAct(c, out_channels, kernel_size=kernel_size, stride=stride, padding=(kernel_size - 1) // 2, bias=bias,
                        groups=groups, dilation=dilation, norm_layer=copy.deepcopy(norm_layer),
                        activation=copy.deepcopy(activation)) for c in chans)
        assert num_outs == len(chans), "oracle shim: extra convs not needed by APE-L_D (5 inputs, 5 outputs)"
        self.input_shapes, self.in_features, self.out_channels = input_shapes, in_features, out_channels

    def forward(self, inputs):
        assert len(inputs) == len(self.convs)
        return tuple(self.convs[i](inputs[self.in_features[i]]) for i in range(len(inputs)))


class DropPath(nn.Module):
    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        assert not self.training or not self.drop_prob
        return x


# --------------------------------------------------------------------------------- installation
def _mod(name, **attrs):
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        m.__path__ = []  # behave like a package
        sys.modules[name] = m
        if "." in name:
            parent, child = name.rsplit(".", 1)
            setattr(_mod(parent), child, m)
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


class TimmMlp(nn.Module):
    """stand-in for timm.models.layers.Mlp as vit_eva.py:258 uses it (published definition: fc1 -> act -> drop -> fc2 -> drop, bias on
    both linears; parity unpinned -- timm is not installed here)"""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


def _noop(*a, **k):
    return None


_installed = False


def install():
    """Register the stand-in modules and load the reference's hot-path files.  Returns the `ape` namespace."""
    global _installed
    if _installed:
        return sys.modules["ape"]
    assert available(), f"reference not found under {REF}"
    _mod("detectron2")
    _mod("detectron2.layers", Conv2d=Conv2d, get_norm=get_norm, CNNBlockBase=CNNBlockBase, ShapeSpec=ShapeSpec,
         move_device_like=move_device_like, batched_nms=lambda b, s, i, t: tp.batched_nms(b.float(), s, i, t))
    _mod("detectron2.modeling", GeneralizedRCNN=object, detector_postprocess=detector_postprocess)
    _mod("detectron2.modeling.backbone", Backbone=Backbone)
    _mod("detectron2.modeling.backbone.fpn", LastLevelMaxPool=LastLevelMaxPool,
         _assert_strides_are_log2_contiguous=_assert_strides_are_log2_contiguous)
    _mod("detectron2.modeling.postprocessing", detector_postprocess=detector_postprocess,
         sem_seg_postprocess=sem_seg_postprocess)
    _mod("detectron2.modeling.meta_arch")
    _mod("detectron2.modeling.meta_arch.panoptic_fpn", combine_semantic_and_instance_outputs=_noop)
    _mod("detectron2.modeling.roi_heads")
    _mod("detectron2.modeling.roi_heads.fast_rcnn", fast_rcnn_inference=_noop)
    _mod("detectron2.structures", BitMasks=BitMasks, Boxes=Boxes, ImageList=ImageList, Instances=Instances)
    _mod("detectron2.utils")
    _mod("detectron2.utils.events", get_event_storage=_noop)
    _mod("detectron2.utils.memory", retry_if_cuda_oom=retry_if_cuda_oom)
    _mod("detectron2.data")
    _mod("detectron2.data.detection_utils", convert_image_to_rgb=_noop)
    _mod("detectron2.data.catalog", MetadataCatalog=_MetadataCatalog())
    _mod("detrex")
    _mod("detrex.layers", FFN=FFN, BaseTransformerLayer=BaseTransformerLayer, MultiheadAttention=MultiheadAttention,
         TransformerLayerSequence=TransformerLayerSequence, MLP=MLP, PositionEmbeddingSine=PositionEmbeddingSine,
         box_cxcywh_to_xyxy=tp.box_cxcywh_to_xyxy, box_xyxy_to_cxcywh=tp.box_xyxy_to_cxcywh)
    _mod("detrex.utils", inverse_sigmoid=tp.inverse_sigmoid)
    _mod("detrex.modeling")
    _mod("detrex.modeling.neck", ChannelMapper=ChannelMapper)
    _mod("torchvision")
    _mod("torchvision.ops")
    _mod("torchvision.ops.boxes", batched_nms=tp.batched_nms, nms=tp.nms)
    _mod("timm")
    _mod("timm.models")
    _mod("timm.models.layers", DropPath=DropPath, Mlp=TimmMlp, trunc_normal_=nn.init.trunc_normal_)
    _mod("fairscale")
    _mod("fairscale.nn")
    _mod("fairscale.nn.checkpoint", checkpoint_wrapper=lambda m, *a, **k: m)     # activation checkpointing: identity at inference
    _mod("fvcore")
    _mod("fvcore.nn")
    _mod("fvcore.nn.weight_init", c2_xavier_fill=_noop, c2_msra_fill=_noop)
    if "cv2" not in sys.modules:
        _mod("cv2")

    # the reference package skeleton (no __init__ executed)
    _mod("ape")
    _mod("ape._C")  # otherwise multi_scale_deform_attn.py:415-423 swaps the class for an ImportError dummy
    _mod("ape.layers")
    _mod("ape.modeling")
    _mod("ape.modeling.backbone")
    _mod("ape.modeling.ape_deta")
    _mod("ape.modeling.text", utils=types.ModuleType("ape.modeling.text.utils"))
    sys.modules["ape.modeling.text.utils"] = sys.modules["ape.modeling.text"].utils

    def load(modname, relpath):
        spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
        m = importlib.util.module_from_spec(spec)
        sys.modules[modname] = m
        parent, child = modname.rsplit(".", 1)
        setattr(sys.modules[parent], child, m)
        spec.loader.exec_module(m)
        return m

    L = sys.modules["ape.layers"]
    m = load("ape.layers.multi_scale_deform_attn", "ape/layers/multi_scale_deform_attn.py")
    L.MultiScaleDeformableAttention = m.MultiScaleDeformableAttention
    L.multi_scale_deformable_attn_pytorch = m.multi_scale_deformable_attn_pytorch
    m = load("ape.layers.fuse_helper", "ape/layers/fuse_helper.py")
    L.BiAttentionBlock, L.BiMultiHeadAttention = m.BiAttentionBlock, m.BiMultiHeadAttention
    m = load("ape.layers.vision_language_fusion", "ape/layers/vision_language_fusion.py")
    L.VisionLanguageFusion = m.VisionLanguageFusion
    m = load("ape.layers.vision_language_align", "ape/layers/vision_language_align.py")
    L.VisionLanguageAlign, L.StillClassifier = m.VisionLanguageAlign, m.StillClassifier
    L.ZeroShotFC = object  # not selected by APE-L_D (deformable_detr.py:86,145)

    load("ape.modeling.backbone.utils_eva02", "ape/modeling/backbone/utils_eva02.py")
    load("ape.modeling.backbone.vit_eva_clip", "ape/modeling/backbone/vit_eva_clip.py")
    load("ape.modeling.backbone.vit_eva02", "ape/modeling/backbone/vit_eva02.py")
    load("ape.modeling.backbone.utils_eva", "ape/modeling/backbone/utils_eva.py")
    load("ape.modeling.backbone.vit_eva", "ape/modeling/backbone/vit_eva.py")

    A = sys.modules["ape.modeling.ape_deta"]
    _mod("ape.modeling.ape_deta.segmentation", MaskHeadSmallConv=object, MHAttentionMap=object)
    load("ape.modeling.ape_deta.deformable_transformer_vl", "ape/modeling/ape_deta/deformable_transformer_vl.py")
    load("ape.modeling.ape_deta.fast_rcnn", "ape/modeling/ape_deta/fast_rcnn.py")
    load("ape.modeling.ape_deta.deformable_detr", "ape/modeling/ape_deta/deformable_detr.py")
    load("ape.modeling.ape_deta.deformable_detr_segm_vl", "ape/modeling/ape_deta/deformable_detr_segm_vl.py")
    # the plain (non-VL) family of APE-L_A/B/C: scripts/eval_APE-L_A.sh
    load("ape.modeling.ape_deta.deformable_transformer", "ape/modeling/ape_deta/deformable_transformer.py")
    load("ape.modeling.ape_deta.deformable_detr_segm", "ape/modeling/ape_deta/deformable_detr_segm.py")
    load("ape.modeling.ape_deta.ape_deta", "ape/modeling/ape_deta/ape_deta.py")
    for sub in ("deformable_transformer_vl", "deformable_transformer", "deformable_detr", "deformable_detr_segm_vl", "deformable_detr_segm", "ape_deta"):
        mod = sys.modules[f"ape.modeling.ape_deta.{sub}"]
        for name in dir(mod):
            obj = getattr(mod, name)
            if isinstance(obj, type) and obj.__module__ == mod.__name__:
                setattr(A, name, obj)
    _installed = True
    return sys.modules["ape"]


def install_text():
    """Load the reference's CLIP text tower files (ape/modeling/text/eva02_clip/{utils,rope,transformer,tokenizer}.py) under
    stand-ins for timm / torchvision / ftfy.  Returns (transformer module, tokenizer module)."""
    install()
    if "ape.modeling.text.eva02_clip.transformer" in sys.modules:
        return sys.modules["ape.modeling.text.eva02_clip.transformer"], sys.modules["ape.modeling.text.eva02_clip.tokenizer"]
    _mod("timm.models.layers", trunc_normal_=nn.init.trunc_normal_)
    _mod("timm.layers", trunc_normal_=nn.init.trunc_normal_)
    _mod("torchvision.ops.misc", FrozenBatchNorm2d=nn.BatchNorm2d)
    if "ftfy" not in sys.modules:
        try:
            import ftfy  # noqa: F401
        except ImportError:
            _mod("ftfy", fix_text=lambda s: s)          # identity on clean text (the tests use ASCII / NFC strings)
    _mod("ape.modeling.text.eva02_clip")

    def load(name):
        modname = f"ape.modeling.text.eva02_clip.{name}"
        spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, "ape/modeling/text/eva02_clip", name + ".py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[modname] = m
        setattr(sys.modules["ape.modeling.text.eva02_clip"], name, m)
        spec.loader.exec_module(m)
        return m

    load("utils")
    load("rope")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tr = load("transformer")
    tok = load("tokenizer")
    return tr, tok
