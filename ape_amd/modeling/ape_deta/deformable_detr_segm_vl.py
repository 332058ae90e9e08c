"""APE meta-architecture (inference forward) on the HIP kernels.

Mirror of ape/modeling/ape_deta/deformable_detr_segm_vl.py (DeformableDETRSegmVL.forward :166-726,
maskdino_mask_features :728-750, inference :759-810, preprocess_image :846-855, _postprocess_instance :857-872) and
of the constructor in ape/modeling/ape_deta/deformable_detr.py:52-296 (heads :100-215): same class name, constructor
kwargs and state-dict keys.  Name / phrase / expression prompts (single-token and dense multi-token fusion) and the
instance / semantic / panoptic tails are implemented; training branches raise NotImplementedError.

Per image (batch 1, like the reference's evaluation) the whole forward is a fixed sequence of HIP kernel launches
on the current stream with no host synchronisation until the final device->host copy of the detections.
"""
import math
import time
from types import SimpleNamespace
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...layers import VisionLanguageAlign
from ...packing import attach_cache, f32, pack_matrix
from ...stagetap import tap
from ...structures import make_instances
from . import geometry as G
from ._containers import MLP


class _ConvGN(nn.Module):
    """detectron2 Conv2d(bias=False[, norm=GN]) parameter holder: weight [Cout,Cin,k,k] (+ norm.{weight,bias})"""

    def __init__(self, cin, cout, k, norm=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        nn.init.kaiming_uniform_(self.weight, a=1)
        if norm:
            self.norm = nn.GroupNorm(32, cout)


class DeformableDETRSegmVL(nn.Module):
    def __init__(self, backbone, position_embedding, neck, transformer, embed_dim, num_classes, num_queries, criterion,
                 pixel_mean, pixel_std, aux_loss=True, with_box_refine=False, as_two_stage=False,
                 select_box_nums_for_evaluation=100, select_box_nums_for_evaluation_list: list = None,
                 input_format: Optional[str] = None, vis_period: int = 0, output_dir: Optional[str] = None,
                 dataset_names: List[str] = [], dataset_metas: List[str] = [], dataset_prompts: List[str] = None,
                 embed_dim_language: int = 512, text_feature_batch_repeat: bool = True, text_feature_bank: bool = False,
                 text_feature_bank_reset: bool = False, text_feature_bank_random_size: bool = False,
                 text_feature_reduce_type: str = "last", text_feature_reduce_before_fusion: bool = True,
                 expression_cumulative_gt_class: bool = True, test_nms_thresh: float = 0.7, test_score_thresh: float = 0.0,
                 last_class_embed_use_mlp: bool = False, openset_classifier: str = "VisionLanguageAlign",
                 # DeformableDETRSegmVL's own kwargs (deformable_detr_segm_vl.py:63-88)
                 instance_on: bool = True, semantic_on: bool = False, panoptic_on: bool = False, freeze_detr=False,
                 input_shapes=[], mask_in_features=[], mask_encode_level=0, stuff_dataset_learn_thing: bool = True,
                 stuff_prob_thing: float = -1.0, name_prompt_fusion_type: str = "none", name_prompt_fusion_text: bool = None,
                 test_mask_on: bool = True, semantic_post_nms: bool = True, panoptic_post_nms: bool = True,
                 aux_mask: bool = False, panoptic_configs: dict = None):
        super().__init__()
        assert with_box_refine and as_two_stage and openset_classifier == "VisionLanguageAlign" and not aux_mask, \
            "ape_amd implements the two-stage, box-refine, VisionLanguageAlign configuration of the APE-*_D models"
        self.backbone, self.position_embedding, self.neck, self.transformer = backbone, position_embedding, neck, transformer
        self.num_queries, self.num_classes = num_queries, num_classes
        self.aux_loss, self.with_box_refine, self.as_two_stage = aux_loss, with_box_refine, as_two_stage
        self.criterion = nn.ModuleList([c for c in (criterion or []) if isinstance(c, nn.Module)])
        num_pred = transformer.decoder.num_layers + 1
        # heads (deformable_detr.py:100-200): class_embed[i] / bbox_embed[i] are SHARED with transformer.decoder
        self.class_embed = nn.ModuleList([VisionLanguageAlign(embed_dim, embed_dim_language) for _ in range(num_pred)])
        self.bbox_embed = nn.ModuleList([MLP(embed_dim, embed_dim, 4, 3) for _ in range(num_pred)])
        self.transformer.decoder.bbox_embed = self.bbox_embed
        self.transformer.decoder.class_embed = self.class_embed
        self.transformer.decoder.class_embed[-1] = nn.Linear(embed_dim, 1)   # class-agnostic encoder classifier (:178-179)
        if self.transformer.proposal_ambiguous:
            n = self.transformer.proposal_ambiguous
            assert n == 1, "ape_amd implements proposal_ambiguous in {0, 1}"
            self.transformer.decoder.bbox_embed_ambiguous = nn.ModuleList([MLP(embed_dim, embed_dim, 4, 3) for _ in range(n)])
            self.transformer.decoder.class_embed_ambiguous = nn.ModuleList([nn.Linear(embed_dim, 1) for _ in range(n)])
        # proposal_ambiguous = 0 (the plain transformer of APE-L_A/B/C): no ambiguous copies (deformable_detr.py:181-200)
        self.select_box_nums_for_evaluation = select_box_nums_for_evaluation
        self.select_box_nums_for_evaluation_list = select_box_nums_for_evaluation_list
        self.test_topk_per_image = select_box_nums_for_evaluation
        self.test_nms_thresh, self.test_score_thresh = test_nms_thresh, test_score_thresh
        self.input_format, self.vis_period, self.output_dir = input_format, vis_period, output_dir
        self.register_buffer("pixel_mean", torch.tensor(pixel_mean).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(pixel_std).view(-1, 1, 1), False)
        self._mean, self._std = tuple(float(v) for v in pixel_mean), tuple(float(v) for v in pixel_std)
        self.dataset_names, self.dataset_prompts = dataset_names, dataset_prompts
        self.dataset_metas = [dataset_metas] if isinstance(dataset_metas, str) else list(dataset_metas)
        # MetadataCatalog stand-in (deformable_detr.py:238-272): entries of dataset_metas may be names (resolved through
        # detectron2 when it is importable) or dicts carrying thing_classes / stuff_classes themselves
        self.metadata_list = [self._resolve_metadata(m) for m in self.dataset_metas]
        self.dataset_entities = [self._entity_of(m) for m in self.metadata_list]
        self.stuff_prob_thing, self.semantic_post_nms = stuff_prob_thing, semantic_post_nms
        self.panoptic_post_nms = panoptic_post_nms
        self.panoptic_configs = panoptic_configs or {"prob": 0.1, "pano_temp": 0.06, "transform_eval": True,
                                                     "object_mask_threshold": 0.01, "overlap_threshold": 0.4}   # (:80-86)
        self.dataset_name_to_idx = {k: i for i, k in enumerate(self.dataset_names)}
        self.class_names = {}                       # dataset name -> list[str]; filled by set_class_names (MetadataCatalog stand-in)
        self.eval_dataset_id, self.eval_dataset_entity = -1, ""
        self.text_feature_bank, self.text_feature_bank_reset = text_feature_bank, text_feature_bank_reset
        self.text_feature_reduce_before_fusion = text_feature_reduce_before_fusion
        # rows of features_phrase_bank (ape_deta/deformable_detr.py:281-291): max criterion.num_classes
        self.phrase_bank_size = max([int(getattr(c, "num_classes", 0)) for c in self.criterion] + [0]) or 256
        self.embed_dim_language = embed_dim_language
        if text_feature_bank:
            # (ape_deta/deformable_detr.py:281-291) one bank per criterion / dataset, not part of the state dict
            self.register_buffer("features_phrase_bank",
                                 torch.zeros((max(len(self.criterion), 1), self.phrase_bank_size, embed_dim_language)), False)
        self.instance_on, self.semantic_on, self.panoptic_on = instance_on, semantic_on, panoptic_on
        self.input_shapes, self.mask_in_features, self.mask_encode_level = input_shapes, mask_in_features, mask_encode_level
        assert len(mask_in_features) == 1 and mask_encode_level == 0
        hidden = self.transformer.embed_dim
        cin = getattr(input_shapes[mask_in_features[0]], "channels", input_shapes[mask_in_features[0]])
        self.lateral_conv = _ConvGN(cin, hidden, 1)
        self.output_conv = _ConvGN(hidden, hidden, 3)
        self.mask_conv = _ConvGN(hidden, hidden, 1, norm=False)
        self.mask_embed = MLP(hidden, hidden, hidden, 3)
        self.test_mask_on = test_mask_on
        self.name_prompt_fusion_type = name_prompt_fusion_type
        self.name_prompt_fusion_text = name_prompt_fusion_text      # per-dataset flags or None
        if name_prompt_fusion_type == "zero":
            self.name_prompt_fusion_feature = nn.Parameter(torch.zeros(1, 1, embed_dim_language), requires_grad=False)
        elif name_prompt_fusion_type == "learnable":
            self.name_prompt_fusion_feature = nn.Parameter(torch.randn(1, 1, embed_dim_language))
        elif getattr(self.transformer.encoder, "vl_layers", None) is not None:
            raise NotImplementedError("ape_amd: with fusion layers name_prompt_fusion_type must be 'zero' or 'learnable' (fusion needs a token)")
        # "none" on the plain encoder (no fusion layers): there is no fusion token and no parameter for one
        self.model_language = None
        self.compute_dtype = torch.bfloat16
        self._geo, self._text = {}, {}
        self.preprocess_time = self.backbone_time = self.transformer_time = self.postprocess_time = 0.0
        attach_cache(self)
        def _drop_caches(m, incompatible):
            m._geo.clear()
            m._text.clear()

        self.register_load_state_dict_post_hook(_drop_caches)

    # ------------------------------------------------------------------ configuration helpers
    @property
    def device(self):
        return self.pixel_mean.device

    def _apply(self, fn, *args, **kwargs):
        """`model.half()` / `model.to(torch.float16)` -- how the reference evaluates (tools/train_net.py:642) -- selects the IEEE-half
        flavour of the kernels, `.to(torch.bfloat16)` the bf16 one: a change of the PARAMETER dtype to a 16-bit type is a request for that
        arithmetic.  (`.float()` leaves the compute dtype alone: fp32 parameters are the normal state of every flavour.)"""
        p0 = next(self.parameters(), None)
        before = p0.dtype if p0 is not None else None
        out = super()._apply(fn, *args, **kwargs)
        p1 = next(self.parameters(), None)
        if p1 is not None and p1.dtype != before and p1.dtype in (torch.float16, torch.bfloat16):
            self.set_compute_dtype(p1.dtype)
        return out

    def set_compute_dtype(self, dt):
        """torch.bfloat16 (BASELINE's dtype, the default), torch.float16 (the reference's evaluation dtype: same kernels on IEEE half)
        or torch.float32 (exact-math validation mode)"""
        self.compute_dtype = dt
        for m in self.modules():
            if hasattr(m, "compute_dtype") and m is not self and not isinstance(getattr(type(m), "compute_dtype", None), property):
                m.compute_dtype = dt
        return self

    def set_model_language(self, model_language):
        self.model_language = model_language

    @staticmethod
    def _resolve_metadata(meta):
        if isinstance(meta, dict):
            return dict(meta)
        try:
            from detectron2.data.catalog import MetadataCatalog
            m = MetadataCatalog.get(meta)
            return {k: m.get(k) for k in ("thing_classes", "stuff_classes", "thing_dataset_id_to_contiguous_id") if m.get(k) is not None} | {"name": meta}
        except ImportError:
            return {"name": str(meta)}

    @staticmethod
    def _entity_of(meta):
        """deformable_detr.py:247-262"""
        if "stuffonly" in meta.get("name", ""):
            meta.pop("thing_classes", None)
        thing, stuff = meta.get("thing_classes"), meta.get("stuff_classes")
        if thing is not None and stuff is not None:
            return "thing+stuff"
        return "stuff" if (thing is None and stuff is not None) else "thing"

    def set_metadata(self, dataset_id, **meta):
        """attach thing_classes / stuff_classes / thing_dataset_id_to_contiguous_id to a dataset slot"""
        while len(self.metadata_list) <= dataset_id:
            self.metadata_list.append({"name": ""})
            self.dataset_entities.append("thing")
        self.metadata_list[dataset_id].update(meta)
        self.dataset_entities[dataset_id] = self._entity_of(self.metadata_list[dataset_id])

    def set_eval_dataset(self, dataset_name):
        """deformable_detr.py:524-532"""
        for d in self.dataset_names:
            if sum([dd in dataset_name for dd in d.split("+")]):
                self.eval_dataset_id = self.dataset_name_to_idx[d]
                self.eval_dataset_entity = self.dataset_entities[self.eval_dataset_id] if self.eval_dataset_id < len(self.dataset_entities) else ""
                break
        else:
            self.eval_dataset_id = -1
            self.eval_dataset_entity = ""

    def set_class_names(self, dataset_id, names):
        self.class_names[dataset_id] = list(names)

    # ------------------------------------------------------------------ packing
    def packed(self, dt):
        def build(dt):
            def conv(m):
                w = m.weight.detach().float()
                return pack_matrix(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1), dt)
            neck = [(pack_matrix(c.conv.weight.detach().reshape(c.conv.weight.shape[0], -1), dt), f32(c.conv.bias),
                     f32(c.norm.weight), f32(c.norm.bias), c.norm.num_groups, c.norm.eps) for c in self.neck.convs] if self.neck is not None else None
            gn = lambda m: (f32(m.norm.weight), f32(m.norm.bias), m.norm.num_groups, m.norm.eps)  # noqa: E731
            return dict(neck=neck, lat=(conv(self.lateral_conv),) + gn(self.lateral_conv),
                        outc=(conv(self.output_conv),) + gn(self.output_conv), maskc=conv(self.mask_conv))
        return self._pack.get(self, dt, build)

    def pos_cfg(self):
        pe = self.position_embedding
        return dict(num_pos_feats=pe.num_pos_feats, temperature=pe.temperature, normalize=pe.normalize, offset=pe.offset,
                    eps=pe.eps, scale=pe.scale)

    def geometry(self, image_size, level_shapes):
        key = (tuple(image_size), tuple(level_shapes), str(self.device))
        if key in self._geo:
            self._geo[key] = self._geo.pop(key)           # most recently used last
        elif len(self._geo) >= 16:                        # ~140 MB of constants per size at 1024^2: keep the 16 latest
            self._geo.pop(next(iter(self._geo)))
        if key not in self._geo:
            pe = self.position_embedding
            cfg = dict(num_pos_feats=pe.num_pos_feats, temperature=pe.temperature, normalize=pe.normalize, offset=pe.offset,
                       eps=pe.eps, scale=pe.scale)
            square = self.backbone.padding_constraints.get("square_size", 0)
            self._geo[key] = G.build_geometry(square, image_size, level_shapes, self.device, cfg)
        return self._geo[key]

    # ------------------------------------------------------------------ text side
    def text_features(self, batched_input):
        """(text bank [K, D_l] fp32 on the device, names, prompt mode) -- deformable_detr_segm_vl.py:204-259, 283-303.
        prompt "name": the bank only feeds the classifier; "phrase" / "expression": it is fused in the encoder."""
        if "text_features" in batched_input:                       # pre-computed CLIP features (the broadcast payload)
            prompt = batched_input.get("prompt", "name")
            if prompt not in ("name", "phrase", "expression"):
                raise ValueError(f"ape_amd: with pre-computed text_features the prompt must be name/phrase/expression, got {prompt!r}")
            return batched_input["text_features"].to(self.device).float(), None, prompt
        if self.eval_dataset_id >= 0:
            prompt = self.dataset_prompts[self.eval_dataset_id]
            names = self.class_names.get(self.eval_dataset_id)
            if names is None:
                names = self.text_list_of(self.metadata_list[self.eval_dataset_id], self.dataset_entities[self.eval_dataset_id])
            cache = True
        else:
            prompt = batched_input.get("prompt", "name")
            if prompt == "text":
                names = [x.strip() for x in batched_input["text_prompt"].split(",") if len(x.strip()) > 0]
                prompt = "phrase" if any(x.count(" ") >= 1 for x in names) else "name"
                cache = False
            else:
                names, cache = sum((self.class_names[i] for i in sorted(self.class_names)), [])[:1203], True
                if prompt in ("phrase", "expression"):       # (:283-290) phrases / expressions carried by the input
                    names = list(batched_input["expressions"]) if prompt == "expression" else list(batched_input["phrases"])
        if prompt not in ("name", "phrase", "expression"):
            raise ValueError(f"ape_amd: unknown prompt mode {prompt!r}")
        if self.model_language is None:
            raise RuntimeError("no text tower attached (set_model_language) and no 'text_features' in the input")
        if prompt == "name":
            out = self.model_language.forward_text(names, cache=cache)
        else:
            out = self.model_language.forward_text(names)
            if not self.text_feature_reduce_before_fusion or "last_hidden_state_eot" not in out:
                raise NotImplementedError("ape_amd: un-reduced text tokens (text_feature_reduce_before_fusion=False) are not implemented")
        return out["last_hidden_state_eot"].to(self.device).float(), names, prompt

    @staticmethod
    def text_list_of(meta, entity):
        """get_text_list (deformable_detr_segm_vl.py:1229-1248): the class-name vocabulary of a dataset's metadata"""
        thing, stuff = list(meta.get("thing_classes") or []), list(meta.get("stuff_classes") or [])
        overlap = len(thing) > 0 and len(stuff) > 0 and (set(thing) <= set(stuff) or set(stuff) <= set(thing))
        if entity == "thing+stuff" and stuff[0] == "things":
            return thing + stuff[1:]
        if entity == "thing+stuff" and overlap:
            return thing if len(thing) > len(stuff) else stuff
        if entity == "thing+stuff":
            return thing + stuff
        return stuff if entity == "stuff" else thing

    def _bank_classes(self, dataset_id):
        """criterion[dataset_id].num_classes of the reference (:314-327); index -1 = the last criterion like Python indexing"""
        if len(self.criterion) == 0:
            return self.phrase_bank_size
        return int(getattr(self.criterion[dataset_id], "num_classes", self.phrase_bank_size))

    def fusion_tokens(self, text_feats, prompt):
        """the language tokens the encoder fuses with: the zero / learnable token in name mode (:349-352); in phrase /
        expression mode the reduced text bank, extended by the phrase bank (:304-327):
          * text_feature_bank_reset: zero rows up to the bank size (negatives carry no text);
          * persistent bank (the APE-*_D default) while evaluating a dataset: the rows kept from earlier images act as
            negatives, then the bank is overwritten with the current tokens (stateful, in place);
          * otherwise (free-text prompts, dataset_id = -1, no reset): just the current tokens."""
        if getattr(self.transformer.encoder, "vl_layers", None) is None:
            return None                                                     # plain encoder (APE-L_A/B/C): nothing is fused
        if prompt == "name":
            nft = self.name_prompt_fusion_text
            if nft is not None and nft[self.eval_dataset_id]:          # (:343-347) ODinW-style: fuse the class names themselves
                return text_feats.float().contiguous()
            return self.name_prompt_fusion_feature.detach().float().reshape(1, -1)
        text_feats = text_feats.float()
        K, D = text_feats.shape
        ds = self.eval_dataset_id
        if self.text_feature_bank and self.text_feature_bank_reset:
            n = max(K, self._bank_classes(ds))
            text_feats = torch.cat([text_feats, text_feats.new_zeros((n - K, D))], 0)
        elif self.text_feature_bank and 0 <= ds < len(self.metadata_list):
            ncls = self._bank_classes(ds)
            row = ds if ds < self.features_phrase_bank.shape[0] else -1
            text_feats = torch.cat([text_feats, self.features_phrase_bank[row].to(text_feats.device)], 0)[: max(K, ncls)]
            self.features_phrase_bank[row, :ncls] = text_feats[:ncls].to(self.features_phrase_bank.device)
        return text_feats.contiguous()

    def class_tokens(self, feats, lvl, dt):
        """per-vocabulary constants of the last-level classifier, cached by tensor IDENTITY: an entry keeps a reference to
        its bank (which also pins the address) and hits only for that very tensor object at the same version -- a freed
        bank whose address the allocator hands to another vocabulary can never match."""
        key = (id(feats), lvl, dt)
        ent = self._text.get(key)
        if ent is not None and ent[0] is feats and ent[1] == feats._version:
            return ent[2]
        if len(self._text) > 16:
            self._text.clear()
        val = self.class_embed[lvl].text_side(feats, dt)
        self._text[key] = (feats, feats._version, val)
        return val

    # ------------------------------------------------------------------ the hot path, one image
    def mask_prompt_tokens(self, mask_prompt, geo):
        """deformable_detr_segm_vl.py:394-414: the [h, w] prompt mask, zero-padded into the square canvas (all-255 when empty),
        resampled bilinearly to every feature level and thresholded at non-zero -> bool [T] over the flattened levels"""
        S = self.backbone.padding_constraints.get("square_size", 0)
        m = mask_prompt.to(dtype=torch.float32)
        h, w = m.shape[-2:]
        side = max(S, h, w) if S > 0 else None
        if side is not None:
            m = F.pad(m, (0, side - w, 0, side - h))
        if float(m.sum()) == 0:
            m = torch.full_like(m, 255.0)
        flat = [F.interpolate(m[None, None], size=(int(H), int(W)), mode="bilinear")[0, 0].to(torch.bool).reshape(-1) for H, W in geo.shapes]
        return torch.cat(flat)

    def forward_single(self, image, text_feats, forced_topk=None, stages=None, with_masks=True, prompt="name", instance=True,
                       semantic=None, detector_columns=None, panoptic=False, vit_feat=None, geo=None, encoder_done=None,
                       mask_prompt=None):
        """image [3,h,w] fp32 0..255 (device), text_feats [K, D_l] -> dict of device tensors (fixed shapes).
        encoder_done: optional torch.cuda.Event recorded on the current stream as soon as the encoder memory exists (the runtime
        orders the next step's ViT behind it, runtime.GraphedForward late_vit).
        instance: run the detection branch; semantic: metadata dict (entity, thing_classes, stuff_classes) to run the
        semantic branch; detector_columns: ("first", n) | ("ids", LongTensor) restriction of the detector's classes;
        vit_feat: this image's rows of a batched ViT pass (backbone.net.forward_tokens on a list of images);
        geo: a geometry.StaticGeometry loaded with the constants of the REAL image size while `image` is the full square canvas
        (pixels outside the image = the per-channel mean, i.e. exact zeros after normalisation) -- the size-agnostic form a
        captured graph needs (runtime.GraphedForward(any_size=True)); instance branch only.
        mask_prompt: [h, w] mask (non-zero = prompted region, :394-414): only encoder tokens inside it become proposals."""
        dt = self.compute_dtype
        P = self.packed(dt)
        h, w = image.shape[-2:]
        t0 = time.perf_counter()
        if stages is not None:
            maps = self.backbone.forward_tokens(image.contiguous(), self._mean, self._std, vit_feat=vit_feat, stages=stages)
            maps = {k: (tap(stages, k, v[0]), v[1]) for k, v in maps.items()}
        else:
            maps = self.backbone.forward_tokens(image.contiguous(), self._mean, self._std, vit_feat=vit_feat)
        self.backbone_time = time.perf_counter() - t0
        # neck = None (APE-L_A/B/C, ape_deta_vitl_eva02_lsj1024_cp_12ep.py:21): the pyramid's maps, in its order, are the levels
        names = self.neck.in_features if self.neck is not None else list(maps.keys())
        level_shapes = [maps[f][1] for f in names]
        if geo is None:
            geo = self.geometry((h, w), level_shapes)
        # (a given geometry = the size-agnostic forward: `image` is the S x S canvas, so (h, w) is the pad.  The semantic branch then yields
        # class scores over the whole pad and the panoptic branch its queries' mask logits over the whole pad; the caller crops / resizes /
        # merges with the image's own sizes behind the replay: runtime.GraphedForward._sem_labels, ._replay)
        t0 = time.perf_counter()
        src = torch.empty((geo.T, self.transformer.embed_dim), dtype=dt, device=image.device)
        def neck_level(i, f):
            if P["neck"] is None:
                src[geo.starts[i]:geo.starts[i] + maps[f][0].shape[0]].copy_(maps[f][0])
                return
            wn, bn, gw, gb, groups, eps = P["neck"][i]
            t = ops.gemm(maps[f][0], wn, bn)
            ops.groupnorm(t, gw, gb, groups, eps, out=src[geo.starts[i]:geo.starts[i] + t.shape[0]])

        jobs = [ops.fork(lambda i=i, f=f: neck_level(i, f)) for i, f in enumerate(names) if i > 0]   # small levels: parallel branches
        neck_level(0, names[0])
        for j in jobs:
            j.join()
        src = tap(stages, "enc_input", src)
        l0 = self.fusion_tokens(text_feats, prompt)
        want_masks = instance and with_masks and self.test_mask_on
        mask_job = []

        def mask_features(memory):
            # maskdino_mask_features (:728-750): lateral 1x1 + GN, + encoder memory of level 0, 3x3 + GN + ReLU, 1x1.
            # Needs only p2 and the encoder memory: forked here, it overlaps proposal selection and the decoder.
            H0, W0 = geo.shapes[0]
            p2 = maps[self.mask_in_features[0]][0]
            lat = ops.gemm(p2, P["lat"][0], None)
            lat = ops.groupnorm(lat, P["lat"][1], P["lat"][2], P["lat"][3], P["lat"][4], add=memory[: H0 * W0])
            y = ops.conv3x3(lat, None, H0, W0, P["outc"][0], None)
            y = ops.groupnorm(y, P["outc"][1], P["outc"][2], P["outc"][3], P["outc"][4], act=ops.ACT_RELU)
            return ops.gemm(y, P["maskc"], None)                                                      # [H0*W0, 256]

        def after_encoder(memory):
            if encoder_done is not None:
                encoder_done.record()
            if want_masks or semantic is not None or panoptic:
                mask_job.append(ops.fork(lambda: mask_features(memory)))

        tr = self.transformer.forward_tokens(src, geo, l0, dt, forced_topk, stages, after_encoder=after_encoder,
                                             mask_prompt=None if mask_prompt is None else self.mask_prompt_tokens(mask_prompt, geo))
        self.transformer_time = time.perf_counter() - t0
        t0 = time.perf_counter()
        # last decoder level only feeds inference (:519-524); "name" mode classifies against the RAW text bank (:446)
        lvl = self.transformer.decoder.num_layers - 1
        x = tr["inter_states"][lvl]
        ref_prev = tr["init_reference"] if lvl == 0 else tr["inter_references"][lvl - 1]
        if prompt == "name":
            tok, cbias, inv_scale = self.class_tokens(text_feats, lvl, dt)
        else:                                    # the FUSED tokens are the vocabulary (:448); they change per image
            tok, cbias, inv_scale = self.class_embed[lvl].text_side(tr["query_l"], dt)
        logits = self.class_embed[lvl].forward_tokens(x, tok, cbias, inv_scale)                       # [Q,K] fp32
        dec = self.transformer.decoder
        if stages is None and dec.bbox_embed is not None and dec.bbox_embed[lvl] is self.bbox_embed[lvl]:
            # with_box_refine: the decoder's own refinement of its last layer IS this head (:240-246 and :490-500 apply the same
            # module to the same tensors: sigmoid(bbox_embed[lvl](x) + inverse_sigmoid(reference))) -- no second evaluation
            boxes = tr["inter_references"][lvl]
        else:
            delta = self.bbox_embed[lvl].forward_tokens(x, dt, out_dtype=torch.float32)
            boxes, _ = ops.box_refine(delta, ref_prev.contiguous(), geo.vr4)
        logits, boxes = tap(stages, "pred_logits", logits), tap(stages, "pred_boxes", boxes)
        out = dict(pred_logits=logits, pred_boxes=boxes, topk_proposals=tr["topk_proposals"], geo=geo)
        det = {}
        if instance:
            det_logits = logits
            if detector_columns is not None:                 # (:578-590) thing columns only
                kind, arg = detector_columns
                if kind == "first":
                    det_logits = logits[:, :arg].contiguous()
                else:
                    det_logits = torch.full_like(logits, float("-inf"))
                    det_logits[:, arg] = logits[:, arg]
            det = self.inference_single(det_logits, boxes, (h, w), geo.box_scale)
            out.update(det)
        if want_masks or semantic is not None or panoptic:
            mask_feat = tap(stages, "mask_features", mask_job[0].join())
            membed = tap(stages, "mask_embed", self.mask_embed.forward_tokens(x, dt, out_dtype=dt))    # [Q,256]
        if semantic is not None:
            out["sem_seg"] = self.semantic_single(logits, boxes, membed, mask_feat, geo, (h, w), semantic, dt, stages)
        if panoptic:
            # panoptic_post_nms (:677-685): a third class-wise NMS on ALL class columns picks the queries; their mask logits
            # are upsampled to the padded input size (:569-572) and cropped to the image (sem_seg_postprocess, :942)
            if self.panoptic_post_nms:
                pdet = self.inference_single(logits, boxes, (h, w), geo.box_scale)
                pq, pvscore = pdet["det_query"], pdet["det_scores"]            # empty slots carry score -1
            else:
                pq, pvscore = ops.arange_i64(logits.shape[0], logits.device), None
            H0, W0 = geo.shapes[0]
            S = self.backbone.padding_constraints.get("square_size", 0)
            pkept = ops.gather_rows(membed, pq)
            plog = ops.gemm(pkept, mask_feat, None, out_dtype=torch.float32)                          # [k, H0*W0]
            up = ops.bilinear_resize(plog.view(-1, H0, W0), S, S)
            out.update(pan_masks=up[:, :h, :w], pan_cls=ops.gather_rows(logits, pq), pan_valid_score=pvscore, pan_query=pq)
        if want_masks:
            H0, W0 = geo.shapes[0]
            # only the kept queries are decoded / upsampled: the einsum (:510) and F.interpolate (:569-572) act per query
            kept = ops.gather_rows(membed, det["det_query"])
            mlog = ops.gemm(kept, mask_feat, None, out_dtype=torch.float32)                           # [n, H0*W0]
            S = self.backbone.padding_constraints.get("square_size", 0)
            bits = ops.mask_upsample_bits(mlog, H0, W0, S)
            out["det_masks128"] = ops.roi_align_bits(bits, det["det_boxes"].contiguous(), 128)
            if stages is not None:
                stages.update(det_mask_logits=mlog)
        if stages is not None:
            stages.update(inter_states=torch.stack(tr["inter_states"])[:, None],
                          inter_references=torch.stack(tr["inter_references"])[:, None], **det)
        self.postprocess_time = time.perf_counter() - t0
        return out

    @staticmethod
    def stuff_score(logits, meta):
        """get_stuff_score (deformable_detr_segm_vl.py:1251-1271): with a leading "things" stuff class, the thing columns
        collapse into one (their minimum) -- one library launch (csrc/softmax.hip stuff_collapse)."""
        thing, stuff, entity = meta.get("thing_classes") or [], meta.get("stuff_classes") or [], meta["entity"]
        overlap = len(thing) > 0 and len(stuff) > 0 and (set(thing) <= set(stuff) or set(stuff) <= set(thing))
        if entity == "thing+stuff" and stuff[0] == "things" and not overlap:
            return ops.stuff_collapse(logits, len(thing))
        return logits

    def semantic_single(self, logits, boxes, membed, mask_feat, geo, image_size, meta, dt, stages=None, pano_temp=0.06):
        """semantic branch for one image (:628-666 + _postprocess_semantic :875-918) -> [K', h, w] fp32 at the
        un-padded input resolution (the crop of sem_seg_postprocess is fused: padded pixels are never produced).
        The kept (query) set comes from a second class-wise NMS on the stuff scores; the einsum over the kept queries
        is one GEMM against pixel-major sigmoid(upsampled mask) probabilities."""
        h, w = image_size
        sem_logits = self.stuff_score(logits, meta)
        if self.semantic_post_nms:
            sdet = self.inference_single(sem_logits, boxes, image_size, geo.box_scale)
            qidx, vscore = sdet["det_query"], sdet["det_scores"]                 # empty slots of the fixed-shape list carry score -1
        else:
            qidx, vscore = ops.arange_i64(logits.shape[0], logits.device), None
        k = qidx.numel()
        kp = (k + 7) // 8 * 8
        # class weights of the kept queries, softmax(sigmoid / T) (:891-894), as the [K', kp] A operand; the kept queries' mask embeddings
        # as a zero-padded [kp, 256] operand -- library launches only (round 5: sigmoid / softmax / mask / zeros / transposed copy in torch)
        A = ops.sem_class_weights(sem_logits, qidx, vscore, pano_temp, kp, dt)
        kept = ops.zeros((kp, membed.shape[1]), dt, membed.device)
        ops.gather_rows(membed, qidx, out=kept[:k])
        H0, W0 = geo.shapes[0]
        S = self.backbone.padding_constraints.get("square_size", 0)
        mlog_t = ops.gemm(mask_feat, kept, None, out_dtype=torch.float32)                               # [H0*W0, kp] pixel-major
        prob = ops.mask_upsample_sigmoid(mlog_t, H0, W0, S, h, w, dt)                                   # [h*w, kp]   (:569-572, 895)
        sem = ops.gemm(A, prob, None, out_dtype=torch.float32)                                          # [K', h*w]   (:899)
        if stages is not None:
            stages.update(sem_box_cls=sem_logits, sem_query=qidx, sem_valid=(vscore >= 0) if vscore is not None else torch.ones_like(qidx, dtype=torch.bool))
        return sem.view(-1, h, w)

    def inference_single(self, logits, boxes, image_size, scale=None):
        """sigmoid scores, cxcywh -> xyxy * (w,h,w,h), clip, score threshold, class-wise NMS, top-k
        (deformable_detr_segm_vl.py:759-810 + ape_deta/fast_rcnn.py:97-201).  Fixed-shape outputs:
        det_boxes [k,4], det_scores [k] (-1 = empty slot), det_classes [k], det_query [k]."""
        h, w = image_size
        if scale is None:
            scale = torch.tensor([w, h, w, h], dtype=torch.float32, device=boxes.device)
        return ops.detections(logits.float() if logits.dtype != torch.float32 else logits, boxes.float().contiguous(), scale,
                              self.test_score_thresh, self.test_nms_thresh, self.test_topk_per_image)

    # ------------------------------------------------------------------ reference entry point
    @torch.no_grad()
    def forward(self, batched_inputs, do_postprocess=True):
        if self.training:
            raise NotImplementedError("ape_amd implements the inference forward only")
        entity, dataset_id = self.eval_dataset_entity, self.eval_dataset_id
        do_pan = self.panoptic_on and not (entity and "thing+stuff" not in entity)          # (:671-673)
        if do_pan and not (0 <= dataset_id < len(self.metadata_list)):
            raise RuntimeError("ape_amd: the panoptic branch needs an evaluation dataset with metadata (set_eval_dataset)")
        do_inst = self.instance_on and not (entity and "thing" not in entity)              # (:575-577)
        do_sem = self.semantic_on and not (entity and "stuff" not in entity)               # (:628-630)
        meta = None
        if do_sem:
            if not self.metadata_list:
                raise RuntimeError("ape_amd: semantic_on needs dataset metadata (dataset_metas dicts or set_metadata)")
            meta = dict(self.metadata_list[dataset_id], entity=self.dataset_entities[dataset_id])   # [-1] like the reference
        cols = None
        if do_inst and 0 <= dataset_id < len(self.metadata_list):                          # (:578-590)
            m = self.metadata_list[dataset_id]
            thing, stuff = m.get("thing_classes") or [], m.get("stuff_classes") or []
            if thing and stuff and (set(thing) <= set(stuff) or set(stuff) <= set(thing)):
                ids = list((m.get("thing_dataset_id_to_contiguous_id") or {}).values())
                cols = ("ids", torch.tensor(ids, dtype=torch.long, device=self.device))
            elif thing:
                cols = ("first", len(thing))
        results = []
        for inp in batched_inputs:
            t0 = time.perf_counter()
            image = inp["image"].to(self.device, non_blocking=True).float()
            feats, _, prompt = self.text_features(inp)
            # detections kept per image (:183-194): 1 for referring expressions, else the configured number(s)
            self.test_topk_per_image = 1 if prompt == "expression" else self.select_box_nums_for_evaluation
            if self.select_box_nums_for_evaluation_list is not None:
                self.test_topk_per_image = self.select_box_nums_for_evaluation_list[dataset_id]
            self.preprocess_time = time.perf_counter() - t0
            mp = inp.get("mask_prompt")
            out = self.forward_single(image, feats, prompt=prompt, instance=do_inst, semantic=meta, detector_columns=cols,
                                      panoptic=do_pan, mask_prompt=None if mp is None else mp.to(self.device))
            h, w = image.shape[-2:]
            height, width = inp.get("height", h), inp.get("width", w)
            res = {}
            if do_inst:
                res["instances"] = self.postprocess_instance(out, (h, w), height, width)
            if do_sem:
                r = ops.bilinear_resize(out["sem_seg"], height, width)                     # sem_seg_postprocess (:916)
                if (dataset_id >= 0 and meta["entity"] == "stuff" and (meta.get("stuff_classes") or [""])[0] == "things"
                        and self.stuff_prob_thing > 0):                                    # (:654-663)
                    r[0] = math.log(self.stuff_prob_thing / (1 - self.stuff_prob_thing))
                res["sem_seg"] = r
            if do_pan:
                res["panoptic_seg"] = self.postprocess_panoptic(out, height, width, self.metadata_list[dataset_id])
            results.append(res)
        return results

    def panoptic_device(self, out, height, width, meta):
        """_postprocess_panoptic (:921-998), Mask2Former-style merge of the kept queries' masks, WITHOUT a host round trip: fixed
        shapes (all k panoptic queries, dropped ones masked), so it captures into a hipGraph.  The per-query scores are tensor-level
        (k x K' values); the pixel work and the sequential walk over the queries are csrc/masks.hip `panoptic_*` (3 launches).
        -> (panoptic_seg int32 [H, W], info int32 [k, 3] = (id, isthing, category_id) per segment, count int32 [1]), all on the device"""
        cfg = self.panoptic_configs
        mask_cls = out["pan_cls"]
        # (:944-949) max of the sigmoid scores, the mask threshold, the optional softmax(sigmoid / T) rescoring: one library launch
        scores, _, keep, labels = ops.pan_class_scores(mask_cls, None, out.get("pan_valid_score"), cfg["object_mask_threshold"],
                                                       bool(cfg["transform_eval"]), cfg["pano_temp"])
        ids = tuple(sorted(i for i in (meta.get("thing_dataset_id_to_contiguous_id") or {}).values() if 0 <= i < mask_cls.shape[1]))
        key = (ids, mask_cls.shape[1], mask_cls.device)
        cache = self.__dict__.setdefault("_pan_thing_cache", {})          # per (thing ids, vocabulary width, device): built outside captures
        if key not in cache:
            thing = torch.zeros(mask_cls.shape[1], dtype=torch.bool)
            thing[list(ids)] = True
            cache[key] = thing.to(mask_cls.device)
        things_first = (meta.get("stuff_classes") or [""])[0] == "things"
        return ops.panoptic_merge(out["pan_masks"], scores, keep, labels, cache[key], height, width, prob=cfg["prob"],
                                  overlap_threshold=cfg["overlap_threshold"],
                                  stuff_offset=len(meta["thing_classes"]) if things_first else -1)

    @staticmethod
    def segments_info(info, count):
        """host view of the device table: [{"id", "isthing", "category_id"}] like the reference's list"""
        n = int(count.reshape(-1)[0])
        rows = info[:n].tolist()
        return [{"id": int(r[0]), "isthing": bool(r[1]), "category_id": int(r[2])} for r in rows]

    def postprocess_panoptic(self, out, height, width, meta):
        """-> (panoptic_seg int32 [H, W] on the device, segments_info list): the reference's return value; one read-back of the
        segment table at the end"""
        seg, info, count = self.panoptic_device(out, height, width, meta)
        return seg, self.segments_info(info.cpu(), count.cpu())

    def postprocess_instance(self, out, image_size, height, width):
        """detector_postprocess (:857-872): rescale to (height, width), clip, drop empty boxes, paste masks; the
        result is a detectron2-style `Instances` (ape_amd/structures.py) moved to the CPU like the reference (`r.to("cpu")`)."""
        h, w = image_size
        boxes = out["det_boxes"].clone()
        sx, sy = width / w, height / h
        boxes[:, 0::2] = (boxes[:, 0::2] * sx).clamp(0, width)
        boxes[:, 1::2] = (boxes[:, 1::2] * sy).clamp(0, height)
        keep = (out["det_scores"] >= 0) & ((boxes[:, 2] - boxes[:, 0]) > 0) & ((boxes[:, 3] - boxes[:, 1]) > 0)
        masks = None
        if "det_masks128" in out:
            masks = ops.paste_bits(out["det_masks128"], boxes.contiguous(), height, width)
        keep_c = keep.cpu()                                   # the one host sync of the forward
        return make_instances((height, width), boxes.cpu()[keep_c], out["det_scores"].cpu()[keep_c], out["det_classes"].cpu()[keep_c],
                              masks.cpu()[keep_c].bool() if masks is not None else None, query_index=out["det_query"].cpu()[keep_c])
