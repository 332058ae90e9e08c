"""Execute the reference model (its own source files under oracle/refshim.py) and capture per-stage tensors.

TEST INFRASTRUCTURE (oracle); this container only (needs /root/reference).  Used by
tests/test_oracle_vs_reference.py and tests/golden/make_golden.py.
"""
import sys

import torch

from . import ref_model, refshim, weights
from .ape_oracle import stable_topk
from .configs import CONFIGS


class _TorchProxy:
    """`torch` as seen by deformable_transformer_vl.py: topk gets the oracle's defined tie rule
    (value desc, index asc -- the reference leaves ties to torch.topk's unspecified order), and the
    proposal gather is recorded so topk_proposals can be compared."""

    def __init__(self, rec, stable_ties, forced_topk=None):
        self._rec, self._stable, self._forced = rec, stable_ties, forced_topk

    def stack(self, tensors, *a, **kw):
        # `topk_proposals = torch.stack(topk_proposals)` (deformable_transformer_vl.py:625): with forced_topk the decoder of THIS run
        # starts from the proposals of another run (the fp32 run's, for a reduced-precision yardstick that measures arithmetic, not a
        # different selection)
        if (self._forced is not None and len(tensors) and all(torch.is_tensor(t) and t.dim() == 1 and t.dtype == torch.int64 for t in tensors)
                and len(tensors) == self._forced.shape[0] and tensors[0].numel() == self._forced.shape[1]):
            return self._forced.clone()
        return torch.stack(tensors, *a, **kw)

    def __getattr__(self, name):
        return getattr(torch, name)

    def topk(self, x, k, dim=-1, **kw):
        if not self._stable or x.dim() != 1:
            return torch.topk(x, k, dim=dim, **kw)
        idx = stable_topk(x, k)
        return x[idx], idx

    def gather(self, inp, dim, index, **kw):
        if index.dim() == 3 and index.shape[-1] == 4 and dim == 1:
            self._rec["topk_proposals"] = index[..., 0].clone()
        return torch.gather(inp, dim, index, **kw)


def spec_of(model):
    return [(k, list(v.shape)) for k, v in model.state_dict().items()]


def run_reference(cfg_name, seed, image, text_feats, height=None, width=None, stable_ties=True, prompt="name", semantic=None,
                  eval_dataset=False, panoptic_configs=None, mask_prompt=None, autocast=None, forced_topk=None):
    """returns (stages dict, instances dict, spec).  prompt="phrase": class names with a space, which the reference
    routes to the dense multi-token fusion (deformable_detr_segm_vl.py:224-232, 283-337).
    autocast = torch.bfloat16: the forward runs under torch.autocast("cpu", dtype) -- the reference's own code at reduced precision, the
    yardstick of tests/golden/make_autocast_yardstick.py; forced_topk [1, Q]: proposals injected (see _TorchProxy.stack)."""
    cfg = CONFIGS[cfg_name]
    refshim.METADATA.clear()
    if semantic is not None:        # semantic branch on: the (only) dataset's metadata carries the thing / stuff split
        refshim.METADATA["coco_2017_val"] = {k: semantic[k] for k in ("thing_classes", "stuff_classes",
                                                                     "thing_dataset_id_to_contiguous_id") if semantic.get(k)}
    # eval_dataset: the model is pointed at its (only) dataset like the evaluators do (set_eval_dataset): class names come
    # from the metadata (get_text_list), the detector sees the thing columns only, and the panoptic merge runs
    model = ref_model.build_reference(cfg, text_feats, semantic_on=semantic is not None, panoptic_on=bool(eval_dataset),
                                      panoptic_configs=panoptic_configs)
    if eval_dataset:
        model.model_vision.set_eval_dataset("coco_2017_val")
    spec = spec_of(model)
    sd = weights.make_state_dict(spec, seed)
    weights.load_into(model, sd)
    mv = model.model_vision
    S = {}
    hooks = []

    def hook(mod, name, fn=lambda o: o):
        hooks.append(mod.register_forward_hook(lambda m, i, o: S.__setitem__(name, fn(o))))

    net = mv.backbone.net
    for i, blk in enumerate(net.blocks):
        hook(blk, f"vit_block{i}")
    hook(net, "last_feat", lambda o: o["last_feat"])
    hooks.append(mv.backbone.register_forward_hook(lambda m, i, o: S.update({k: v for k, v in o.items()})))
    enc = mv.transformer.encoder
    vl = cfg.get("vl", True)          # False: APE-L_A/B/C, the plain DeformableDETRSegm / DeformableDetrTransformer (no fusion layers)
    for i in range(len(enc.layers)):
        if vl:
            hooks.append(enc.vl_layers[i].register_forward_hook(
                lambda m, inp, o, i=i: S.update({f"enc{i}_fused_v": o[0], f"enc{i}_fused_l": o[1]})))
        hook(enc.layers[i], f"enc{i}_out")
    hooks.append(mv.transformer.register_forward_hook(lambda m, i, o: S.update({
        "inter_states": o[0], "init_reference": o[1], "inter_references": o[2], "enc_class": o[3],
        "enc_coord_unact": o[4], "anchors": o[5], "memory": o[6], **({"query_l": o[7]} if len(o) > 7 else {})})))
    hooks.append(mv.transformer.register_forward_pre_hook(lambda m, a: S.__setitem__("transformer_inputs", a)))
    hooks.append(mv.transformer.decoder.register_forward_pre_hook(
        lambda m, a, kw: S.update({"query_init": kw["query"], "query_pos": kw["query_pos"]}), with_kwargs=True))
    hook(mv.transformer.enc_output_norm, "output_memory")
    hook(mv.mask_embed, "mask_embed")
    hook(mv.class_embed[len(mv.transformer.decoder.layers) - 1], "pred_logits_full")   # all K columns (the detector may see fewer)

    tmod = sys.modules["ape.modeling.ape_deta.deformable_transformer_vl" if vl else "ape.modeling.ape_deta.deformable_transformer"]
    smod = sys.modules["ape.modeling.ape_deta.deformable_detr_segm_vl" if vl else "ape.modeling.ape_deta.deformable_detr_segm"]
    old_torch = tmod.torch
    tmod.torch = _TorchProxy(S, stable_ties, forced_topk)
    old_mf = mv.maskdino_mask_features
    old_inf = mv.inference
    old_retry = smod.retry_if_cuda_oom

    def mf(*a, **k):
        out = old_mf(*a, **k)
        S["mask_features"] = out
        return out

    calls = []

    def inf(box_cls, box_pred, image_sizes, use_sigmoid=True):
        calls.append(1)
        if "pred_logits" in S:      # later calls: semantic_post_nms (:638-647), then panoptic_post_nms (:677-685)
            res, filt = old_inf(box_cls, box_pred, image_sizes, use_sigmoid=use_sigmoid)
            if semantic is not None and "sem_query" not in S:
                S["sem_box_cls"], S["sem_query"] = box_cls, filt[0]
            else:
                S["pan_query"] = filt[0]
            return res, filt
        S["pred_logits"], S["pred_boxes"] = box_cls, box_pred
        res, filt = old_inf(box_cls, box_pred, image_sizes, use_sigmoid=use_sigmoid)
        S["det_boxes"], S["det_scores"] = res[0].pred_boxes.tensor, res[0].scores
        S["det_classes"], S["det_query"] = res[0].pred_classes, filt[0]
        return res, filt

    def retry(func):
        def wrapped(x, **kw):
            S["pred_masks"] = x
            return func(x, **kw)
        return wrapped

    mv.maskdino_mask_features, mv.inference, smod.retry_if_cuda_oom = mf, inf, retry
    try:
        h, w = image.shape[-2:]
        inputs = {"image": image, "height": height or h, "width": width or w}
        if not eval_dataset:
            inputs.update(prompt="text", text_prompt=",".join((f"c {i}" if prompt == "phrase" else f"c{i}")
                                                                for i in range(text_feats.shape[0])))
        if mask_prompt is not None:                    # the predictor's inputs["mask_prompt"] (ape/engine/defaults.py:226-228)
            inputs["mask_prompt"] = mask_prompt
        import contextlib
        ctx = torch.autocast("cpu", dtype=autocast) if autocast is not None else contextlib.nullcontext()
        with torch.no_grad(), ctx:
            out = model([inputs])[0]
    finally:
        tmod.torch = old_torch
        mv.maskdino_mask_features, mv.inference, smod.retry_if_cuda_oom = old_mf, old_inf, old_retry
        for hk in hooks:
            hk.remove()
    if "sem_seg" in out:
        S["sem_seg"] = out["sem_seg"]
    if "panoptic_seg" in out:
        S["panoptic_seg"], S["segments_info"] = out["panoptic_seg"]
    inst = out["instances"]
    instances = {"pred_boxes": inst.pred_boxes.tensor, "scores": inst.scores, "pred_classes": inst.pred_classes,
                 "pred_masks": inst.pred_masks if inst.has("pred_masks") else None}
    return S, instances, spec, sd
