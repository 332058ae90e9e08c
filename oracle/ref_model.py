"""Instantiate the REFERENCE APE model (its own classes, loaded by oracle/refshim.py) with the constructor
arguments of the APE-L_D LazyConfig, plus scaled-down variants for fast CPU fixtures.

TEST INFRASTRUCTURE (oracle); needs /root/reference.  Constructor kwargs follow
  configs/common/backbone/vitl_eva02_clip.py:9-48,
  configs/COCO_InstanceSegmentation/ape_deta/models/ape_deta_r50.py:24-137,
  configs/LVISCOCOCOCOSTUFF_O365_OID_VGR_SA1B_REFCOCO_GQA_PhraseCut_Flickr30k/ape_deta/
      ape_deta_vitl_eva02_clip_vlf_lsj1024_cp_16x4_1080k.py:36-108,171-177.
"""
from functools import partial
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import refshim
from .configs import CONFIGS, window_block_indexes  # noqa: F401


class _TextStub(nn.Module):
    """Stands in for EVA02CLIP.forward_text (clip_wrapper_eva02.py:88-128): returns a fixed [K,1024] bank."""

    def __init__(self, feats):
        super().__init__()
        self.feats = feats

    def forward_text(self, text_list, cache=False):
        return {"last_hidden_state_eot": self.feats[: len(text_list)].clone()}


def _build_plain(c, A, backbone, feats, text_feats, semantic_on, panoptic_on, panoptic_configs):
    """APE-L_A/B/C: DeformableDETRSegm on the plain DeformableDetrTransformer, neck = None, constructor arguments of
    configs/COCO_InstanceSegmentation/ape_deta/models/ape_deta_r50.py:24-137 as overridden by ape_deta_vitl_eva02_lsj1024_cp_12ep.py:19-33
    and LVISCOCOCOCOSTUFF_O365_OID_VG/ape_deta/ape_deta_vitl_eva02_lsj1024_cp_720k.py:18-42"""
    encoder = A.DeformableDetrTransformerEncoder(embed_dim=256, num_heads=8, feedforward_dim=2048, attn_dropout=0.0, ffn_dropout=0.0,
                                                 num_layers=c.enc_layers, post_norm=False, num_feature_levels=5, pytorch_attn=True)
    decoder = A.DeformableDetrTransformerDecoder(embed_dim=256, num_heads=8, feedforward_dim=2048, attn_dropout=0.0, ffn_dropout=0.0,
                                                 num_layers=c.dec_layers, return_intermediate=True, num_feature_levels=5, pytorch_attn=True)
    transformer = A.DeformableDetrTransformer(encoder=encoder, decoder=decoder, as_two_stage=True, num_feature_levels=5,
                                              two_stage_num_proposals=c.num_queries, assign_first_stage=True)
    criterion = [nn.Module() for _ in range(1)]
    for cr in criterion:
        cr.loss_class_type = "focal_loss"
        cr.num_classes = 256
    model_vision = A.DeformableDETRSegm(
        backbone=backbone, position_embedding=refshim.PositionEmbeddingSine(num_pos_feats=128, temperature=10000,
                                                                          normalize=True, offset=-0.5),
        neck=None, transformer=transformer, embed_dim=256, num_classes=1256, num_queries=c.num_queries, aux_loss=True,
        with_box_refine=True, as_two_stage=True, criterion=criterion, pixel_mean=[123.675, 116.280, 103.530],
        pixel_std=[58.395, 57.120, 57.375], select_box_nums_for_evaluation=c.topk_eval, input_format="RGB",
        mask_encode_level=0, mask_in_features=["p2"], input_shapes={f: refshim.ShapeSpec(channels=256) for f in feats},
        output_dir=None, vis_period=0, embed_dim_language=1024, instance_on=True, semantic_on=semantic_on, panoptic_on=panoptic_on,
        dataset_prompts=["name"], dataset_names=["coco"], dataset_metas=["coco_2017_val"],
        **({"panoptic_configs": panoptic_configs} if panoptic_configs is not None else {}),
    )
    model = A.SomeThing(model_vision=model_vision, model_language=_TextStub(text_feats))
    model.eval()
    return model


def build_reference(cfg, text_feats, semantic_on=False, panoptic_on=False, panoptic_configs=None):
    refshim.install()
    import ape.layers as L
    import ape.modeling.ape_deta as A
    from ape.modeling.backbone.vit_eva_clip import SimpleFeaturePyramid, ViT

    c = SimpleNamespace(**cfg)
    if cfg.get("backbone") == "eva02":          # APE-Ti: configs/common/backbone/vitt_eva02.py:10-41; APE-L_A/B/C: vitl_eva02.py:10-41
        from ape.modeling.backbone.vit_eva02 import SimpleFeaturePyramid, ViT
        ge = cfg.get("global_every", 3)
        subln = bool(cfg.get("subln", False))
        net = ViT(img_size=c.img_size, patch_size=16, embed_dim=c.embed_dim, depth=c.depth, num_heads=c.num_heads,
                  drop_path_rate=0.0, window_size=c.window_size, mlp_ratio=4 * 2 / 3, qkv_bias=True,
                  norm_layer=partial(nn.LayerNorm, eps=1e-6), window_block_indexes=[i for i in range(c.depth) if i % ge != ge - 1],
                  residual_block_indexes=[], use_rel_pos=True, out_feature="last_feat", use_act_checkpoint=False, xattn=True,
                  subln=subln, swiglu=not subln, naiveswiglu=subln)
    elif cfg.get("backbone") == "clip_g":       # EVA-01-CLIP ViT-g: configs/common/backbone/vitg_eva01_clip_1024.py:9-45
        ge = cfg.get("global_every", 4)
        net = ViT(img_size=c.img_size, patch_size=16, embed_dim=c.embed_dim, depth=c.depth, num_heads=c.num_heads, drop_path_rate=0.0,
                  window_size=c.window_size, mlp_ratio=6144 / 1408, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                  window_block_indexes=[i for i in range(c.depth) if i % ge != ge - 1], residual_block_indexes=[], use_rel_pos=True,
                  out_feature="last_feat", use_act_checkpoint=False, xattn=True, pretrain_img_size=c.pretrain_img_size,
                  pretrain_use_cls_token=True)
    elif cfg.get("backbone") == "eva01":        # EVA-01 MIM ViT-g of vit_eva.py: configs/common/backbone/vitg_eva01.py:9-47 (relative positions)
        from ape.modeling.backbone.vit_eva import SimpleFeaturePyramid, ViT
        ge = cfg.get("global_every", 4)
        net = ViT(img_size=c.img_size, patch_size=16, embed_dim=c.embed_dim, depth=c.depth, num_heads=c.num_heads, drop_path_rate=0.0,
                  window_size=c.window_size, mlp_ratio=6144 / 1408, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                  window_block_indexes=[i for i in range(c.depth) if i % ge != ge - 1], residual_block_indexes=[], use_rel_pos=True,
                  rel_pos_zero_init=False, out_feature="last_feat", use_act_checkpoint=True, beit_like_qkv_bias=True, beit_like_gamma=False,
                  freeze_patch_embed=True, pretrain_img_size=c.pretrain_img_size)
    elif cfg.get("backbone") == "clip_e":       # ViT-e: configs/common/backbone/vite_eva02_clip_1024.py:9-49
        ge = cfg.get("global_every", 4)
        net = ViT(img_size=c.img_size, patch_size=16, embed_dim=c.embed_dim, depth=c.depth, num_heads=c.num_heads, drop_path_rate=0.0,
                  window_size=c.window_size, mlp_ratio=8.571428571428571, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                  window_block_indexes=[i for i in range(c.depth) if i % ge != ge - 1], residual_block_indexes=[], use_rel_pos=True,
                  out_feature="last_feat", use_act_checkpoint=False, xattn=True, pretrain_img_size=c.pretrain_img_size,
                  pretrain_use_cls_token=True, postnorm=True)
    else:
        net = ViT(
            img_size=c.img_size, patch_size=16, embed_dim=c.embed_dim, depth=c.depth, num_heads=c.num_heads,
            drop_path_rate=0.0, window_size=c.window_size, mlp_ratio=4 * 2 / 3, qkv_bias=True,
            norm_layer=partial(nn.LayerNorm, eps=1e-6), window_block_indexes=window_block_indexes(c.depth),
            residual_block_indexes=[], use_rel_pos=True, out_feature="last_feat", use_act_checkpoint=False, xattn=True,
            rope=True, pt_hw_seq_len=16, intp_freq=True, naiveswiglu=True, subln=True, pretrain_img_size=c.pretrain_img_size,
            pretrain_use_cls_token=True,
        )
    backbone = SimpleFeaturePyramid(net=net, in_feature="last_feat", out_channels=256, scale_factors=(4.0, 2.0, 1.0, 0.5),
                                    top_block=refshim.LastLevelMaxPool(), norm="LN", square_pad=c.img_size)
    feats = ["p2", "p3", "p4", "p5", "p6"]
    if not cfg.get("vl", True):
        return _build_plain(c, A, backbone, feats, text_feats, semantic_on, panoptic_on, panoptic_configs)
    neck = refshim.ChannelMapper(input_shapes={f: refshim.ShapeSpec(channels=256) for f in feats}, in_features=feats,
                                 out_channels=256, num_outs=5, kernel_size=1, norm_layer=nn.GroupNorm(32, 256))
    vl_layer = L.VisionLanguageFusion(v_dim=256, l_dim=1024, embed_dim=2048, num_heads=8, dropout=0.1, drop_path=0.0,
                                      init_values=1.0 / 6, stable_softmax_2d=True, clamp_min_for_underflow=True,
                                      clamp_max_for_overflow=True, use_checkpoint=True)
    encoder = A.DeformableDetrTransformerEncoderVL(embed_dim=256, num_heads=8, feedforward_dim=2048, attn_dropout=0.0,
                                                   ffn_dropout=0.0, num_layers=c.enc_layers, post_norm=False,
                                                   num_feature_levels=5, vl_layer=vl_layer, use_act_checkpoint=True,
                                                   pytorch_attn=True)
    decoder = A.DeformableDetrTransformerDecoderVL(embed_dim=256, num_heads=8, feedforward_dim=2048, attn_dropout=0.0,
                                                   ffn_dropout=0.0, num_layers=c.dec_layers, return_intermediate=True,
                                                   num_feature_levels=5, pytorch_attn=True)
    transformer = A.DeformableDetrTransformerVL(encoder=encoder, decoder=decoder, as_two_stage=True, num_feature_levels=5,
                                                two_stage_num_proposals=c.num_queries, assign_first_stage=True,
                                                proposal_ambiguous=1)
    criterion = [nn.Module() for _ in range(1)]
    for cr in criterion:
        cr.loss_class_type = "focal_loss"
        cr.num_classes = 256
    model_vision = A.DeformableDETRSegmVL(
        backbone=backbone, position_embedding=refshim.PositionEmbeddingSine(num_pos_feats=128, temperature=10000,
                                                                          normalize=True, offset=-0.5),
        neck=neck, transformer=transformer, embed_dim=256, num_classes=1256, num_queries=c.num_queries, aux_loss=True,
        with_box_refine=True, as_two_stage=True, criterion=criterion, pixel_mean=[123.675, 116.280, 103.530],
        pixel_std=[58.395, 57.120, 57.375], select_box_nums_for_evaluation=c.topk_eval, input_format="RGB",
        mask_encode_level=0, mask_in_features=["p2"], input_shapes={f: refshim.ShapeSpec(channels=256) for f in feats},
        output_dir=None, vis_period=0, embed_dim_language=1024, instance_on=True, semantic_on=semantic_on, panoptic_on=panoptic_on,
        text_feature_bank=True, text_feature_reduce_before_fusion=True, text_feature_batch_repeat=True,
        expression_cumulative_gt_class=True, name_prompt_fusion_type="zero", dataset_prompts=["name"],
        dataset_names=["coco"], dataset_metas=["coco_2017_val"], text_feature_bank_reset=True,
        **({"panoptic_configs": panoptic_configs} if panoptic_configs is not None else {}),
    )
    model = A.SomeThing(model_vision=model_vision, model_language=_TextStub(text_feats))
    model.eval()
    return model
