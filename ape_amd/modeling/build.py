"""Build the APE-L_D model graph from our classes with the constructor arguments of the reference's LazyConfig
(configs/common/backbone/vitl_eva02_clip.py:9-48, configs/COCO_InstanceSegmentation/ape_deta/models/ape_deta_r50.py:24-137,
configs/LVISCOCOCOCOSTUFF_O365_OID_VGR_SA1B_REFCOCO_GQA_PhraseCut_Flickr30k/ape_deta/
ape_deta_vitl_eva02_clip_vlf_lsj1024_cp_16x4_1080k.py:36-108,171-177) -- what detectron2.config.instantiate(cfg.model)
does in a full environment.  Scaled-down sizes exist for CPU-side tests.
"""
import math
from functools import partial
from types import SimpleNamespace

import torch
import torch.nn as nn

from ..layers import VisionLanguageFusion
from .ape_deta import (DeformableDETRSegmVL, DeformableDetrTransformerDecoderVL, DeformableDetrTransformerEncoderVL,
                       DeformableDetrTransformerVL, SomeThing)
from .ape_deta._containers import ChannelMapper, PositionEmbeddingSine
from .backbone.vit_eva_clip import LastLevelMaxPool, SimpleFeaturePyramid, ViT

SIZES = {
    "tiny": dict(img_size=256, embed_dim=128, depth=3, num_heads=2, window_size=8, pretrain_img_size=112, enc_layers=2,
                 dec_layers=2, num_queries=300, topk_eval=50),
    "small": dict(img_size=512, embed_dim=256, depth=6, num_heads=4, window_size=16, pretrain_img_size=224, enc_layers=3,
                  dec_layers=3, num_queries=900, topk_eval=100),
    # select_box_nums_for_evaluation: 300 in the APE-L_D joint config (..._lsj1024_cp_16x4_1080k.py:108), 100 in the COCO
    # instance-segmentation config that scripts/eval_APE-L_D.sh evaluates next (ape_deta_r50.py:121)
    "L_D": dict(img_size=1024, embed_dim=1024, depth=24, num_heads=16, window_size=32, pretrain_img_size=336, enc_layers=6,
                dec_layers=6, num_queries=900, topk_eval=300),
    "L_D_coco": dict(img_size=1024, embed_dim=1024, depth=24, num_heads=16, window_size=32, pretrain_img_size=336, enc_layers=6,
                     dec_layers=6, num_queries=900, topk_eval=100),
    # APE-Ti (BASELINE config 1): configs/common/backbone/vitt_eva02.py:10-41 -- the EVA-02 MIM ViT-Ti of vit_eva02.py
    "Ti": dict(img_size=1024, embed_dim=192, depth=12, num_heads=3, window_size=14, pretrain_img_size=224, enc_layers=6,
               dec_layers=6, num_queries=900, topk_eval=300, backbone="eva02"),
    # APE-L_A / L_B / L_C: the EVA-02 MIM ViT-L of vit_eva02.py (configs/common/backbone/vitl_eva02.py:10-41: 16 x 16 windows on the
    # 64 x 64 grid, every sixth block global, separate q/k/v projections, SwiGLU with its sub-LayerNorm)
    # ... under the PLAIN model family (vl=False): DeformableDETRSegm on DeformableDetrTransformer, neck = None, no ambiguous heads
    # (configs/COCO_InstanceSegmentation/ape_deta/models/ape_deta_r50.py:24-137 + ape_deta_vitl_eva02_lsj1024_cp_12ep.py:19-33)
    "L_A": dict(img_size=1024, embed_dim=1024, depth=24, num_heads=16, window_size=16, pretrain_img_size=224, enc_layers=6,
                dec_layers=6, num_queries=900, topk_eval=300, backbone="eva02", subln=True, global_every=6, vl=False),
    "small_A": dict(img_size=512, embed_dim=256, depth=6, num_heads=4, window_size=16, pretrain_img_size=224, enc_layers=2,
                    dec_layers=2, num_queries=300, topk_eval=50, backbone="eva02", subln=True, global_every=3, vl=False),
    "L_D_1536": dict(img_size=1536, embed_dim=1024, depth=24, num_heads=16, window_size=32, pretrain_img_size=336,
                     enc_layers=6, dec_layers=6, num_queries=900, topk_eval=500),
    # APE on ViT-e (ape_deta_vite_eva02_clip_vlf_lsj1024_cp_16x4_1080k_mdl_fsdp.py:24,65-66; vite_eva02_clip_1024.py:9-49): 64
    # post-norm blocks of width 1792 = 16 heads x 112 (zero-padded to the attention kernel's 128), packed qkv, GELU MLP, no rope,
    # every fourth block global, 9 + 9 DETA layers; small_E keeps the head width and a layer count other than 6
    "E_D": dict(img_size=1024, embed_dim=1792, depth=64, num_heads=16, window_size=32, pretrain_img_size=224, enc_layers=9,
                dec_layers=9, num_queries=900, topk_eval=300, backbone="clip_e", global_every=4),
    "small_E": dict(img_size=512, embed_dim=224, depth=4, num_heads=2, window_size=16, pretrain_img_size=224, enc_layers=3,
                    dec_layers=3, num_queries=300, topk_eval=50, backbone="clip_e", global_every=4),
    # APE on the EVA-01-CLIP ViT-g (ape_deta_vitg_eva01_clip_lsj1536_cp_64x90k.py; vitg_eva01_clip_1536.py): packed qkv, GELU MLP, no rope,
    # pre-norm, 40 blocks of width 1408 = 16 heads x 88 (zero-padded to 128), plain model family
    "G_A": dict(img_size=1536, embed_dim=1408, depth=40, num_heads=16, window_size=32, pretrain_img_size=224, enc_layers=6,
                dec_layers=6, num_queries=900, topk_eval=300, backbone="clip_g", global_every=4, vl=False),
    "small_G": dict(img_size=512, embed_dim=352, depth=4, num_heads=4, window_size=16, pretrain_img_size=224, enc_layers=2,
                    dec_layers=2, num_queries=300, topk_eval=50, backbone="clip_g", global_every=4, vl=False),
    # APE on the EVA-01 MIM ViT-g of vit_eva.py (ape_deta_vitg_eva01_lsj1536_cp_64x90k.py; vitg_eva01.py / vitg_eva01_1536.py): pre-norm,
    # packed qkv with q / v bias, GELU MLP, DECOMPOSED RELATIVE POSITIONS in every attention, 16 x 16 windows, every fourth block global,
    # 16 heads x 88, plain model family.  V_A_1536 is the configuration the reference trains; small_V keeps the head width on a 32 x 32 grid
    "V_A": dict(img_size=1024, embed_dim=1408, depth=40, num_heads=16, window_size=16, pretrain_img_size=224, enc_layers=6,
                dec_layers=6, num_queries=900, topk_eval=300, backbone="eva01", global_every=4, vl=False),
    "V_A_1536": dict(img_size=1536, embed_dim=1408, depth=40, num_heads=16, window_size=32, pretrain_img_size=224, enc_layers=6,
                     dec_layers=6, num_queries=900, topk_eval=300, backbone="eva01", global_every=4, vl=False),
    "small_V": dict(img_size=512, embed_dim=352, depth=4, num_heads=4, window_size=16, pretrain_img_size=224, enc_layers=2,
                    dec_layers=2, num_queries=300, topk_eval=50, backbone="eva01", global_every=4, vl=False),
}


def _build_plain(c, backbone, shapes, model_language, vision_kwargs):
    """APE-L_A/B/C: the reference's DeformableDETRSegm / DeformableDetrTransformer (no vision-language fusion), neck = None"""
    from .ape_deta import (DeformableDETRSegm, DeformableDetrTransformer, DeformableDetrTransformerDecoder,
                           DeformableDetrTransformerEncoder)
    encoder = DeformableDetrTransformerEncoder(embed_dim=256, num_heads=8, feedforward_dim=2048, attn_dropout=0.0, ffn_dropout=0.0,
                                               num_layers=c.enc_layers, post_norm=False, num_feature_levels=5)
    decoder = DeformableDetrTransformerDecoder(embed_dim=256, num_heads=8, feedforward_dim=2048, attn_dropout=0.0, ffn_dropout=0.0,
                                               num_layers=c.dec_layers, return_intermediate=True, num_feature_levels=5)
    transformer = DeformableDetrTransformer(encoder=encoder, decoder=decoder, as_two_stage=True, num_feature_levels=5,
                                            two_stage_num_proposals=c.num_queries, assign_first_stage=True)
    vkw = dict(
        backbone=backbone, position_embedding=PositionEmbeddingSine(num_pos_feats=128, temperature=10000, normalize=True, offset=-0.5),
        neck=None, transformer=transformer, embed_dim=256, num_classes=1256, num_queries=c.num_queries, criterion=[],
        pixel_mean=[123.675, 116.280, 103.530], pixel_std=[58.395, 57.120, 57.375], aux_loss=True, with_box_refine=True,
        as_two_stage=True, select_box_nums_for_evaluation=c.topk_eval, input_format="RGB", mask_encode_level=0,
        mask_in_features=["p2"], input_shapes=shapes, embed_dim_language=1024, instance_on=True, semantic_on=False,
        panoptic_on=False, dataset_prompts=["name"], dataset_names=["coco"], dataset_metas=["coco_2017_val"], stuff_prob_thing=0.9)
    vkw.update(vision_kwargs or {})
    model = SomeThing(model_vision=DeformableDETRSegm(**vkw), model_language=model_language)
    model.eval()
    return model


def build_ape(size="L_D", model_language=None, vision_kwargs=None, **overrides):
    c = SimpleNamespace(**{**SIZES[size], **overrides}) if isinstance(size, str) else SimpleNamespace(**{**size, **overrides})
    feats = ["p2", "p3", "p4", "p5", "p6"]
    if getattr(c, "backbone", "eva_clip") == "eva02":
        from .backbone import vit_eva02
        net = vit_eva02.ViT(img_size=c.img_size, patch_size=16, embed_dim=c.embed_dim, depth=c.depth, num_heads=c.num_heads,
                            drop_path_rate=0.8, window_size=c.window_size, mlp_ratio=4 * 2 / 3, qkv_bias=True,
                            norm_layer=partial(nn.LayerNorm, eps=1e-6),
                            window_block_indexes=[i for i in range(c.depth) if i % getattr(c, "global_every", 3) != getattr(c, "global_every", 3) - 1],
                            residual_block_indexes=[], use_rel_pos=True, out_feature="last_feat", use_act_checkpoint=False, xattn=True,
                            subln=getattr(c, "subln", False), swiglu=not getattr(c, "subln", False), naiveswiglu=getattr(c, "subln", False))
    elif getattr(c, "backbone", "eva_clip") == "clip_g":      # EVA-01-CLIP ViT-g: configs/common/backbone/vitg_eva01_clip_1024.py:9-45
        ge = getattr(c, "global_every", 4)
        net = ViT(img_size=c.img_size, patch_size=16, embed_dim=c.embed_dim, depth=c.depth, num_heads=c.num_heads, drop_path_rate=0.6,
                  window_size=c.window_size, mlp_ratio=6144 / 1408, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                  window_block_indexes=[i for i in range(c.depth) if i % ge != ge - 1], residual_block_indexes=[], use_rel_pos=True,
                  out_feature="last_feat", use_act_checkpoint=True, xattn=True, pretrain_img_size=c.pretrain_img_size,
                  pretrain_use_cls_token=True)
    elif getattr(c, "backbone", "eva_clip") == "eva01":       # EVA-01 MIM ViT-g: configs/common/backbone/vitg_eva01.py:9-47
        from .backbone import vit_eva
        ge = getattr(c, "global_every", 4)
        net = vit_eva.ViT(img_size=c.img_size, patch_size=16, embed_dim=c.embed_dim, depth=c.depth, num_heads=c.num_heads, drop_path_rate=0.6,
                          window_size=c.window_size, mlp_ratio=6144 / 1408, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                          window_block_indexes=[i for i in range(c.depth) if i % ge != ge - 1], residual_block_indexes=[], use_rel_pos=True,
                          rel_pos_zero_init=False, out_feature="last_feat", use_act_checkpoint=True, beit_like_qkv_bias=True,
                          beit_like_gamma=False, freeze_patch_embed=True, pretrain_img_size=c.pretrain_img_size)
    elif getattr(c, "backbone", "eva_clip") == "clip_e":      # ViT-e: configs/common/backbone/vite_eva02_clip_1024.py:9-49
        ge = getattr(c, "global_every", 4)
        net = ViT(img_size=c.img_size, patch_size=16, embed_dim=c.embed_dim, depth=c.depth, num_heads=c.num_heads, drop_path_rate=0.4,
                  window_size=c.window_size, mlp_ratio=8.571428571428571, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                  window_block_indexes=[i for i in range(c.depth) if i % ge != ge - 1], residual_block_indexes=[], use_rel_pos=True,
                  out_feature="last_feat", use_act_checkpoint=True, xattn=True, pretrain_img_size=c.pretrain_img_size,
                  pretrain_use_cls_token=True, postnorm=True)
    else:
        net = ViT(img_size=c.img_size, patch_size=16, embed_dim=c.embed_dim, depth=c.depth, num_heads=c.num_heads,
                  drop_path_rate=0.4, window_size=c.window_size, mlp_ratio=4 * 2 / 3, qkv_bias=True,
                  norm_layer=partial(nn.LayerNorm, eps=1e-6), window_block_indexes=[i for i in range(c.depth) if i % 3 != 2],
                  residual_block_indexes=[], use_rel_pos=True, out_feature="last_feat", use_act_checkpoint=True, xattn=True,
                  rope=True, pt_hw_seq_len=16, intp_freq=True, naiveswiglu=True, subln=True,
                  pretrain_img_size=c.pretrain_img_size, pretrain_use_cls_token=True)
    backbone = SimpleFeaturePyramid(net=net, in_feature="last_feat", out_channels=256, scale_factors=(4.0, 2.0, 1.0, 0.5),
                                    top_block=LastLevelMaxPool(), norm="LN", square_pad=c.img_size)
    shapes = {f: SimpleNamespace(channels=256) for f in feats}
    if not getattr(c, "vl", True):
        return _build_plain(c, backbone, shapes, model_language, vision_kwargs)
    neck = ChannelMapper(input_shapes=shapes, in_features=feats, out_channels=256, num_outs=5, kernel_size=1,
                         norm_layer=nn.GroupNorm(num_groups=32, num_channels=256))
    vl = VisionLanguageFusion(v_dim=256, l_dim=1024, embed_dim=2048, num_heads=8, dropout=0.1, drop_path=0.0,
                              init_values=1.0 / 6, stable_softmax_2d=True, clamp_min_for_underflow=True,
                              clamp_max_for_overflow=True, use_checkpoint=True)
    encoder = DeformableDetrTransformerEncoderVL(embed_dim=256, num_heads=8, feedforward_dim=2048, attn_dropout=0.0,
                                                 ffn_dropout=0.0, num_layers=c.enc_layers, post_norm=False,
                                                 num_feature_levels=5, vl_layer=vl, use_act_checkpoint=True)
    decoder = DeformableDetrTransformerDecoderVL(embed_dim=256, num_heads=8, feedforward_dim=2048, attn_dropout=0.0,
                                                 ffn_dropout=0.0, num_layers=c.dec_layers, return_intermediate=True,
                                                 num_feature_levels=5)
    transformer = DeformableDetrTransformerVL(encoder=encoder, decoder=decoder, as_two_stage=True, num_feature_levels=5,
                                              two_stage_num_proposals=c.num_queries, assign_first_stage=True,
                                              proposal_ambiguous=1)
    vkw = dict(
        backbone=backbone, position_embedding=PositionEmbeddingSine(num_pos_feats=128, temperature=10000, normalize=True, offset=-0.5),
        neck=neck, transformer=transformer, embed_dim=256, num_classes=1256, num_queries=c.num_queries, criterion=[],
        pixel_mean=[123.675, 116.280, 103.530], pixel_std=[58.395, 57.120, 57.375], aux_loss=True, with_box_refine=True,
        as_two_stage=True, select_box_nums_for_evaluation=c.topk_eval, input_format="RGB", mask_encode_level=0,
        mask_in_features=["p2"], input_shapes=shapes, embed_dim_language=1024, instance_on=True, semantic_on=False,
        panoptic_on=False, text_feature_bank=True, text_feature_reduce_before_fusion=True, text_feature_batch_repeat=True,
        name_prompt_fusion_type="zero", dataset_prompts=["name"], dataset_names=["coco"], dataset_metas=["coco_2017_val"],
        text_feature_bank_reset=False,      # the APE-*_D default (only the D3 configs reset the bank)
        stuff_prob_thing=0.9)               # config :172; semantic_on is True there (:176) and decided per evaluation dataset --
    vkw.update(vision_kwargs or {})         # e.g. vision_kwargs=dict(semantic_on=True, dataset_metas=[{...}]) for stuff datasets
    mv = DeformableDETRSegmVL(**vkw)
    model = SomeThing(model_vision=mv, model_language=model_language)
    model.eval()
    return model


@torch.no_grad()
def init_synthetic(model, seed=0):
    """Seeded NON-degenerate weights for benchmarking / smoke runs when no checkpoint is available: unit-variance
    preserving linears, non-zero deformable offsets and attention logits (the reference zero-initialises those,
    multi_scale_deform_attn.py:194-209, which would collapse every sample onto the reference point)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    seen = set()
    for name, p in list(model.named_parameters()) + list(model.named_buffers()):
        if id(p) in seen or name.endswith(("freqs_cos", "freqs_sin", "pixel_mean", "pixel_std")):
            continue
        seen.add(id(p))
        leaf = name.rsplit(".", 1)[-1]
        r = torch.randn(p.shape, generator=g)
        if name.endswith("name_prompt_fusion_feature"):
            t = torch.zeros(p.shape)
        elif leaf in ("gamma_v", "gamma_l"):
            t = 1.0 / 6 + 0.02 * r
        elif leaf == "log_scale":
            t = torch.zeros(p.shape)
        elif leaf == "bias0" or (leaf == "bias" and p.numel() == 1):
            t = -math.log(99.0) + 0.1 * r
        elif leaf == "level_embeds":
            t = r
        elif leaf == "pos_embed":
            t = 0.02 * r
        elif name.endswith("sampling_offsets.bias"):
            L = p.numel() // 64
            th = torch.arange(8, dtype=torch.float32) * (2.0 * math.pi / 8)
            grid = torch.stack([th.cos(), th.sin()], -1)
            grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(8, 1, 1, 2).repeat(1, L, 4, 1)
            for i in range(4):
                grid[:, :, i, :] *= i + 1
            t = grid.reshape(-1) + 0.1 * r
        elif p.dim() >= 2:
            t = r * ((0.5 if "sampling_offsets" in name else 1.0) / math.sqrt(p[0].numel()))
        elif leaf == "weight":
            t = 1.0 + 0.1 * r
        else:
            t = 0.02 * r
        p.copy_(t.to(p.dtype))
    return model
