"""The plain (non-VL) APE family of scripts/eval_APE-L_A.sh / L_B / L_C on the HIP kernels.

Mirror of ape/modeling/ape_deta/deformable_detr_segm.py (DeformableDETRSegm :32-1629) and deformable_transformer.py
(DeformableDetrTransformerEncoder :20-106, DeformableDetrTransformerDecoder :109-238, DeformableDetrTransformer :241-644): same class
names, constructor kwargs and state-dict keys.  In the reference these files are the VL files minus the vision-language fusion
(`diff` of the two pairs: the encoder loop without `vl_layers`, no fusion token selection, no mask prompts, transformer.forward
returning 7 instead of 8 values); here they are the same classes with the fusion switched off, so every kernel, the two-stage
selection and the instance / semantic / panoptic tails are shared.  The L_A configs also set neck = None (the pyramid maps feed the
transformer directly, ape_deta_vitl_eva02_lsj1024_cp_12ep.py:21) and leave proposal_ambiguous at 0 (ape_deta_r50.py:55-86).
"""
from .deformable_detr_segm_vl import DeformableDETRSegmVL
from .deformable_transformer_vl import (DeformableDetrTransformerDecoderVL, DeformableDetrTransformerEncoderVL,
                                        DeformableDetrTransformerVL)


class DeformableDetrTransformerEncoder(DeformableDetrTransformerEncoderVL):
    def __init__(self, embed_dim=256, num_heads=8, feedforward_dim=1024, attn_dropout=0.1, ffn_dropout=0.1, num_layers=6,
                 post_norm=False, num_feature_levels=4, use_act_checkpoint=False, pytorch_attn=False):
        super().__init__(embed_dim=embed_dim, num_heads=num_heads, feedforward_dim=feedforward_dim, attn_dropout=attn_dropout,
                         ffn_dropout=ffn_dropout, num_layers=num_layers, post_norm=post_norm, num_feature_levels=num_feature_levels,
                         vl_layer=None, use_act_checkpoint=use_act_checkpoint, pytorch_attn=pytorch_attn)


class DeformableDetrTransformerDecoder(DeformableDetrTransformerDecoderVL):
    def __init__(self, embed_dim=256, num_heads=8, feedforward_dim=1024, attn_dropout=0.1, ffn_dropout=0.1, num_layers=6,
                 return_intermediate=True, num_feature_levels=4, use_act_checkpoint=False, pytorch_attn=False):
        super().__init__(embed_dim=embed_dim, num_heads=num_heads, feedforward_dim=feedforward_dim, attn_dropout=attn_dropout,
                         ffn_dropout=ffn_dropout, num_layers=num_layers, return_intermediate=return_intermediate,
                         num_feature_levels=num_feature_levels, use_act_checkpoint=use_act_checkpoint, look_forward_twice=False,
                         pytorch_attn=pytorch_attn)


class DeformableDetrTransformer(DeformableDetrTransformerVL):
    def forward(self, multi_level_feats, multi_level_masks, multi_level_pos_embeds, query_embed=None, **kwargs):
        """reference signature (deformable_transformer.py:394-644): the VL forward without language tokens; 7 return values"""
        import torch
        B = multi_level_feats[0].shape[0]
        dummy = torch.zeros((B, 1, 1), dtype=torch.float32, device=multi_level_feats[0].device)
        out = super().forward(multi_level_feats, multi_level_masks, multi_level_pos_embeds, query_embed, query_l=dummy, **kwargs)
        return out[:7]


class DeformableDETRSegm(DeformableDETRSegmVL):
    def __init__(self, instance_on: bool = True, semantic_on: bool = False, panoptic_on: bool = False, freeze_detr=False,
                 input_shapes=[], mask_in_features=[], mask_encode_level=0, stuff_dataset_learn_thing: bool = True,
                 stuff_prob_thing: float = -1.0, test_mask_on: bool = True, semantic_post_nms: bool = True, panoptic_post_nms: bool = True,
                 aux_mask: bool = False, panoptic_configs: dict = None, **kwargs):
        # DeformableDETRSegm's own kwargs (deformable_detr_segm.py:62-85) + DeformableDETR's (**kwargs, deformable_detr.py:52-87)
        super().__init__(instance_on=instance_on, semantic_on=semantic_on, panoptic_on=panoptic_on, freeze_detr=freeze_detr,
                         input_shapes=input_shapes, mask_in_features=mask_in_features, mask_encode_level=mask_encode_level,
                         stuff_dataset_learn_thing=stuff_dataset_learn_thing, stuff_prob_thing=stuff_prob_thing,
                         name_prompt_fusion_type="none", name_prompt_fusion_text=None, test_mask_on=test_mask_on,
                         semantic_post_nms=semantic_post_nms, panoptic_post_nms=panoptic_post_nms, aux_mask=aux_mask,
                         panoptic_configs=panoptic_configs, **kwargs)
        if getattr(self.transformer.encoder, "vl_layers", None) is not None:
            raise ValueError("DeformableDETRSegm takes the plain DeformableDetrTransformer (no fusion layers); "
                             "use DeformableDETRSegmVL with DeformableDetrTransformerVL")
