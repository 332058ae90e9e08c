"""Model-size configurations shared by the oracle, the reference runner and the tests.

"L_D" is APE-L_D (configs/common/backbone/vitl_eva02_clip.py:9-48 + the ape_deta L_D config); "tiny" and
"small" keep every structural feature (windowed + global RoPE attention, sub-LN, SwiGLU, SimpleFPN, 5 levels,
VL fusion, two-stage selection with ambiguous heads, 8x32 deformable attention) at sizes a CPU finishes in
seconds.
"""

CONFIGS = {
    # 16x16 tokens, windows of 8x8, 5456 encoder tokens
    "tiny": dict(img_size=256, embed_dim=128, depth=3, num_heads=2, window_size=8, pretrain_img_size=112,
                 enc_layers=2, dec_layers=2, num_queries=300, topk_eval=50),
    # 32x32 tokens, windows of 16x16, 21824 encoder tokens
    "small": dict(img_size=512, embed_dim=256, depth=6, num_heads=4, window_size=16, pretrain_img_size=224,
                  enc_layers=3, dec_layers=3, num_queries=900, topk_eval=100),
    # select_box_nums_for_evaluation: 300 in the APE-L_D joint config (ape_deta_vitl_eva02_clip_vlf_lsj1024_cp_16x4_1080k.py:108,
    # LVIS-1203 evaluation), 100 in the COCO config scripts/eval_APE-L_D.sh also runs (ape_deta_r50.py:121) -> "L_D_coco"
    "L_D": dict(img_size=1024, embed_dim=1024, depth=24, num_heads=16, window_size=32, pretrain_img_size=336,
                enc_layers=6, dec_layers=6, num_queries=900, topk_eval=300),
    "L_D_coco": dict(img_size=1024, embed_dim=1024, depth=24, num_heads=16, window_size=32, pretrain_img_size=336,
                     enc_layers=6, dec_layers=6, num_queries=900, topk_eval=100, spec="L_D"),
    # config 1: APE-Ti (configs/common/backbone/vitt_eva02.py:10-41 + ape_deta_vitt_eva02_vlf_lsj1024_cp_16x4_1080k.py): the
    # EVA-02 MIM ViT-Ti of vit_eva02.py, 14 x 14 windows on the 64 x 64 grid (zero-padded to 70 x 70), packed SwiGLU
    "Ti": dict(img_size=1024, embed_dim=192, depth=12, num_heads=3, window_size=14, pretrain_img_size=224,
               enc_layers=6, dec_layers=6, num_queries=900, topk_eval=300, backbone="eva02"),
    # APE-L_A / L_B / L_C (scripts/eval_APE-L_A.sh): the EVA-02 MIM ViT-L of vit_eva02.py in its sub-LN / naive-SwiGLU configuration
    # (configs/common/backbone/vitl_eva02.py:10-41: 16 x 16 windows, every sixth block global), NO neck (the pyramid maps feed the
    # transformer directly), the plain DeformableDETRSegm / DeformableDetrTransformer (no vision-language fusion, no ambiguous heads:
    # configs/COCO_InstanceSegmentation/ape_deta/models/ape_deta_r50.py:24-137 + ape_deta_vitl_eva02_lsj1024_cp_12ep.py:19-33), top-300
    "L_A": dict(img_size=1024, embed_dim=1024, depth=24, num_heads=16, window_size=16, pretrain_img_size=224,
                enc_layers=6, dec_layers=6, num_queries=900, topk_eval=300, backbone="eva02", subln=True, global_every=6, vl=False),
    "small_A": dict(img_size=512, embed_dim=256, depth=6, num_heads=4, window_size=16, pretrain_img_size=224,
                    enc_layers=2, dec_layers=2, num_queries=300, topk_eval=50, backbone="eva02", subln=True, global_every=3, vl=False),
    # APE with the ViT-e backbone (configs/.../ape_deta_vite_eva02_clip_vlf_lsj1024_cp_16x4_1080k_mdl_fsdp.py:24,65-66 +
    # configs/common/backbone/vite_eva02_clip_1024.py:9-49): EVA-02-CLIP ViT-e -- 64 post-norm blocks of width 1792 (16 heads of
    # 112), packed qkv, GELU MLP (ratio 8.5714), no rope, every fourth block global -- in front of a 9 + 9 layer DETA; and a small
    # copy that keeps the head width of 112 and a layer count other than 6
    # the EVA-01 MIM ViT-g of vit_eva.py (configs/common/backbone/vitg_eva01.py / vitg_eva01_1536.py under ape_deta_vitg_eva01_lsj1536_cp_64x90k.py,
    # plain model family): pre-norm, packed qkv with q / v bias, GELU MLP, DECOMPOSED RELATIVE POSITIONS in every attention (16 x 16
    # windows, every fourth block global), 16 heads x 88.  small_V keeps the head width (88) on a 32 x 32 grid
    "small_V": dict(img_size=512, embed_dim=352, depth=4, num_heads=4, window_size=16, pretrain_img_size=224, enc_layers=2, dec_layers=2,
                    num_queries=300, topk_eval=50, backbone="eva01", global_every=4, vl=False),
    "V_A": dict(img_size=1024, embed_dim=1408, depth=40, num_heads=16, window_size=16, pretrain_img_size=224, enc_layers=6, dec_layers=6,
                num_queries=900, topk_eval=300, backbone="eva01", global_every=4, vl=False),
    "E_D": dict(img_size=1024, embed_dim=1792, depth=64, num_heads=16, window_size=32, pretrain_img_size=224,
                enc_layers=9, dec_layers=9, num_queries=900, topk_eval=300, backbone="clip_e", global_every=4),
    "small_E": dict(img_size=512, embed_dim=224, depth=4, num_heads=2, window_size=16, pretrain_img_size=224,
                    enc_layers=3, dec_layers=3, num_queries=300, topk_eval=50, backbone="clip_e", global_every=4),
    # APE on the EVA-01-CLIP ViT-g (configs/COCO_InstanceSegmentation/ape_deta/ape_deta_vitg_eva01_clip_lsj1536_cp_64x90k.py +
    # configs/common/backbone/vitg_eva01_clip_1536.py): the vit_eva_clip classes with packed qkv, GELU MLP (ratio 6144 / 1408), no
    # rope, PRE-norm, 40 blocks of width 1408 = 16 heads x 88, every fourth block global, under the plain model family (neck = None)
    "G_A": dict(img_size=1536, embed_dim=1408, depth=40, num_heads=16, window_size=32, pretrain_img_size=224,
                enc_layers=6, dec_layers=6, num_queries=900, topk_eval=300, backbone="clip_g", global_every=4, vl=False),
    "small_G": dict(img_size=512, embed_dim=352, depth=4, num_heads=4, window_size=16, pretrain_img_size=224,
                    enc_layers=2, dec_layers=2, num_queries=300, topk_eval=50, backbone="clip_g", global_every=4, vl=False),
    "L_D_1536": dict(img_size=1536, embed_dim=1024, depth=24, num_heads=16, window_size=32, pretrain_img_size=336,
                     enc_layers=6, dec_layers=6, num_queries=900, topk_eval=500, spec="L_D"),
}


def spec_name(cfg_name):
    """state_spec_<name>.json that holds the state-dict contract of this configuration (top-k variants share one)"""
    return CONFIGS[cfg_name].get("spec", cfg_name)


def window_block_indexes(depth):
    """every third block is global: windowed = {0,1,3,4,...} (vitl_eva02_clip.py:21-28)"""
    return [i for i in range(depth) if i % 3 != 2]


def swiglu_hidden(embed_dim):
    return int(embed_dim * (4 * 2 / 3))  # vit_eva_clip.py:450: int(dim * mlp_ratio) -> 2730 for 1024
