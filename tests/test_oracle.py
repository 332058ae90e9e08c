"""The oracle (oracle/ape_oracle.py) against (a) the committed outputs of the reference's own code
(tests/golden/ref_*.pt, made by tests/golden/make_golden.py) and (b) the live reference when /root/reference
is present.  CPU only."""
import os

import pytest
import torch

import oracle_util as U
from oracle import ape_oracle, refshim, weights
from oracle.configs import CONFIGS

FP32_TOL = 2e-4  # two fp32 implementations of the same math (different op order)


# small_A: the plain (non-VL) family of APE-L_A/B/C -- DeformableDETRSegm on DeformableDetrTransformer, no neck, no fusion, no
# ambiguous heads, the EVA-02 MIM ViT with sub-LN (fixture produced by the reference's deformable_detr_segm.py / deformable_transformer.py)
# small_E: APE on the ViT-e backbone (post-norm blocks, packed qkv, GELU MLP, head width 112) with 3 + 3 layers
# tiny_maskprompt: a mask prompt restricts the proposals to the prompted region (and drives the selection into its fall-back list)
# small_G: the plain family on the EVA-01-CLIP ViT-g flavour (pre-norm, packed qkv, GELU MLP, head width 88)
# small_V: the plain family on the EVA-01 MIM ViT-g of vit_eva.py (decomposed relative positions)
@pytest.mark.parametrize("case", ["tiny_square", "tiny_padded", "small_padded", "tiny_phrase", "small_A", "small_E", "tiny_maskprompt", "small_G", "small_V"])
def test_oracle_matches_reference_golden(case):
    gold = U.load_golden(case)
    cfg_name, wseed, image, text = U.case_inputs(gold)
    sd = weights.make_state_dict(U.load_spec(cfg_name), wseed)
    orc = ape_oracle.ApeOracle(CONFIGS[cfg_name], sd)
    ref_topk = gold["full"]["topk_proposals"]
    prompt = U.case_prompt(gold)
    mp = U.case_mask_prompt(gold, image.shape[-2:])
    out = orc.forward(image, text, prompt=prompt, mask_prompt=mp)
    S = orc.stages
    # stages upstream of the proposal selection: elementwise
    upstream = [k for k in gold["stages"] if k.startswith(("vit_block", "last_feat", "p", "enc", "memory", "query_l",
                                                           "output_memory", "mask_features")) and not k.startswith("pred")]
    for k in upstream:
        if k in S:
            U.check_fingerprint(S[k], gold["stages"][k], FP32_TOL, k)
    # the selected proposals: same SET (near-equal scores may swap places between two fp32 implementations)
    assert set(S["topk_proposals"][0].tolist()) == set(ref_topk[0].tolist())
    # downstream with the reference's proposal order injected
    out = orc.forward(image, text, forced_topk=ref_topk, prompt=prompt, mask_prompt=mp)
    S = orc.stages
    for k in ("query_init", "query_pos", "init_reference", "inter_states", "inter_references", "pred_masks"):
        U.check_fingerprint(S[k], gold["stages"][k], 5e-4, k)
    assert U.relerr(S["pred_logits"], gold["full"]["pred_logits"]) < 1e-3
    assert U.relerr(S["pred_boxes"], gold["full"]["pred_boxes"]) < 1e-3
    frac = U.match_detections(S["det_boxes"], S["det_scores"], S["det_classes"], gold["full"]["det_boxes"],
                              gold["full"]["det_scores"], gold["full"]["det_classes"])
    assert frac >= 0.97, f"only {frac:.2%} of the reference detections reproduced"
    gi = gold["instances"]
    oi = out["instances"]
    frac = U.match_detections(oi["pred_boxes"], oi["scores"], oi["pred_classes"], gi["pred_boxes"], gi["scores"], gi["pred_classes"])
    assert frac >= 0.97
    assert list(oi["pred_masks"].shape[1:]) == gi["mask_shape"][1:]


def test_oracle_semantic_branch_matches_reference_golden():
    """a22: second NMS on the stuff scores, softmax(sigmoid/0.06), einsum with the sigmoid masks, crop + resize"""
    gold = U.load_golden("tiny_semantic")
    cfg_name, wseed, image, text = U.case_inputs(gold)
    sd = weights.make_state_dict(U.load_spec(cfg_name), wseed)
    orc = ape_oracle.ApeOracle(CONFIGS[cfg_name], sd)
    H, W = gold["out_hw"]
    out = orc.forward(image, text, forced_topk=gold["full"]["topk_proposals"], semantic=gold["semantic_meta"], height=H, width=W)
    S = orc.stages
    assert tuple(out["sem_seg"].shape) == (5, H, W)
    assert U.relerr(S["sem_box_cls"], gold["full"]["sem_box_cls"]) < 1e-3
    assert set(S["sem_query"].tolist()) == set(gold["full"]["sem_query"].tolist())
    U.check_fingerprint(out["sem_seg"], gold["stages"]["sem_seg"], 1e-3, "sem_seg")
    agree = (out["sem_seg"].argmax(0).to(torch.uint8) == gold["full"]["sem_seg_argmax"]).float().mean().item()
    assert agree > 0.999, agree


def test_oracle_eval_dataset_and_panoptic_match_reference_golden():
    """evaluation-dataset mode: detector on the thing columns only (:578-590), semantic branch, and the panoptic merge
    (:671-695, 921-998) -- against the reference run stored in ref_tiny_panoptic.pt"""
    gold = U.load_golden("tiny_panoptic")
    cfg_name, wseed, image, text = U.case_inputs(gold)
    sd = weights.make_state_dict(U.load_spec(cfg_name), wseed)
    orc = ape_oracle.ApeOracle(CONFIGS[cfg_name], sd)
    H, W = gold["out_hw"]
    meta = gold["semantic_meta"]
    out = orc.forward(image, text, forced_topk=gold["full"]["topk_proposals"], height=H, width=W, semantic=meta,
                      detector_columns=len(meta["thing_classes"]), panoptic=dict(meta=meta, cfg=gold["panoptic_cfg"]))
    S = orc.stages
    assert U.relerr(S["pred_logits"], gold["full"]["pred_logits_full"]) < 1e-3
    assert int(out["instances"]["pred_classes"].max()) < len(meta["thing_classes"])
    frac = U.match_detections(S["det_boxes"], S["det_scores"], S["det_classes"], gold["full"]["det_boxes"],
                              gold["full"]["det_scores"], gold["full"]["det_classes"])
    assert frac >= 0.97
    assert set(S["pan_query"].tolist()) == set(gold["full"]["pan_query"].tolist())
    seg, info = out["panoptic_seg"]
    ref_seg = gold["full"]["panoptic_seg"].to(torch.int32)
    assert tuple(seg.shape) == (H, W)
    assert [(d["isthing"], d["category_id"]) for d in info] == [(d["isthing"], d["category_id"]) for d in gold["segments_info"]]
    assert (seg == ref_seg).float().mean().item() > 0.999


def test_state_spec_contract():
    """checkpoint-key contract of the full-size model, as enumerated by the reference itself"""
    spec = dict(U.load_spec("L_D"))
    assert spec["model_vision.backbone.net.blocks.23.mlp.w1.weight"] == (2730, 1024)
    assert spec["model_vision.backbone.net.pos_embed"] == (1, 442, 1024)
    assert spec["model_vision.transformer.encoder.vl_layers.5.b_attn.attn.values_l_proj.weight"] == (2048, 1024)
    assert spec["model_vision.transformer.decoder.class_embed.6.weight"] == (1, 256)
    assert spec["model_vision.class_embed.6.weight"] == (1, 256)
    assert spec["model_vision.transformer.encoder.layers.0.attentions.0.sampling_offsets.weight"] == (320, 256)
    sd = weights.make_state_dict(U.load_spec("tiny"), 0)
    a = sd["model_vision.class_embed.0.bias_lang"]
    assert a is sd["model_vision.transformer.decoder.class_embed.0.bias_lang"]


@pytest.mark.skipif(not os.path.isdir("/root/reference/ape"), reason="needs the reference checkout (build container only)")
def test_oracle_matches_live_reference():
    from oracle import run_reference as rr

    image = torch.randint(0, 256, (3, 176, 256), generator=torch.Generator().manual_seed(11)).float()
    text = torch.randn(5, 1024, generator=torch.Generator().manual_seed(12))
    S, inst, spec, sd = rr.run_reference("tiny", 7, image, text, height=352, width=512)
    orc = ape_oracle.ApeOracle(CONFIGS["tiny"], sd)
    out = orc.forward(image, text, height=352, width=512, forced_topk=S["topk_proposals"])
    O = orc.stages
    for k in ("vit_block2", "p2", "p6", "enc1_out", "enc1_fused_l", "enc_class", "enc_coord_unact", "inter_states",
              "pred_logits", "pred_boxes", "pred_masks", "mask_features"):
        assert U.relerr(O[k], S[k]) < 5e-4, k
    oi = out["instances"]
    assert U.match_detections(oi["pred_boxes"], oi["scores"], oi["pred_classes"], inst["pred_boxes"], inst["scores"],
                              inst["pred_classes"]) >= 0.97
    assert oi["pred_masks"].shape == inst["pred_masks"].shape
    assert (oi["pred_masks"] != inst["pred_masks"]).float().mean().item() < 2e-3
    # RoPE tables are persistent buffers of the reference model: the oracle recomputes them
    m_sd = dict(spec)
    assert "model_vision.backbone.net.rope_win.freqs_cos" in m_sd


def test_oracle_vit_eva02_backbone_matches_reference_golden():
    """BASELINE config 1 (APE-Ti): the oracle's restatement of vit_eva02.py (packed qkv, packed SwiGLU, zero-padded 14 x 14
    windows) + SimpleFPN against the reference run of tests/golden/ref_Ti_512.pt -- backbone stages only (the transformer
    behind it is the one the other cases pin, and a full 87 296-token CPU forward would take a minute)"""
    gold = U.load_golden("Ti_512")
    cfg_name, wseed, image, text = U.case_inputs(gold)
    sd = weights.make_state_dict(U.load_spec(cfg_name), wseed)
    orc = ape_oracle.ApeOracle(CONFIGS[cfg_name], sd)
    x, _, _ = orc.preprocess(image)
    feat = orc.vit(x)
    for i in range(CONFIGS[cfg_name]["depth"]):
        U.check_fingerprint(orc.stages[f"vit_block{i}"], gold["stages"][f"vit_block{i}"], FP32_TOL, f"vit_block{i}")
    U.check_fingerprint(feat, gold["stages"]["last_feat"], FP32_TOL, "last_feat")
    for k, v in orc.fpn(feat).items():
        U.check_fingerprint(v, gold["stages"][k], FP32_TOL, k)


@pytest.mark.skipif(not refshim.available(), reason="needs /root/reference")
@pytest.mark.parametrize("case", ["tiny_padded", "tiny_maskprompt"])
def test_transformer_reference_signature_matches_live_reference(fake_ops, case):
    """`DeformableDetrTransformerVL.forward(multi_level_feats, masks, pos_embeds, query_embed, query_l, attention_mask_l,
    masks_prompt)` -> the reference's 8-tuple (deformable_transformer_vl.py:422-689).  The reference model is run on a padded
    image, the transformer's own inputs are captured by a hook and handed to the HIP-path module (ops = their definitions)."""
    from ape_amd.modeling.build import build_ape
    from oracle import run_reference as RR, weights

    gold = U.load_golden(case)
    cfg_name, wseed, image, text = U.case_inputs(gold)
    mask_prompt = None
    if case == "tiny_maskprompt":          # the predictor's inputs["mask_prompt"]: multi_level_masks_prompt is the 7th positional argument
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
        from make_golden import CASES, case_mask_prompt
        mask_prompt = case_mask_prompt(CASES[case], image.shape[-2:])
    S, _, spec, _ = RR.run_reference(cfg_name, wseed, image, text, mask_prompt=mask_prompt)
    assert (S["transformer_inputs"][6] is not None) == (mask_prompt is not None)
    model = build_ape(cfg_name)
    model.load_state_dict(weights.make_state_dict(spec, wseed), strict=False)
    mv = model.model_vision
    mv.set_compute_dtype(torch.float32)
    assert mv.transformer.compute_dtype == torch.float32
    with torch.no_grad():
        out = mv.transformer(*S["transformer_inputs"])
    names = ["inter_states", "init_reference", "inter_references", "enc_class", "enc_coord_unact", "anchors", "memory", "query_l"]
    assert len(out) == 8
    for name, got in zip(names, out):
        ref = S[name]
        assert tuple(got.shape) == tuple(ref.shape), (name, got.shape, ref.shape)
        fin = torch.isfinite(ref)
        assert torch.equal(fin, torch.isfinite(got)), name                  # +inf anchors of padded / out-of-range tokens
        e = U.relerr(got[fin], ref[fin])
        print(f"transformer.forward {name}: {e:.2e}")
        assert e < 1e-3, (name, e)


@pytest.mark.skipif(not refshim.available(), reason="needs /root/reference")
def test_rel_pos_tables_match_the_live_reference():
    """get_rel_pos "vitdet" (utils_eva.py:65-129), also when the checkpoint's table has another length (linear resize): the oracle's
    restatement and the host packing's `resized_rel_pos` (row (q - k) + size - 1 = offset q - k) against the reference's function, and
    add_decomposed_rel_pos (:132-161) against the extra-channel formulation of ape_amd/modeling/backbone/vit_eva.py on one head"""
    import importlib

    from ape_amd.modeling.backbone import vit_eva
    from oracle.ape_oracle import ApeOracle
    refshim.install()
    R = importlib.import_module("ape.modeling.backbone.utils_eva")
    g = torch.Generator().manual_seed(0)
    for size, rows in ((16, 31), (16, 27), (32, 63), (32, 127), (12, 31)):
        tbl = torch.randn(rows, 24, generator=g)
        want = R.get_rel_pos(size, size, tbl, "vitdet")                         # [size, size, C]
        got_o = ApeOracle.rel_pos_table(size, size, tbl)
        rs = vit_eva.resized_rel_pos(tbl, size)
        idx = (torch.arange(size)[:, None] - torch.arange(size)[None, :]) + size - 1
        assert torch.allclose(got_o, want, atol=1e-6) and torch.allclose(rs[idx], want, atol=1e-6), (size, rows)
    # one head, 8 x 8 tokens: scores with the reference's bias == q_ext . k_ext
    H = W = 8
    hd = 24
    q, k = torch.randn(1, H * W, hd, generator=g), torch.randn(1, H * W, hd, generator=g)
    rh, rw = torch.randn(2 * H - 1, hd, generator=g), torch.randn(2 * W - 1, hd, generator=g)
    scale = hd ** -0.5
    attn = (q * scale) @ k.transpose(-2, -1)
    want = R.add_decomposed_rel_pos(attn.clone(), q, rh, rw, (H, W), (H, W), "vitdet")[0]
    import ref_ops
    t = (q[0] @ torch.cat([rh, rw], 0).t()).contiguous()                                  # [tokens, 2H-1 + 2W-1]
    ty = (torch.arange(H * W) // W).int()
    tx = (torch.arange(H * W) % W).int()
    qe, ke = ref_ops.relpos_extend(q[0].contiguous(), k[0].contiguous(), t, ty, tx, heads=1, head_stride=hd, head_dim=hd, hk=H, wk=W,
                                   ext_dim=vit_eva.ext_width(hd, H, W), scale=scale)
    assert torch.allclose(qe @ ke.t(), want, atol=1e-5)


@pytest.mark.skipif(not refshim.available(), reason="needs /root/reference")
@pytest.mark.parametrize("variant", ["beit_bias", "layer_scale", "packed_bias"])
def test_vit_eva_host_module_matches_the_live_reference_class(fake_ops, variant):
    """ape/modeling/backbone/vit_eva.py ViT in the parameterisations its constructor offers -- q / v bias vectors (`beit_like_qkv_bias`,
    the APE config), the BEiT layer scale `gamma_1 / gamma_2` (`beit_like_gamma`: folded into the block's last linears here), a plain
    packed qkv bias -- instantiated from the reference's own file, its state dict loaded into the HIP-path class of the same name
    (ops = their definitions): same keys, same feature map"""
    import importlib
    from functools import partial

    import torch.nn as nn

    from ape_amd.modeling.backbone import vit_eva
    refshim.install()
    R = importlib.import_module("ape.modeling.backbone.vit_eva")
    kw = dict(img_size=256, patch_size=16, embed_dim=128, depth=4, num_heads=2, mlp_ratio=2.0, qkv_bias=True, drop_path_rate=0.0,
              norm_layer=partial(nn.LayerNorm, eps=1e-6), window_size=8, window_block_indexes=[0, 1, 2], residual_block_indexes=[],
              use_rel_pos=True, rel_pos_zero_init=False, out_feature="last_feat", use_act_checkpoint=False, pretrain_img_size=224,
              beit_like_qkv_bias=variant != "packed_bias", beit_like_gamma=variant == "layer_scale")
    torch.manual_seed(3)
    ref = R.ViT(**kw).eval()
    with torch.no_grad():
        for n, p in ref.named_parameters():                      # biases / layer scales / tables away from their trivial initial values
            if n.endswith(("q_bias", "v_bias", "qkv.bias", "proj.bias", "fc1.bias", "fc2.bias")):
                p.normal_(std=0.2)
            elif "gamma_" in n:
                p.uniform_(0.5, 1.5)
    ours = vit_eva.ViT(**kw)
    assert {k: tuple(v.shape) for k, v in ours.state_dict().items()} == {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    ours.load_state_dict(ref.state_dict())
    ours.compute_dtype = torch.float32
    x = torch.randn(1, 3, 256, 256)
    with torch.no_grad():
        want = ref(x)["last_feat"]
        got = ours(x)["last_feat"]
    e = U.relerr(got, want)
    assert tuple(got.shape) == tuple(want.shape) == (1, 128, 16, 16) and e < 2e-5, (variant, e)
