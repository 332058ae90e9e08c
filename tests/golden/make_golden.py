"""Generate tests/golden/* by EXECUTING THE REFERENCE (oracle/refshim.py + /root/reference) on seeded inputs.

Run in the build container only:  python tests/golden/make_golden.py
Outputs (small, committed):
  state_spec_<cfg>.json   -- the reference model's state_dict() names/shapes (checkpoint-key contract)
  ref_<case>.pt           -- per-stage fingerprints (shape, moments, 512 seeded samples) + the small head /
                             detection tensors in full, for the cases in CASES
Weights are NOT stored: oracle/weights.py regenerates them from (spec, seed) bit-identically.
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_model, run_reference as rr  # noqa: E402
from oracle.configs import CONFIGS, spec_name  # noqa: E402

CASES = {
    # name: (cfg, weight seed, image seed, (h, w), K classes, text seed)
    "tiny_square": ("tiny", 0, 2, (256, 256), 10, 3),
    "tiny_padded": ("tiny", 1, 5, (200, 144), 7, 6),
    "small_padded": ("small", 0, 2, (384, 512), 10, 3),
    # phrase prompt: the 6 class tokens + 250 zero bank slots are fused densely with the vision tokens
    "tiny_phrase": ("tiny", 2, 7, (224, 256), 6, 8, "phrase"),
    # mask prompt (deformable_detr_segm_vl.py:394-414, deformable_transformer_vl.py:356-365): only tokens inside the prompted
    # rectangle may become proposals
    "tiny_maskprompt": ("tiny", 1, 5, (200, 144), 7, 6, "name", None, None, "mask"),
    # semantic branch on (a22): 10 classes = 6 things + ("things", 4 stuff) -> 5 semantic channels; output resized x1.5
    "tiny_semantic": ("tiny", 3, 9, (208, 240), 10, 4, "name", "semantic"),
    # evaluation-dataset mode (set_eval_dataset): names from the metadata, detector on the 6 thing columns, semantic AND
    # panoptic branches on (panoptic thresholds loosened so that seeded weights produce segments)
    "tiny_panoptic": ("tiny", 3, 9, (208, 240), 10, 21, "name", "semantic", "panoptic"),
    # config 1: APE-Ti, a 512 x 512 image top-left in the mandatory 1024 square pad, 10 classes (text seed 1)
    "Ti_512": ("Ti", 0, 2, (512, 512), 10, 1),
    # ---- the BASELINE.json configurations at full size: minutes of CPU each, generated once
    # config 2: APE-L_D, 1024x1024 uint8-uniform image seed 2, 80 classes seed 3, name prompt, top-100 (COCO config)
    "L_D_coco80": ("L_D_coco", 0, 2, (1024, 1024), 80, 3),
    # config 3: 1203-class vocabulary seed 4, top-300 (the L_D config's own select_box_nums_for_evaluation)
    "L_D_lvis1203": ("L_D", 0, 2, (1024, 1024), 1203, 4),
    # config 4 flavour: a COCO-shaped (padded) image in the 1024 square
    "L_D_padded": ("L_D_coco", 0, 5, (683, 1024), 80, 3),
    # config 3 flavour: phrase prompt at full size -- 24 phrase tokens + 232 zero bank slots = L = 256 language tokens
    # fused DENSELY with the 87 296 vision tokens in every encoder layer; the fused tokens are the vocabulary (256 columns)
    "L_D_phrase256": ("L_D_coco", 0, 2, (1024, 1024), 24, 9, "phrase"),
    # f4: APE-L_A -- the plain (non-VL) model family of scripts/eval_APE-L_A.sh at full size, and a small copy of it
    "L_A_coco80": ("L_A", 0, 2, (1024, 1024), 80, 3),
    "small_A": ("small_A", 1, 4, (416, 512), 9, 5),
    # f4: APE on the ViT-e backbone (post-norm blocks, head width 112, 3 + 3 layers) at reduced size
    "small_E": ("small_E", 2, 6, (448, 512), 8, 7),
    # f4: APE on the EVA-01-CLIP ViT-g backbone (pre-norm, packed qkv, GELU MLP, head width 88) under the plain family, reduced size
    "small_G": ("small_G", 3, 8, (512, 400), 9, 5),
    # f4: APE on the EVA-01 MIM ViT-g of vit_eva.py (decomposed relative positions in window and global attention, head width 88), reduced size
    "small_V": ("small_V", 4, 11, (480, 512), 9, 5),
    # config 2 with a REAL photograph (demo/examples/*.jpg): the reference's demo image, decoded, BGR -> RGB, resized by
    # Pillow exactly as ape/engine/defaults.py:213-222 does (ResizeShortestEdge(1024, 1024): 394 x 700 -> 576 x 1024).  The JPEG
    # bytes (42 KB) travel inside the fixture; the tests decode them with the same library.
    "L_D_jpeg": ("L_D_coco", 0, "jpeg:Pisa.jpg", (576, 1024), 80, 3),
    # f4: APE on ViT-e at FULL size (64 post-norm blocks x 1792, 9 + 9 layers: ape_deta_vite_eva02_clip_vlf_lsj1024_cp_16x4_1080k_mdl_fsdp.py)
    "E_D_coco80": ("E_D", 0, 2, (1024, 1024), 80, 3),
    # f4c at FULL size: APE on the EVA-01 MIM ViT-g of vit_eva.py (40 pre-norm blocks x 1408, 16 x 16 windows, every fourth block global
    # over 4096 tokens with decomposed relative positions; plain family, 6 + 6 layers)
    "V_A_coco80": ("V_A", 0, 2, (1024, 1024), 80, 3),
    # f4b at FULL size: APE on the EVA-01-CLIP ViT-g (ape_deta_vitg_eva01_clip_lsj1536_cp_64x90k.py): 40 pre-norm blocks x 1408, 1536^2
    # (9216 ViT tokens, 32 x 32 windows), plain family
    "G_A_1536": ("G_A", 0, 2, (1536, 1536), 80, 3),
    # config 5: 1536x1536, semantic branch on (80 things + "things" + 53 stuff names -> 54 channels), top-500
    "L_D_1536_sseg": ("L_D_1536", 0, 2, (1536, 1536), 134, 3, "name", "semantic"),
}
BIG_SEMANTIC_META = {"entity": "thing+stuff", "thing_classes": [f"t{i}" for i in range(80)],
                     "stuff_classes": ["things"] + [f"s{i}" for i in range(53)]}
PANOPTIC_CFG = dict(prob=0.45, pano_temp=0.06, transform_eval=True, object_mask_threshold=0.0, overlap_threshold=0.0)
SEMANTIC_META = {"entity": "thing+stuff", "thing_classes": [f"t{i}" for i in range(6)],
                 "stuff_classes": ["things"] + [f"s{i}" for i in range(4)]}
FULL_SEM = ("sem_seg", "sem_query", "sem_box_cls")
FULL = ("pred_logits", "pred_boxes", "topk_proposals", "det_boxes", "det_scores", "det_classes", "det_query",
        "init_reference", "enc_class")


def jpeg_model_input(data, hw):
    """file bytes -> the model's `image` input as the reference's predictor builds it (defaults.py:213-222): decode, RGB, Pillow
    bilinear resize of the uint8 image (detectron2 ResizeTransform.apply_image), float32 CHW"""
    import io

    import numpy as np
    from PIL import Image
    rgb = Image.open(io.BytesIO(bytes(data))).convert("RGB")
    h, w = hw
    return torch.from_numpy(np.asarray(rgb.resize((w, h), Image.BILINEAR)).astype("float32").transpose(2, 0, 1).copy())


def make_inputs(case):
    cfg, wseed, iseed, (h, w), K, tseed = CASES[case][:6]
    if isinstance(iseed, str) and iseed.startswith("jpeg:"):
        with open(os.path.join("/root/reference/demo/examples", iseed[5:]), "rb") as fh:
            image = jpeg_model_input(fh.read(), (h, w))
    else:
        image = torch.randint(0, 256, (3, h, w), generator=torch.Generator().manual_seed(iseed)).float()
    text = torch.randn(K, 1024, generator=torch.Generator().manual_seed(tseed))
    return cfg, wseed, image, text


def case_mask_prompt(case, hw):
    """the prompt mask of a "mask" case: a rectangle over the middle of the image, 255 inside (what a user paints in the demo)"""
    if len(case) <= 9 or case[9] != "mask":
        return None
    h, w = hw
    m = torch.zeros(h, w)
    m[h // 5: (3 * h) // 5, w // 4: (3 * w) // 4] = 255.0
    return m


def fingerprint(t, nsamp=512):
    t = t.detach()
    flat = t.reshape(-1)
    idx = torch.randint(0, flat.numel(), (min(nsamp, flat.numel()),), generator=torch.Generator().manual_seed(flat.numel() % 9973))
    f = flat.float()
    fin = torch.isfinite(f)
    return {"shape": list(t.shape), "dtype": str(t.dtype), "idx": idx, "samples": flat[idx].clone(),
            "mean": f[fin].mean().item() if fin.any() else 0.0, "absmax": f[fin].abs().max().item() if fin.any() else 0.0,
            "n_nonfinite": int((~fin).sum())}


def main():
    torch.set_num_threads(8)
    only = sys.argv[1:]                      # optional: regenerate just the named cases
    for case in CASES:
        if only and case not in only:
            continue
        cfg, wseed, image, text = make_inputs(case)
        prompt = CASES[case][6] if len(CASES[case]) > 6 else "name"
        sem = (BIG_SEMANTIC_META if cfg.startswith("L_D") else SEMANTIC_META) if len(CASES[case]) > 7 and CASES[case][7] else None
        h, w = image.shape[-2:]
        out_hw = ((h, w) if cfg.startswith("L_D") else (int(1.5 * h), int(1.5 * w))) if sem else (None, None)
        pan = len(CASES[case]) > 8 and bool(CASES[case][8])
        mask_prompt = case_mask_prompt(CASES[case], image.shape[-2:])
        if pan:
            sem = dict(sem, thing_dataset_id_to_contiguous_id={i + 1: i for i in range(len(sem["thing_classes"]))})
        S, inst, spec, _ = rr.run_reference(cfg, wseed, image, text, prompt=prompt, semantic=sem, height=out_hw[0], width=out_hw[1],
                                            eval_dataset=pan, panoptic_configs=PANOPTIC_CFG if pan else None, mask_prompt=mask_prompt)
        if spec_name(cfg) == cfg:
            with open(os.path.join(HERE, f"state_spec_{cfg}.json"), "w") as fh:
                json.dump(spec, fh)
        gold = {"case": CASES[case], "stages": {}, "full": {}}
        if isinstance(CASES[case][2], str):
            with open(os.path.join("/root/reference/demo/examples", CASES[case][2][5:]), "rb") as fh:
                gold["jpeg"] = torch.frombuffer(bytearray(fh.read()), dtype=torch.uint8).clone()
        big = cfg.startswith("L_D") or cfg in ("Ti", "L_A", "E_D", "G_A", "V_A", "V_A_1536")
        for k, v in S.items():
            if torch.is_tensor(v):
                gold["stages"][k] = fingerprint(v)
                if k in FULL or (k in FULL_SEM and k != "sem_seg"):
                    gold["full"][k] = v.clone()
        if big:
            # full-size cases: keep the fixture small.  Logits of wide vocabularies: a seeded 128-column subset (the
            # fingerprint and the detections still cover every column)
            if gold["full"]["pred_logits"].shape[-1] > 256:
                cols = torch.randperm(gold["full"]["pred_logits"].shape[-1], generator=torch.Generator().manual_seed(11))[:128].sort()[0]
                gold["logit_cols"] = cols
                gold["full"]["pred_logits"] = gold["full"]["pred_logits"][..., cols].clone()
            # argmax masks: sign bits of the low-resolution mask logits of the first 100 kept detections (+ a tie mask:
            # |logit| < 1e-3 * absmax is excluded from the comparison), and the final pasted masks of the first 4 instances
            import numpy as np
            pm = S["pred_masks"][0][S["det_query"][:100]]
            gold["full"]["mask_sign_kept"] = torch.from_numpy(np.packbits((pm > 0).numpy(), axis=-1))
            gold["full"]["mask_tie_kept"] = torch.from_numpy(np.packbits((pm.abs() < 1e-3 * pm.abs().max()).numpy(), axis=-1))
            gold["full"]["final_masks4"] = torch.from_numpy(np.packbits(inst["pred_masks"][:4].bool().numpy(), axis=-1))
        if sem:
            gold["semantic_meta"], gold["out_hw"] = sem, out_hw
            lab = S["sem_seg"].argmax(0).to(torch.uint8)                                   # [H, W] labels
            gold["sem_stride"] = 4 if lab.numel() > (1 << 20) else 1                       # big maps: every 4th row / column
            gold["full"]["sem_seg_argmax"] = lab[:: gold["sem_stride"], :: gold["sem_stride"]].clone()
        if pan:
            gold["panoptic_cfg"] = PANOPTIC_CFG
            gold["full"]["panoptic_seg"] = S["panoptic_seg"].to(torch.int16)
            gold["full"]["pan_query"] = S["pan_query"].clone()
            gold["full"]["pred_logits_full"] = S["pred_logits_full"].clone()
            gold["segments_info"] = S["segments_info"]
        gold["instances"] = {"pred_boxes": inst["pred_boxes"], "scores": inst["scores"], "pred_classes": inst["pred_classes"],
                             "mask_area": inst["pred_masks"].flatten(1).sum(1), "mask_shape": list(inst["pred_masks"].shape),
                             "mask_rowsum0": inst["pred_masks"][0].sum(1) if len(inst["pred_masks"]) else None}
        torch.save(gold, os.path.join(HERE, f"ref_{case}.pt"))
        print(case, "->", len(gold["stages"]), "stages,", len(inst["scores"]), "instances")
    if only:
        return
    # checkpoint-key contract of the full-size model (no forward: 1-2 min/image on CPU)
    m = ref_model.build_reference(CONFIGS["L_D"], torch.zeros(1, 1024))
    with open(os.path.join(HERE, "state_spec_L_D.json"), "w") as fh:
        json.dump(rr.spec_of(m), fh)
    print("L_D spec:", len(m.state_dict()), "tensors,", sum(v.numel() for v in m.state_dict().values()) / 1e6, "M values")


if __name__ == "__main__":
    main()
