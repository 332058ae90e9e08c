"""-m gpu end-to-end parity: the full HIP pipeline (C-ABI kernels) vs the oracle / reference fixtures."""
import pytest
import torch

import model_util as M
import oracle_util as U

pytestmark = pytest.mark.gpu

import os

# APE_TEST_SELFCHECK=1: run the full-size parity tests on the CPU with ops := their torch definitions (validates the
# harness and the host-side composition at APE-L_D size; minutes per case)
SELF = os.environ.get("APE_TEST_SELFCHECK") == "1"
DEV = "cpu" if SELF else "cuda"
if SELF:
    import ape_amd.ops as _ops
    import ref_ops as _ref
    for _n in dir(_ref):
        if not _n.startswith("_") and callable(getattr(_ref, _n)) and hasattr(_ops, _n):
            setattr(_ops, _n, getattr(_ref, _n))


def _run(case, dtype):
    model, orc, image, text, gold = M.build_pair(case, device="cuda", dtype=dtype)
    return model, orc, image.cuda(), text.cuda(), gold, image, text


@pytest.mark.parametrize("case", ["tiny_padded", "tiny_square", "small_padded", "tiny_phrase", "small_A", "small_E", "tiny_maskprompt", "small_G", "small_V"])
def test_fp32_pipeline_matches_oracle_and_reference(case):
    """T1: every HIP kernel in its fp32 instantiation; tolerance = north_star's 1e-3 on logits / boxes"""
    model, orc, image, text, gold, image_c, text_c = _run(case, torch.float32)
    prompt = U.case_prompt(gold)           # tiny_phrase: dense multi-token fusion (phrase / expression prompts)
    mp = U.case_mask_prompt(gold, image_c.shape[-2:])      # tiny_maskprompt: proposals restricted to the prompted region
    mpd = None if mp is None else mp.cuda()
    mv = model.model_vision
    stages = {}
    mv.forward_single(image, text, stages=stages, prompt=prompt, mask_prompt=mpd)
    orc.forward(image_c, text_c, prompt=prompt, mask_prompt=mp)
    O = orc.stages
    for k in ("p2", "p4", "p6", "enc0_fused_v", "enc0_fused_l", "memory", "query_l", "output_memory", "enc_class", "enc_coord_unact"):
        if O.get(k) is None:
            continue                    # small_A: the plain family has no fusion stages
        b = M.token_major(k, O[k])
        e = U.relerr(stages[k].float().cpu().reshape(b.shape), b)
        print(f"[fp32 {case}] {k}: {e:.2e}")
        assert e < 3e-4, k
    ov = M.set_overlap(stages["topk_proposals"].cpu(), gold["full"]["topk_proposals"][0])
    print(f"[fp32 {case}] proposal overlap with the reference run: {ov:.4f}")
    # mask prompt: the fall-back list ends in exact ties among the masked tokens, which the reference run breaks by BLAS noise
    # (tests/test_host_model.py::test_mask_prompt_restricts_the_proposals): the tokens inside the region must agree
    assert ov >= (0.99 if mp is None else 0.93)
    ref_topk = gold["full"]["topk_proposals"][0]
    stages = {}
    out = mv.forward_single(image, text, forced_topk=ref_topk.cuda(), stages=stages, prompt=prompt, mask_prompt=mpd)
    el = U.relerr(stages["pred_logits"].cpu(), gold["full"]["pred_logits"][0])
    eb = U.relerr(stages["pred_boxes"].cpu(), gold["full"]["pred_boxes"][0])
    print(f"[fp32 {case}] pred_logits {el:.2e} pred_boxes {eb:.2e} (vs reference fixture)")
    assert el < 1e-3 and eb < 1e-3
    frac = U.match_detections(out["det_boxes"].cpu(), out["det_scores"].cpu(), out["det_classes"].cpu(),
                              gold["full"]["det_boxes"], gold["full"]["det_scores"], gold["full"]["det_classes"])
    print(f"[fp32 {case}] detections reproduced: {frac:.3f}")
    assert frac >= 0.97
    orc.forward(image_c, text_c, forced_topk=ref_topk[None], prompt=prompt, mask_prompt=mp)
    # masks are compared per (query, class) pair: two detections with near-equal scores may swap places
    ours = {(int(q), int(c)): i for i, (q, c) in enumerate(zip(out["det_query"].cpu(), out["det_classes"].cpu()))}
    pairs = [(ours[(int(q), int(c))], j) for j, (q, c) in enumerate(zip(orc.stages["det_query"], orc.stages["det_classes"]))
             if (int(q), int(c)) in ours]
    assert len(pairs) >= 0.97 * len(orc.stages["det_query"])
    a = out["det_masks128"].bool().cpu()[[i for i, _ in pairs]]
    b = orc.stages["det_masks128"][[j for _, j in pairs]]
    mm = (a != b).float().mean().item()
    print(f"[fp32 {case}] 128x128 mask mismatch fraction {mm:.2e} over {len(pairs)} matched detections")
    assert mm < 2e-3


@pytest.mark.parametrize("case", ["tiny_padded", "small_padded", "tiny_phrase", "small_A", "small_E", "small_G", "small_V"])
def test_bf16_pipeline(case):
    """bf16 storage + MFMA, fp32 accumulate on the small models (all prompt modes): (a) every stage, fed the fp32 pipeline's
    input (teacher forcing), is inside the tolerance derived from bf16's 8 significant bits (tests/teacher_forced.py); (b) the
    free-running pipeline vs the fp32 ORACLE: reported, and inside the same derivation applied to the whole path"""
    import teacher_forced as TF

    model, orc, image, text, gold, image_c, text_c = _run(case, torch.float32)
    ref_topk = gold["full"]["topk_proposals"][0]
    prompt = U.case_prompt(gold)
    ferr, free_err, outs = TF.run(model, image, text, ref_topk.cuda(), prompt=prompt)
    TF.report(f"bf16 {case}", ferr, free_err)
    assert len(ferr) >= 25 and not TF.violations(ferr), TF.violations(ferr)
    orc.forward(image_c, text_c, forced_topk=ref_topk[None], prompt=prompt)
    O, free = orc.stages, outs["free"]
    stages = {k: v for k, v in outs["free_stages"].items()}
    for k in ("p2", "p6", "memory", "enc_class", "pred_logits", "pred_boxes"):
        b = M.token_major(k, O[k])
        got = stages[k].float().cpu().reshape(b.shape)
        fin = torch.isfinite(b)
        # rms error relative to the rms of the reference (for the final logits that includes their constant bias log(1/99): the
        # free-running heads sit behind the decoder's iterative refinement, which amplifies input error query by query -- the
        # derivation bounds rounding accumulation, so the head statement is deliberately the weak one; per-stage: above)
        r = ((got[fin] - b[fin]).pow(2).mean().sqrt() / (b[fin] - (b[fin].mean() if k == "enc_class" else 0)).pow(2).mean().sqrt()).item()
        print(f"[bf16 {case}] free-running {k} vs fp32 oracle: max {U.relerr(got, b):.2e} rms {r:.2e} (path bound {TF.path_bound(model, k):.2e})")
        assert r < TF.path_bound(model, k), (k, r)
    assert free is not None and torch.isfinite(free["det_scores"]).all()


def test_forward_api_on_gpu():
    model, orc, image, text, gold, image_c, text_c = _run("tiny_padded", torch.float32)
    h, w = image_c.shape[-2:]
    res = model([{"image": image_c, "height": 2 * h, "width": 2 * w, "text_features": text_c}])[0]["instances"]
    oi = orc.forward(image_c, text_c, height=2 * h, width=2 * w)["instances"]
    frac = U.match_detections(res.pred_boxes, res.scores, res.pred_classes, oi["pred_boxes"], oi["scores"], oi["pred_classes"])
    assert frac >= 0.95 and res.pred_masks.shape[1:] == (2 * h, 2 * w)
    assert not res.pred_boxes.tensor.is_cuda and len(res) == len(res.scores)


def test_phrase_prompt_through_graph_runtime():
    """phrase prompt (dense fusion) through the reference entry point and through the hipGraph runtime: same detections"""
    from ape_amd.runtime import GraphedForward

    model, orc, image, text, gold, image_c, text_c = _run("tiny_phrase", torch.float32)
    h, w = image_c.shape[-2:]
    res = model([{"image": image_c, "height": h, "width": w, "text_features": text_c, "prompt": "phrase"}])[0]["instances"]
    oi = orc.forward(image_c, text_c, prompt="phrase")["instances"]
    frac = U.match_detections(res.pred_boxes, res.scores, res.pred_classes, oi["pred_boxes"], oi["scores"], oi["pred_classes"])
    print(f"[phrase] forward() vs oracle instances: {frac:.3f}")
    assert frac >= 0.95
    run = GraphedForward(model.model_vision)
    for _ in range(2):                       # second call replays the captured graph
        inst, _ = run(image, text, prompt="phrase")
    frac = U.match_detections(inst.pred_boxes, inst.scores, inst.pred_classes, res.pred_boxes, res.scores, res.pred_classes)
    print(f"[phrase] graph replay vs eager: {frac:.3f}")
    assert frac >= 0.99


def test_semantic_branch_on_gpu():
    """a22: second NMS, pixel-major sigmoid(upsample) kernel, class x query GEMM, bilinear resize -- fp32 kernels"""
    model, orc, image, text, gold, image_c, text_c = _run("tiny_semantic", torch.float32)
    M.check_semantic(model, orc, image_c, text_c, gold, "cuda")


def test_semantic_branch_in_graph_runtime():
    """GraphedForward(semantic=meta): the semantic branch captured with the step; the label map (per-pixel argmax of the
    [K', H, W] scores) equals the argmax of model.forward()'s sem_seg, plain and software-pipelined"""
    from ape_amd.runtime import GraphedForward

    model, orc, image, text, gold, image_c, text_c = _run("tiny_semantic", torch.float32)
    mv = model.model_vision
    meta = gold["semantic_meta"]
    mv.semantic_on = True
    mv.set_metadata(0, name="coco_2017_val", thing_classes=meta["thing_classes"], stuff_classes=meta["stuff_classes"])
    H, W = gold["out_hw"]
    res = model([{"image": image_c, "height": H, "width": W, "text_features": text_c}])[0]
    ref = res["sem_seg"].argmax(0).cpu()
    sem_meta = dict(mv.metadata_list[-1], entity=mv.dataset_entities[-1])
    for kw in (dict(), dict(images_per_step=2, pipeline=True)):
        run = GraphedForward(mv, semantic=sem_meta, **kw)
        B = kw.get("images_per_step", 1)
        tickets = [run.submit([image] * B if B > 1 else image, text, H, W) for _ in range(2)]      # second submit replays the graph
        for t in tickets:
            inst, _ = run.result(t)
            for lab in t.sem_labels:
                assert lab.shape == (H, W) and lab.dtype == torch.int16
                agree = (lab.long() == ref).float().mean().item()
                assert agree > 0.999, (kw, agree)
        first = inst[0] if B > 1 else inst
        frac = U.match_detections(first.pred_boxes, first.scores, first.pred_classes, res["instances"].pred_boxes, res["instances"].scores,
                                  res["instances"].pred_classes)
        assert frac >= 0.99


@pytest.mark.parametrize("pipeline", [False, True])
def test_semantic_branch_in_the_size_agnostic_graph(pipeline):
    """GraphedForward(semantic=meta, any_size=True): ONE graph serves images of different sizes (also inside one step); the class
    scores are captured over the whole pad, the crop to each image's region / resize to its frame / argmax run behind the replay with
    the ticket's sizes -- label maps equal those of model.forward() on the same image (sem_seg_postprocess, :875-918)"""
    from ape_amd.runtime import GraphedForward

    model, orc, image, text, gold, image_c, text_c = _run("tiny_semantic", torch.float32)
    mv = model.model_vision
    meta = gold["semantic_meta"]
    mv.semantic_on = True
    mv.set_metadata(0, name="coco_2017_val", thing_classes=meta["thing_classes"], stuff_classes=meta["stuff_classes"])
    sem_meta = dict(mv.metadata_list[-1], entity=mv.dataset_entities[-1])
    g = torch.Generator().manual_seed(11)
    sizes = [tuple(image.shape[-2:]), (256, 256), (128, 240), (176, 208)]
    imgs = [image] + [torch.randint(0, 256, (3, h, w), generator=g).float().cuda() for h, w in sizes[1:]]
    frames = [(h + h // 2, w + 8) for h, w in sizes]
    ref = []
    for im, (fh, fw) in zip(imgs, frames):
        res = model([{"image": im, "height": fh, "width": fw, "text_features": text}])[0]
        ref.append((res["sem_seg"].argmax(0).cpu(), res["instances"]))
    run = GraphedForward(mv, semantic=sem_meta, images_per_step=2, pipeline=pipeline, any_size=True, max_out_pixels=400 * 272)
    labels, insts, queue = [], [], []

    def take(t):
        inst, _ = run.result(t)
        insts.extend(_own(inst))
        labels.extend(lab.clone() for lab in t.sem_labels)

    for rnd_ in range(2):                                                # second round replays the captured graph
        for i in range(0, len(imgs), 2):
            queue.append(run.submit(imgs[i:i + 2], text, [f[0] for f in frames[i:i + 2]], [f[1] for f in frames[i:i + 2]]))
            if len(queue) > (2 if pipeline else 1):
                take(queue.pop(0))
    while queue:
        take(queue.pop(0))
    assert len(run._graphs) == 1 and len(labels) == 2 * len(imgs)
    for i, lab in enumerate(labels):
        want, rinst = ref[i % len(imgs)]
        assert lab.shape == want.shape and lab.dtype == torch.int16, (i, lab.shape, want.shape)
        agree = (lab.long() == want).float().mean().item()
        assert agree > 0.999, (i, agree)
        frac = U.match_detections(insts[i].pred_boxes, insts[i].scores, insts[i].pred_classes, rinst.pred_boxes, rinst.scores, rinst.pred_classes)
        assert frac >= 0.99, (i, frac)


def test_half_model_selects_the_f16_flavour():
    """model.half() -- the reference's evaluation cast (tools/train_net.py:642) -- switches every module to the IEEE-half kernels; the
    MFMA weights are then bit-identical to the f16 flavour of the fp32 model (fp32 -> f16 rounds once either way), norms / biases carry
    one extra rounding: same detections"""
    model, orc, image, text, gold, image_c, text_c = _run("tiny_padded", torch.float32)
    mv = model.model_vision
    h, w = image.shape[-2:]
    mv.set_compute_dtype(torch.float16)
    ref = model([{"image": image, "height": h, "width": w, "text_features": text}])[0]["instances"]
    mv.set_compute_dtype(torch.float32)
    model.half()
    assert mv.compute_dtype == torch.float16 and mv.backbone.net.compute_dtype == torch.float16
    got = model([{"image": image.half(), "height": h, "width": w, "text_features": text.half()}])[0]["instances"]
    frac = U.match_detections(got.pred_boxes, got.scores, got.pred_classes, ref.pred_boxes, ref.scores, ref.pred_classes,
                              box_tol=3e-2, score_tol=3e-2)
    print(f"[half model] {len(got.scores)} instances, {frac:.3f} of the f16 flavour's detections matched")
    assert frac >= 0.9, frac


def test_parallel_images_in_one_graph():
    """images_per_step = 2: two batch-1 forwards as parallel branches of one hipGraph give the same detections and masks
    as two sequential single-image replays"""
    from ape_amd.runtime import GraphedForward

    model, orc, image, text, gold, image_c, text_c = _run("tiny_padded", torch.float32)
    image2 = torch.flip(image, dims=[2]).contiguous()
    one = GraphedForward(model.model_vision)
    ref = []
    for im in (image, image2):
        inst, _ = one(im, text)
        ref.append((inst.pred_boxes.clone(), inst.scores.clone(), inst.pred_classes.clone(), inst.pred_masks.clone()))
    two = GraphedForward(model.model_vision, images_per_step=2)
    for _ in range(2):                                  # second call replays the captured graph
        insts, rec6 = two([image, image2], text)
    assert rec6.shape[0] == 2 and len(insts) == 2
    for inst, (b, s, c, m) in zip(insts, ref):
        frac = U.match_detections(inst.pred_boxes, inst.scores, inst.pred_classes, b, s, c)
        assert frac >= 0.99, frac
        assert inst.pred_masks.shape == m.shape
        assert (inst.pred_masks != m).float().mean().item() < 1e-3


def _own(insts):
    """pred_masks are views of a pinned slot that later steps overwrite: keep copies"""
    for inst in (insts if isinstance(insts, list) else [insts]):
        inst.pred_masks = inst.pred_masks.clone()
    return insts


@pytest.mark.parametrize("B", [1, 2])
def test_software_pipelined_runtime(B):
    """pipeline=True: a step's graph = ViT of the new images (batched) || tails of the previous step's images.  Tickets
    complete one submit later (or on flush) and carry the same detections / masks as the plain runtime"""
    from ape_amd.runtime import GraphedForward

    model, orc, image, text, gold, image_c, text_c = _run("tiny_padded", torch.float32)
    imgs = [image, torch.flip(image, dims=[2]).contiguous(), torch.flip(image, dims=[1]).contiguous(), (255.0 - image).contiguous()]
    plain = GraphedForward(model.model_vision)
    ref = []
    for i_, im in enumerate(imgs):
        print(f"[pipelined] plain replay {i_}", flush=True)
        inst, _ = plain(im, text)
        ref.append((inst.pred_boxes.clone(), inst.scores.clone(), inst.pred_classes.clone(), inst.pred_masks.clone()))
    piped = GraphedForward(model.model_vision, images_per_step=B, pipeline=True)
    steps = [imgs[i:i + B] for i in range(0, len(imgs), B)] * 2          # second round replays the captured graph
    got, queue = [], []
    for st in steps:
        queue.append(piped.submit(st if B > 1 else st[0], text))
        if len(queue) > 2:
            got.append(_own(piped.result(queue.pop(0))[0]))
    while queue:
        got.append(_own(piped.result(queue.pop(0))[0]))                  # the last ticket needs a flush
    flat = [g for step in got for g in (step if B > 1 else [step])]
    assert len(flat) == 2 * len(imgs)
    for i, inst in enumerate(flat):
        b, sc, c, m = ref[i % len(imgs)]
        frac = U.match_detections(inst.pred_boxes, inst.scores, inst.pred_classes, b, sc, c)
        assert frac >= 0.99, (i, frac)
        assert inst.pred_masks.shape == m.shape and (inst.pred_masks != m).float().mean().item() < 1e-3


def test_eva02_subln_backbone_on_gpu():
    """the APE-L_A/B/C backbone configuration (vit_eva02.ViT with sub-LN / naive SwiGLU) on the HIP kernels vs the reference
    run: fp32 kernels <= 1e-3 (north_star's tolerance), bf16 reported and bounded"""
    from test_host_model import _eva02_subln_case

    for dt, tol in ((torch.float32, 1e-3), (torch.bfloat16, 5e-2)):
        net, image, gold = _eva02_subln_case("cuda", dt)
        feat = net.forward_tokens(image, (120.0, 120.0, 120.0), (60.0, 60.0, 60.0))
        e = U.relerr(feat.float().cpu(), gold["last_feat"].reshape(128, -1).t())
        print(f"[eva02 sub-LN backbone {dt}] last_feat vs reference run: {e:.2e}")
        assert e < tol


def test_vite_backbone_on_gpu():
    """the ViT-e configuration (post-norm, packed qkv, GELU MLP, head width 112 -> the 128-wide flash-attention kernel) on the HIP
    kernels vs the reference run: fp32 kernels <= 1e-3 (north_star's tolerance), bf16 reported and bounded"""
    from test_host_model import _vite_case

    for dt, tol in ((torch.float32, 1e-3), (torch.bfloat16, 5e-2)):
        net, image, gold = _vite_case("cuda", dt)
        feat = net.forward_tokens(image, (120.0, 120.0, 120.0), (60.0, 60.0, 60.0))
        r2t = net.packed(dt)["r2t"].long()
        e = U.relerr(feat[r2t].float().cpu(), gold["last_feat"].reshape(224, -1).t())
        print(f"[ViT-e backbone {dt}] last_feat vs reference run: {e:.2e}")
        assert e < tol


def test_runtime_rle_mask_format_and_predictor_pipeline():
    """mask_format="rle": the runtime's device-side COCO RLE == the bitmask output encoded by the oracle; the predictor's
    device-side input pipeline (upload uint8, resize + BGR flip + float CHW in one kernel) == the reference's host pipeline"""
    import numpy as np
    from oracle import imageio as IO
    from ape_amd import evaluation
    from ape_amd.engine import DefaultPredictor
    from ape_amd.runtime import GraphedForward

    model, orc, image, text, gold, image_c, text_c = _run("tiny_padded", torch.float32)
    plain = GraphedForward(model.model_vision)
    rle = GraphedForward(model.model_vision, mask_format="rle", rle_cap=512)
    both = GraphedForward(model.model_vision, mask_format="both", rle_cap=512)     # a data-parallel rank: bitmasks to its host + runs to gather
    h, w = image.shape[-2:]
    for fh, fw in [(h, w), (2 * h + 3, w + 17)]:
        a, _ = plain(image, text, fh, fw)
        b, _ = rle(image, text, fh, fw)
        tk = both.submit(image, text, fh, fw)
        c, _ = both.result(tk)
        assert c.has("pred_masks") and c.has("pred_masks_rle") and tk.runs is not None
        assert torch.equal(c.pred_masks, a.pred_masks) and [r["counts"] for r in c.pred_masks_rle] == [r["counts"] for r in b.pred_masks_rle]
        assert len(a) == len(b) > 0 and b.has("pred_masks_rle") and not b.has("pred_masks")
        assert torch.equal(a.pred_classes, b.pred_classes)
        for i in range(len(a)):
            r = b.pred_masks_rle[i]
            ref = IO.rle_encode(a.pred_masks[i].numpy())
            assert r["size"] == [fh, fw] and r["counts"] == IO.rle_to_string(ref), i
        js = evaluation.instances_to_coco_json(b, 7)
        js2 = evaluation.instances_to_coco_json(a, 7)                       # host bitmasks -> uploaded -> encoded on the device
        assert [d["segmentation"] for d in js] == [d["segmentation"] for d in js2]
        assert js[0]["bbox"][2] >= 0 and js[0]["image_id"] == 7
    rng = np.random.default_rng(3)
    pred = DefaultPredictor(model=model, short_edge_length=h, max_size=max(h, w), input_format="RGB")
    bgr = rng.integers(0, 256, (300, 420, 3), dtype=np.uint8)
    got = pred.preprocess(bgr)
    ref = IO.predictor_input(bgr, h, max(h, w), "RGB")
    assert got.is_cuda and np.array_equal(got.cpu().numpy(), ref)


@pytest.mark.parametrize("pipeline", [False, True])
def test_any_size_runtime(pipeline):
    """any_size=True: ONE size-agnostic graph (image canvas + StaticGeometry buffers + device frame vector) serves a stream
    of different image sizes -- also different sizes inside one step -- with the results of the per-size forward"""
    from ape_amd.runtime import GraphedForward

    model, orc, image, text, gold, image_c, text_c = _run("tiny_padded", torch.float32)
    mv = model.model_vision
    g = torch.Generator().manual_seed(5)
    sizes = [(200, 144), (256, 256), (128, 256), (176, 208), (240, 96), (256, 160)]
    imgs = [torch.randint(0, 256, (3, h, w), generator=g).float().cuda() for h, w in sizes]
    frames = [(2 * h, w + 16) for h, w in sizes]                       # output frames differ from the input sizes
    ref = []
    for im, (fh, fw) in zip(imgs, frames):
        r = model([{"image": im, "height": fh, "width": fw, "text_features": text}])[0]["instances"]
        ref.append((r.pred_boxes.tensor.clone(), r.scores.clone(), r.pred_classes.clone(), r.pred_masks.clone()))
    run = GraphedForward(mv, images_per_step=2, pipeline=pipeline, any_size=True, max_out_pixels=512 * 272)
    got, queue = [], []
    for rnd_ in range(2):                                                # second round replays the captured graph
        for i in range(0, len(imgs), 2):
            queue.append(run.submit(imgs[i:i + 2], text, [f[0] for f in frames[i:i + 2]], [f[1] for f in frames[i:i + 2]]))
            if len(queue) > (2 if pipeline else 1):      # a pipelined ticket completes with the next submit
                got.extend(_own(run.result(queue.pop(0))[0]))
    while queue:
        got.extend(_own(run.result(queue.pop(0))[0]))
    assert len(run._graphs) == 1 and len(got) == 2 * len(imgs)
    for i, inst in enumerate(got):
        b, sc, c, m = ref[i % len(imgs)]
        frac = U.match_detections(inst.pred_boxes, inst.scores, inst.pred_classes, b, sc, c)
        assert frac >= 0.99, (i, frac)
        assert inst.pred_masks.shape == m.shape and (inst.pred_masks != m).float().mean().item() < 1e-3, i


def test_eval_dataset_panoptic_on_gpu():
    """evaluation-dataset mode + panoptic merge with the fp32 HIP kernels"""
    model, orc, image, text, gold, image_c, text_c = _run("tiny_panoptic", torch.float32)
    M.check_panoptic(model, orc, image_c, text_c, gold, "cuda")


@pytest.mark.parametrize("kw", [dict(), dict(images_per_step=2, pipeline=True)], ids=["plain", "pipelined2"])
def test_panoptic_merge_in_graph_runtime(kw):
    """GraphedForward(panoptic=meta): the panoptic branch AND its merge (csrc/masks.hip panoptic_*: no host round trip) captured
    with the step; `ticket.panoptic` = (panoptic_seg, segments_info) equal to model.forward()'s, which check_panoptic holds against
    the reference-generated fixture"""
    from ape_amd.runtime import GraphedForward

    model, orc, image, text, gold, image_c, text_c = _run("tiny_panoptic", torch.float32)
    M.check_panoptic(model, orc, image_c, text_c, gold, "cuda")          # evaluation-dataset mode on; forward() vs the reference fixture
    mv = model.model_vision
    H, W = gold["out_hw"]
    res = model([{"image": image_c, "height": H, "width": W}])[0]
    seg_ref, info_ref = res["panoptic_seg"]
    feats, _, prompt = mv.text_features({"image": image_c, "height": H, "width": W})
    run = GraphedForward(mv, panoptic=mv.metadata_list[0], **kw)
    B = kw.get("images_per_step", 1)
    tickets = [run.submit([image_c] * B if B > 1 else image_c, feats, H, W, prompt) for _ in range(2)]      # second submit replays the graph
    for t in tickets:
        run.result(t)
        assert len(t.panoptic) == B
        for seg, info in t.panoptic:
            assert seg.shape == (H, W) and seg.dtype == torch.int32
            assert torch.equal(seg, seg_ref.cpu()) and info == info_ref, (len(info), len(info_ref))
    # any_size: ONE size-agnostic graph; the merge (crop to the image's own region, resize to its frame, walk) runs behind the replay
    # with the ticket's sizes.  Different sizes inside one step, each against model.forward() on that image
    g = torch.Generator().manual_seed(9)
    h0, w0 = image_c.shape[-2:]
    imgs = [image_c, image_c[:, : h0 - 32, : w0 - 48].contiguous(), torch.randint(0, 256, (3, h0 - 16, w0), generator=g).float().cuda(), image_c]
    frames = [(H, W), (H - 20, W + 8), (2 * (h0 - 16), w0 + 4), (H, W)]
    want = []
    for im, (fh, fw) in zip(imgs, frames):
        r = model([{"image": im, "height": fh, "width": fw}])[0]["panoptic_seg"]
        want.append((r[0].cpu().clone(), r[1]))
    S = mv.backbone.padding_constraints["square_size"]
    run = GraphedForward(mv, panoptic=mv.metadata_list[0], any_size=True, images_per_step=2, max_out_pixels=max(fh * fw for fh, fw in frames))
    got = []
    for rnd_ in range(2):                                                # the second round replays the captured graph
        for i in range(0, len(imgs), 2):
            t = run.submit(imgs[i:i + 2], feats, [f[0] for f in frames[i:i + 2]], [f[1] for f in frames[i:i + 2]], prompt)
            run.result(t)
            got.extend((seg.clone(), info) for seg, info in t.panoptic)
    assert len(run._graphs) == 1 and len(got) == 2 * len(imgs)
    for i, (seg, info) in enumerate(got):
        seg_w, info_w = want[i % len(imgs)]
        assert seg.shape == seg_w.shape and [(d["isthing"], d["category_id"]) for d in info] == [(d["isthing"], d["category_id"]) for d in info_w], i
        assert (seg != seg_w).float().mean().item() < 1e-3, i            # the canvas path rounds the geometry constants differently


# ------------------------------------------------------------------------------------------------------------------
# The BASELINE.json configurations at full size (APE-L_D, 1024^2 / 1536^2), against fixtures produced by executing the
# reference on the same seeded inputs (tests/golden/make_golden.py: L_D_coco80 = config 2, L_D_lvis1203 = config 3,
# L_D_padded = a COCO-shaped image of config 4, L_D_1536_sseg = config 5; E_D_coco80 = APE on ViT-e at full size, 64 post-norm blocks
# of width 1792 and a 9 + 9 layer DETA; V_A_coco80 = APE on the EVA-01 MIM ViT-g of vit_eva.py at full size: 40 pre-norm blocks of width
# 1408 with decomposed relative positions, plain family; G_A_1536 = APE on the EVA-01-CLIP ViT-g at its own 1536^2).  No oracle run here:
# fixtures only.
# ------------------------------------------------------------------------------------------------------------------
LD_STAGES = ("p2", "p4", "p6", "enc0_fused_v", "enc0_out", "memory", "output_memory", "enc_class", "enc_coord_unact",
             "mask_features")


def _ld_heads(stages, gold, rms=False):
    logits = stages["pred_logits"].float().cpu()
    if "logit_cols" in gold:
        logits = logits[:, gold["logit_cols"]]
    rl, rb = gold["full"]["pred_logits"][0], gold["full"]["pred_boxes"][0]
    boxes = stages["pred_boxes"].float().cpu()
    if rms:          # root-mean-square error over all queries / classes, relative to the reference's rms
        return ((logits - rl).pow(2).mean().sqrt() / rl.pow(2).mean().sqrt()).item(), ((boxes - rb).pow(2).mean().sqrt() / rb.pow(2).mean().sqrt()).item()
    return U.relerr(logits, rl), U.relerr(boxes, rb)


def _ld_mask_sign_mismatch(stages, out, gold):
    """argmax masks: sign of the low-resolution mask logits of the kept detections vs the reference's, per query; pixels the
    reference marks as ties (|logit| < 1e-3 absmax) are excluded"""
    q_ref = gold["full"]["det_query"][:100].tolist()
    npix = stages["det_mask_logits"].shape[1]
    sign_ref = M.unpack_bits(gold["full"]["mask_sign_kept"].flatten(1), npix)
    tie_ref = M.unpack_bits(gold["full"]["mask_tie_kept"].flatten(1), npix)
    ours = {int(q): i for i, q in enumerate(out["det_query"].cpu().tolist())}
    rows = [(ours[q], j) for j, q in enumerate(q_ref) if q in ours]
    mine = (stages["det_mask_logits"].float().cpu() > 0)[[i for i, _ in rows]]
    ref, tie = sign_ref[[j for _, j in rows]], tie_ref[[j for _, j in rows]]
    bad = ((mine != ref) & ~tie).float().sum().item()
    return bad / max((~tie).float().sum().item(), 1.0), len(rows)


def _ld_mask_sign_pixels(stages, out, gold):
    """the same comparison in PIXELS: (mismatching non-tie pixels, non-tie pixels, shared detections, the largest |our logit| / absmax
    among the mismatching pixels -- how far outside the fixture's tie band (|reference logit| < 1e-3 absmax) a flipped pixel sits)"""
    q_ref = gold["full"]["det_query"][:100].tolist()
    logits = stages["det_mask_logits"].float().cpu()
    npix = logits.shape[1]
    sign_ref = M.unpack_bits(gold["full"]["mask_sign_kept"].flatten(1), npix)
    tie_ref = M.unpack_bits(gold["full"]["mask_tie_kept"].flatten(1), npix)
    ours = {int(q): i for i, q in enumerate(out["det_query"].cpu().tolist())}
    rows = [(ours[q], j) for j, q in enumerate(q_ref) if q in ours]
    mine = logits[[i for i, _ in rows]]
    ref, tie = sign_ref[[j for _, j in rows]], tie_ref[[j for _, j in rows]]
    wrong = ((mine > 0) != ref) & ~tie
    margin = float((mine.abs() / mine.abs().amax(dim=1, keepdim=True).clamp_min(1e-30))[wrong].max()) if bool(wrong.any()) else 0.0
    return int(wrong.sum()), int((~tie).sum()), len(rows), margin


@pytest.mark.parametrize("case", ["Ti_512", "L_D_coco80", "L_D_padded", "L_D_lvis1203", "L_D_1536_sseg", "L_D_phrase256", "L_A_coco80", "L_D_jpeg", "V_A_coco80", "G_A_1536"])
def test_L_D_fp32_matches_reference(case):
    """T1 at the benchmarked sizes: fp32 HIP kernels vs the reference run; north_star tolerance 1e-3 on logits / boxes,
    identical argmax masks"""
    model, image, text, gold = M.build_model(case, DEV, torch.float32)
    mv = model.model_vision
    image, text = image.to(DEV), text.to(DEV)
    prompt = U.case_prompt(gold)          # L_D_phrase256: dense fusion of L = 256 language tokens with the 87 296 vision tokens
    sem = None
    if "semantic_meta" in gold:
        meta = gold["semantic_meta"]
        mv.semantic_on = True
        mv.set_metadata(0, name="coco_2017_val", thing_classes=meta["thing_classes"], stuff_classes=meta["stuff_classes"])
        sem = dict(mv.metadata_list[-1], entity=mv.dataset_entities[-1])
    stages = {}
    own = mv.forward_single(image, text, stages=stages, prompt=prompt)        # own proposal selection
    if "mask_sign_kept" in gold["full"] and "det_mask_logits" in stages:
        # north_star: "identical argmax masks" -- fp32 kernels, OWN proposal selection, the kept detections' mask logits against the
        # reference run's sign bits, pixel by pixel.  Excluded: the pixels the FIXTURE marks as ties (|reference logit| < 1e-3 of the
        # mask's largest).  Measured on MI355X (profiles/r05_*): 0 of 6.53 million non-tie pixels differ over all 100 kept detections at
        # L_D_coco80, L_D_padded and Ti_512 -- so the assertion is ZERO for the hot path's own configurations.  The widened families and the
        # phrase / LVIS / 1536^2 cases keep the fractional bound (< 1e-4) with the flipped pixels required to be near-ties (|logit| within
        # 5e-3 of zero relative to the mask's largest), count and margin printed.
        bad, total, shared, margin = _ld_mask_sign_pixels(stages, own, gold)
        print(f"[L_D fp32 {case}] argmax masks with own proposal selection: {bad} of {total} non-tie pixels differ over {shared} shared "
              f"detections; largest |logit| / absmax among them {margin:.2e}")
        assert shared >= 95 and bad <= 1e-4 * total and margin < 5e-3, (bad, total, shared, margin)
        if case in ("L_D_coco80", "L_D_padded", "Ti_512"):
            assert bad == 0 and shared == 100, (bad, shared)
    for k in LD_STAGES:
        if k not in gold["stages"]:
            continue                    # L_A_coco80 (plain family): no fusion stage
        fp = gold["stages"][k]
        e = U.check_fingerprint(M.ref_layout(k, stages[k].float(), fp["shape"]), fp, 1e-3, k)
        print(f"[L_D fp32 {case}] {k}: {e:.2e}")
    ov = M.set_overlap(stages["topk_proposals"].cpu(), gold["full"]["topk_proposals"][0])
    print(f"[L_D fp32 {case}] proposal overlap with the reference run: {ov:.4f}")
    assert ov >= 0.99
    ref_topk = gold["full"]["topk_proposals"][0]
    stages = {}
    # free-text prompt (dataset_id = -1): the detector sees every class column (:578-592)
    out = mv.forward_single(image, text, forced_topk=ref_topk.to(DEV), stages=stages, semantic=sem, prompt=prompt)
    el, eb = _ld_heads(stages, gold)
    em = U.check_fingerprint(stages["mask_embed"].float().reshape(gold["stages"]["mask_embed"]["shape"]), gold["stages"]["mask_embed"], 1e-3, "mask_embed")
    print(f"[L_D fp32 {case}] pred_logits {el:.2e} pred_boxes {eb:.2e} mask_embed {em:.2e} (vs reference fixture, tolerance 1e-3)")
    assert el < 1e-3 and eb < 1e-3
    frac = U.match_detections(out["det_boxes"].cpu(), out["det_scores"].cpu(), out["det_classes"].cpu(),
                              gold["full"]["det_boxes"], gold["full"]["det_scores"], gold["full"]["det_classes"])
    print(f"[L_D fp32 {case}] detections reproduced: {frac:.3f} of {len(gold['full']['det_scores'])}")
    assert frac >= 0.97
    mm, n = _ld_mask_sign_mismatch(stages, out, gold)
    print(f"[L_D fp32 {case}] mask-logit sign mismatch {mm:.2e} over {n} kept detections")
    assert n >= 95 and mm < 1e-4
    # final pasted masks of the first detections (detector_postprocess), when the ordering agrees
    inst = mv.postprocess_instance(out, tuple(image.shape[-2:]), image.shape[-2], image.shape[-1])
    want = M.unpack_bits(gold["full"]["final_masks4"], image.shape[-1])
    same_order = torch.equal(inst.query_index[:4], gold["full"]["det_query"][:4]) and torch.equal(inst.pred_classes[:4], gold["full"]["det_classes"][:4])
    if same_order:
        mm = (inst.pred_masks[:4] != want).float().mean().item()
        print(f"[L_D fp32 {case}] final mask mismatch (first 4 instances): {mm:.2e}")
        assert mm < 1e-3
    areas = inst.pred_masks.flatten(1).sum(1).float()
    if len(areas) == len(gold["instances"]["mask_area"]) and torch.equal(inst.query_index, gold["full"]["det_query"]):
        want_a = gold["instances"]["mask_area"].float()
        rel = ((areas - want_a).abs().sum() / want_a.sum().clamp_min(1.0)).item()
        print(f"[L_D fp32 {case}] mask-area difference summed over {len(areas)} instances / total area: {rel:.2e}")
        # the boxes agree to ~1e-4 of the image (asserted above), which moves the paste grid by a few hundredths of a pixel:
        # smooth masks change by < 1e-4 of their area; the phrase fixture's masks are pixel noise (256 random vocabulary
        # columns on random weights), where every boundary pixel of the 128 x 128 -> box resampling can flip (measured 2.7e-3,
        # identically with the torch definitions on the CPU).  The sharp statements are the two above: mask-logit signs
        # identical, the first pasted masks within 1e-3.
        assert rel < (5e-3 if prompt != "name" else 1e-3)
    if sem:
        st = gold.get("sem_stride", 1)
        lab = out["sem_seg"].argmax(0).to(torch.uint8).cpu()[::st, ::st]
        agree = (lab == gold["full"]["sem_seg_argmax"]).float().mean().item()
        print(f"[L_D fp32 {case}] semantic label agreement with the reference run: {agree:.5f}")
        assert agree > 0.999


# ------------------------------------------------------------------------------------------------------------------
# The benchmarked arithmetic (bf16 storage, fp32 accumulate, half offsets) FREE-RUNNING at the benchmarked size vs the fp32
# reference fixture.  The per-stage statement -- every stage inside the tolerance derived from bf16's 8 significant bits --
# is tests/test_teacher_forced.py; here the accumulated error is REPORTED (max-norm and rms; profiles/r03_*) and bounded by
# the same derivation applied to the whole path: R_path roundings in sequence give a relative rms error <= GAIN * U_RMS *
# sqrt(R_path) (tests/teacher_forced.py).  R_path: ViT 24 x 12, pyramid 12, neck 4, encoder 6 x 15 -> memory 394; + two-stage
# heads / query init 4, decoder 6 x 21, head 3 -> logits 527.  That bound is loose (residual streams dilute each block's error:
# measured p2 4.7e-3, memory 7e-3, logits rms 3.7e-3, boxes rms 1.1e-2) but it is not fitted to one input.  The max-norm of the
# head errors is a single-query statistic of the decoder's refinement on a random-weight model (error trace:
# profiles/r03_bf16_error_trace.log) and is reported, not asserted; detection-level agreement is asserted as a SANITY floor only
# -- the parity claim at detection level is the box AP that bench.py prints (bf16 vs fp32 detections over 16 images).
# ------------------------------------------------------------------------------------------------------------------
import math

import teacher_forced as TF

R_PATH = ("p2", "memory", "pred_logits", "pred_boxes")


def test_predictor_input_pipeline_on_the_real_photograph():
    """demo/examples/Pisa.jpg through ape_amd.engine.DefaultPredictor.preprocess (upload of the ORIGINAL BGR bytes + the resize
    kernel) == the model input the reference's predictor builds on the host (decode, RGB, Pillow resize, float CHW: the image the
    `L_D_jpeg` fixture was generated from) -- bit exact"""
    import io

    import numpy as np
    from PIL import Image
    from ape_amd.engine import DefaultPredictor
    gold = U.load_golden("L_D_jpeg")
    _, _, image, _ = U.case_inputs(gold)
    rgb = np.asarray(Image.open(io.BytesIO(gold["jpeg"].numpy().tobytes())).convert("RGB"))
    bgr = np.ascontiguousarray(rgb[:, :, ::-1])                                 # what cv2.imread hands the predictor
    pred = DefaultPredictor(model=torch.nn.Linear(1, 1).to(DEV), short_edge_length=1024, max_size=1024, input_format="RGB")
    got = pred.preprocess(bgr)
    assert tuple(got.shape) == tuple(image.shape) == (3, 576, 1024)
    assert torch.equal(got.cpu(), image)
    # a mask prompt takes the same transform (defaults.py:226-228): Pillow's bilinear resize of the single-channel uint8 mask
    m = np.zeros(rgb.shape[:2], dtype=np.uint8)
    m[80:250, 150:520] = 255
    want = torch.from_numpy(np.asarray(Image.fromarray(m).resize((1024, 576), Image.BILINEAR)).astype("float32"))
    assert torch.equal(pred.preprocess_mask(m, 576, 1024).cpu(), want)


def test_E_D_full_size_fp32_vs_reference_through_the_chaotic_stack():
    """APE on ViT-e at FULL size (64 post-norm blocks x 1792, 4.5 B seeded parameters) vs the reference run (ref_E_D_coco80.pt).
    With RANDOM weights this 64-block post-norm stack is chaotic: in the CPU oracle itself a 9e-8 (rms) perturbation of the input grows
    x 1.32 per block -- 2.5e-7 after block 0, 8.9e-6 after 10, 2.5e-3 after 30, 0.2 after 62 (measured with oracle/ape_oracle.py, DESIGN
    section 5) -- so two correct fp32 implementations that round in a different order agree in the first blocks only, and nothing
    after the backbone is comparable between ANY two implementations.  Asserted: the fp32 HIP path is inside 1e-3 for the first 16
    blocks and its distance from the reference run grows no faster than that of the reference's own arithmetic under perturbation
    (<= 2e-6 * 1.40^i).  What the rest of the model does on this backbone is covered per stage by the teacher-forced tests
    (tests/test_teacher_forced.py) and at reduced depth by small_E."""
    model, image, text, gold = M.build_model("E_D_coco80", DEV, torch.float32)
    mv = model.model_vision
    stages = {}
    mv.forward_single(image.to(DEV), text.to(DEV), stages=stages, prompt=U.case_prompt(gold))
    r2t = mv.backbone.net.packed(torch.float32)["r2t"].long()
    errs = []
    for i in range(64):
        fp = gold["stages"][f"vit_block{i}"]
        got = stages[f"vit_blk{i}"].float()[r2t].reshape(-1)[fp["idx"]].cpu()
        want = fp["samples"].float()
        errs.append(((got - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item())
    print("[E_D fp32] rms distance from the reference run after block 0, 4, 8, ...: " + " ".join(f"{e:.1e}" for e in errs[::4]))
    rate = (errs[40] / errs[8]) ** (1 / 32)
    print(f"[E_D fp32] growth per block over blocks 8..40: x {rate:.3f} (the oracle under a 9e-8 input perturbation: x 1.32)")
    assert max(errs[:16]) < 1e-3, errs[:16]
    for i, e in enumerate(errs):
        assert e < max(2e-6 * 1.40 ** i, 2e-6) or e < 0.5, (i, e)                  # saturates near the decorrelation level
    assert 1.15 < rate < 1.45, rate


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("case", ["L_D_coco80", "L_D_lvis1203", "L_D_padded", "L_D_phrase256", "L_A_coco80", "L_D_jpeg", "V_A_coco80", "G_A_1536"])
def test_L_D_bf16_pipeline(case, dt):
    tag = "bf16" if dt == torch.bfloat16 else "f16"
    model, image, text, gold = M.build_model(case, DEV, dt)
    mv = model.model_vision
    image, text = image.to(DEV), text.to(DEV)
    prompt = U.case_prompt(gold)
    ref_topk = gold["full"]["topk_proposals"][0].to(DEV)
    stages = {}
    out = mv.forward_single(image, text, forced_topk=ref_topk, stages=stages, prompt=prompt)
    t3 = {}
    for k in ("p2", "memory"):
        fp = gold["stages"][k]
        got = M.ref_layout(k, stages[k].float(), fp["shape"]).reshape(-1)[fp["idx"]].cpu()
        ref = fp["samples"].float()
        t3[k] = (((got - ref).abs().max() / fp["absmax"]).item(), ((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item())
    mx_l, mx_b = _ld_heads(stages, gold)
    rms_l, rms_b = _ld_heads(stages, gold, rms=True)
    frac = U.match_detections(out["det_boxes"].cpu(), out["det_scores"].cpu(), out["det_classes"].cpu(),
                              gold["full"]["det_boxes"], gold["full"]["det_scores"], gold["full"]["det_classes"],
                              box_tol=5e-2, score_tol=5e-2)
    mm, n = _ld_mask_sign_mismatch(stages, out, gold)
    print(f"[L_D {tag} {case}] free-running vs the fp32 reference fixture (max / rms): "
          + ", ".join(f"{k} {a:.2e} / {b:.2e}" for k, (a, b) in t3.items())
          + f", pred_logits {mx_l:.2e} / {rms_l:.2e}, pred_boxes {mx_b:.2e} / {rms_b:.2e}; detections matched (box 5%, score 0.05): "
            f"{frac:.3f}; mask sign mismatch {mm:.2e} over {n} shared detections; path bounds: "
          + ", ".join(f"{k} {TF.path_bound(model, k, dt):.2e}" for k in R_PATH))
    # ceilings: the derived path bounds; regression pins: 1.5 x the committed MI355X measurement (tests/golden/stage_pins.json)
    for k, (_, r) in t3.items():
        assert r < TF.path_bound(model, k, dt), (k, r)
    assert rms_l < TF.path_bound(model, "pred_logits", dt) and rms_b < TF.path_bound(model, "pred_boxes", dt), (rms_l, rms_b)
    assert math.isfinite(mx_l) and math.isfinite(mx_b)
    M.check_pins(f"pipeline/{case}/{tag}", {"p2_rms": t3["p2"][1], "memory_rms": t3["memory"][1], "pred_logits_rms": rms_l, "pred_boxes_rms": rms_b,
                                            "mask_sign_mismatch": mm, "detections_unmatched": 1.0 - frac})
    assert n >= 50
    if dt == torch.float16:
        # the reference's own evaluation precision: memory <= 2e-3 rms (measured 1.3-1.7e-3; bf16: 1.0-1.4e-2); the heads stay at
        # 1.4-1.7e-3 / 3.6-6.0e-3 (bf16: 3.5-4.5e-3 / 1.1-2.0e-2) -- every decoder layer's OWN error is 1/8 of bf16's (teacher-forced
        # table), but the free-running refinement of this random-weight decoder multiplies whatever reaches it by ~1.7 per layer
        assert t3["memory"][1] < 2e-3 and rms_l < 2.5e-3, (t3["memory"], rms_l)
        assert frac >= 0.85 and mm < 8e-3, (frac, mm)
    # bf16: no absolute floor to fall back to -- every end-to-end quantity above (incl. mask_sign_mismatch and detections_unmatched) is
    # held to 1.5 x its committed MI355X measurement by check_pins, which also fails when the group has no committed pins
