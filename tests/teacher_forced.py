"""Teacher-forced per-stage check of the bf16 pipeline (the arithmetic bench.py times) against the fp32 pipeline.

The fp32-kernel run of the SAME model on the SAME input is the teacher: it is pinned to the reference's fixtures by
test_L_D_fp32_matches_reference (stages <= 2e-6 ... 1e-3, logits / boxes <= 1e-3), so its stage tensors are the reference's
fp32 tensors.  The bf16 run is teacher forced (ape_amd/stagetap.py): every stage (ViT block, pyramid map, encoder layer,
decoder layer, head) starts from the teacher's input, rounded to the stage's storage dtype exactly as the product rounds it,
and its output is compared with the teacher's output -- one stage's own error, not the accumulation of 60 stages.

Tolerances are DERIVED, not fitted.  bf16 keeps 8 significant bits: one round-to-nearest has relative error <= 2^-8 and
rms 2^-8 / sqrt(3) = 2.26e-3 (uniform mantissa).  All accumulation is fp32, so a contraction of ANY length over operands with
independent relative errors of rms u has output error rms ~ u * sqrt(sum_k (a_k w_k)^2) -- relative to an output of rms
sqrt(sum_k (a_k w_k)^2) that is u, independent of K (cancellation inside the sum is what GAIN allows for).  A stage whose
longest input -> output path crosses R bf16 roundings (one per tensor stored in bf16, one per bf16 weight matrix) therefore
has relative rms error <= GAIN * U_RMS * sqrt(R), with GAIN = 2 for the places where a normalisation or a softmax divides by
something smaller than the stream it was computed from.  R per stage is listed in ROUNDINGS with what was counted.
Box-like outputs are sigmoid(delta + logit(ref)): |d sigmoid| <= |d delta| / 4, so their ABSOLUTE rms tolerance is
tol(delta's R) * rms(delta) / 4.  Max-norm errors are reported next to the rms ones (over N elements a Gaussian error reaches
sqrt(2 ln N) ~ 5.7 sigma at N = 1e7) and asserted at 8 sigma.
"""
import math
import re

import torch

from ape_amd.stagetap import StageTap, rel_max, rel_rms

U_RMS = 2.0 ** -8 / math.sqrt(3.0)
# the same derivation for the IEEE-half flavour of the pipeline (11 significant bits; the reference's own evaluation dtype)
U_RMS_BY_DTYPE = {torch.bfloat16: 2.0 ** -8 / math.sqrt(3.0), torch.float16: 2.0 ** -11 / math.sqrt(3.0)}
GAIN = 2.0
SIGMA_MAX = 8.0

# stage key (regex) -> (R, what is counted)
ROUNDINGS = [
    (r"vit_embed$", 2, "patch pixels stored bf16, patch-embed weight"),
    (r"vit_blk\d+$", 12, "LN1 out, Wqk, q|k store, P, attention out, inner LN out, Wproj, LN2 out, W12, SwiGLU out, W3' (+ bf16 store of the last block)"),
    (r"p2$", 12, "ViT feature, deconv W + store, LN/GELU store, deconv W + store, 1x1 W + store, LN store, 3x3 W + store, LN store"),
    (r"p3$", 9, "ViT feature, deconv W + store, 1x1 W + store + LN store, 3x3 W + store + LN store"),
    (r"p[456]$", 7, "ViT feature, 1x1 W + store + LN store, 3x3 W + store + LN store"),
    (r"enc_input$", 4, "pyramid map, neck W, store, GroupNorm store"),
    (r"enc\d+_out$", 15, "x, fused v store, Wval + value store, q+pos store, Woff, sampler out, Wout + store, LN store, W1 + store, W2 + store, LN store"),
    (r"enc\d+_fused_l$", 6, "x, LN store, score W, pooled values, value / output projections of the language side"),
    (r"memory$", 1, "cast of the teacher's last encoder output"),
    (r"output_memory$", 4, "memory, Wenc, store, LN store"),
    (r"enc_cls2$", 2, "output_memory, Wcls (fp32 output): the main | ambiguous logits BEFORE the per-token max"),
    (r"enc_delta8$", 6, "output_memory, W1 + store, W2 + store, W3 (fp32 output): the main | ambiguous box deltas BEFORE the selection"),
    (r"enc_class$", 2, "max of the two logits of enc_cls2 (continuous)"),
    (r"query_init$", 4, "gathered output_memory, Wpix, position embedding store, query store"),
    (r"query_pos$", 3, "position embedding store, Wpos, query_pos store"),
    (r"dec\d+_out$", 21, "query, q+pos store, Wqk + store, P, attention out, Wo + store, LN store, Woff, memory value (Wval + store), "
                         "sampler out, Wout + store, LN store, W1 + store, W2 + store, LN store"),
    (r"dec\d+_delta$", 6, "query, W1 + store, W2 + store, W3 (fp32 output)"),
    (r"pred_logits$", 3, "query, text tokens store, (fp32 output, fp32 bias / scale)"),
    (r"mask_features$", 10, "p2, lateral W + store, GN(+memory) store, 3x3 W + store, GN store, 1x1 W + store"),
    (r"mask_embed$", 7, "query, (W + store) x 3"),
]
BOX_KEYS = re.compile(r"(dec\d+_ref|pred_boxes|init_reference)$")
# Classifier logits = <x, w> + bias with a large constant bias (the prior-probability initialisation log(1/99) = -4.6, reference
# vision_language_align.py:13-19 / deformable_detr.py:119): the rounding error of the dot product scales with sqrt(sum (x_k w_k)^2),
# i.e. with the rms of the logit WITHOUT its bias, so that is the denominator (the bias parameter is read from the model; a
# denominator that also removed the common mode of <x, w> over the tokens would understate the terms that were rounded).
# enc_coord_unact is NOT in the table: per token it is the box of whichever of the two heads has the larger logit
# (deformable_transformer_vl.py:508-533) -- a discrete choice that a rounding can flip; its continuous inputs enc_cls2 / enc_delta8
# are checked instead and the number of flipped tokens is reported.
BIASED = re.compile(r"(pred_logits|enc_class|enc_cls2)$")


def head_biases(model):
    """the constant biases of the classifier logits (see BIASED)"""
    mv = model.model_vision
    dec = mv.transformer.decoder
    nd = dec.num_layers
    amb = dec.class_embed_ambiguous[0] if hasattr(dec, "class_embed_ambiguous") else dec.class_embed[nd]     # plain family: one head
    b_enc = 0.5 * (float(dec.class_embed[nd].bias.detach().float().mean()) + float(amb.bias.detach().float().mean()))
    return {"enc_class": b_enc, "enc_cls2": b_enc, "pred_logits": float(mv.class_embed[nd - 1].bias0.detach().float().mean())}


def roundings(key):
    for pat, r, _ in ROUNDINGS:
        if re.match(pat, key):
            return r
    return None


def tolerance(key, dt=torch.bfloat16):
    r = roundings(key)
    return None if r is None else GAIN * U_RMS_BY_DTYPE[dt] * math.sqrt(r)


def _f(t):
    return t.detach().float()


def linear_head_scales(model, teacher):
    """For a head that is ONE linear map of a teacher-forced input (enc_cls2 / enc_class = output_memory . Wcls^T + b), the first-order
    rounding model can be evaluated exactly instead of through the output's rms: y = sum_k x_k w_k with independent relative errors
    of rms u on every x_k and w_k has error rms u sqrt(2) sqrt(sum_k (x_k w_k)^2).  The term norm sqrt(sum_k (x_k w_k)^2) is the right
    denominator -- the output itself can be much smaller than its terms (cancellation: measured rms(y - b) / rms(term norm) ~ 0.5
    for this head on the seeded weights).  -> {key: rms of the term norm over the tokens}"""
    if "output_memory" not in teacher:
        return {}
    dec = model.model_vision.transformer.decoder
    nd = dec.num_layers
    amb = dec.class_embed_ambiguous[0] if hasattr(dec, "class_embed_ambiguous") else dec.class_embed[nd]
    W = torch.cat([dec.class_embed[nd].weight.detach().float(), amb.weight.detach().float()], 0)                           # [2, 256]
    x = teacher["output_memory"].detach().float()
    tn = ((x * x) @ (W * W).t().to(x.device)).sqrt()                                                                       # [T, 2]
    s = float(tn.pow(2).mean().sqrt())
    return {"enc_cls2": s, "enc_class": s}


def stage_errors(got, teacher, biases=None, scales=None, dt=torch.bfloat16):
    """-> {key: dict(rms, max, tol, tol_max, R, n)} for every key the teacher-forced run tapped and the table knows.
    scales: {key: absolute rms scale} replaces the denominator rms(ref) (linear_head_scales)"""
    res = {}
    biases = biases or {}
    scales = scales or {}
    for key, g in got.items():
        if key not in teacher or not torch.is_tensor(g) or not g.is_floating_point():
            continue
        t = teacher[key]
        if tuple(t.shape) != tuple(g.shape):
            continue
        g, t = _f(g), _f(t).to(g.device)
        fin = torch.isfinite(t)
        if not bool(fin.all()):                   # anchors of padded / out-of-range positions are +-inf by construction (:352-357)
            if not torch.equal(torch.isfinite(g), fin):
                res[key] = dict(rms=float("inf"), max=float("inf"), tol=0.0, tol_max=0.0, R=0, n=g.numel(), kind="finite-pattern")
                continue
            g, t = g[fin], t[fin]
        n = g.numel()
        if BOX_KEYS.search(key):
            # absolute error in sigmoid space, bounded through the box head's delta of the same decoder level
            last = max([int(m.group(1)) for m in (re.match(r"dec(\d+)_delta$", k) for k in teacher) if m] + [0])
            dkey = f"dec{last}_delta" if key == "pred_boxes" else key.replace("_ref", "_delta")
            err = (g - t)
            rms = err.pow(2).mean().sqrt().item()
            mx = err.abs().max().item()
            if dkey in teacher and key != "init_reference":
                d_rms = _f(teacher[dkey]).pow(2).mean().sqrt().item()
                tol = 0.25 * tolerance("dec0_delta", dt) * d_rms
                R = roundings("dec0_delta")
            else:
                tol, R = 1e-6, 0                 # sigmoid of teacher-forced fp32 coordinates: fp32 arithmetic only
            res[key] = dict(rms=rms, max=mx, tol=tol, tol_max=SIGMA_MAX * tol, R=R, n=n, kind="abs")
            continue
        tol = tolerance(key, dt)
        if tol is None:
            continue
        if key in scales:
            rms = ((g - t).pow(2).mean().sqrt() / scales[key]).item()
            mx = ((g - t).abs().max() / scales[key]).item()
        elif BIASED.search(key):
            tc = t - biases.get(key, float(t.mean()))
            rms = ((g - t).pow(2).mean().sqrt() / tc.pow(2).mean().sqrt().clamp_min(1e-30)).item()
            mx = ((g - t).abs().max() / tc.abs().max().clamp_min(1e-30)).item()
        else:
            rms, mx = rel_rms(g, t), rel_max(g, t)
        # the max-norm bound is on err / rms(ref); rel_max divides by max|ref| >= rms(ref), so it is the weaker statement
        res[key] = dict(rms=rms, max=mx, tol=tol, tol_max=SIGMA_MAX * tol, R=roundings(key), n=n, kind="rel")
    return res


def _natural(key):
    return [int(p) if p.isdigit() else p for p in re.split(r"(\d+)", key)]


ORDER = ["vit_embed", "vit_blk", "p2", "p3", "p4", "p5", "p6", "enc_input", "enc", "memory", "output_memory", "enc_class",
         "enc_coord_unact", "query_init", "query_pos", "init_reference", "dec", "pred_logits", "pred_boxes", "mask_features", "mask_embed"]


def _rank(key):
    for i, p in enumerate(ORDER):
        if key == p or (key.startswith(p) and key[len(p):len(p) + 1].isdigit()):
            return (i, _natural(key))
    return (len(ORDER), _natural(key))


def path_roundings(model):
    """roundings in sequence from the image to a stage of the FREE-RUNNING pipeline (the per-stage R of ROUNDINGS summed along the
    path): relative rms error <= GAIN * U_RMS * sqrt(R_path) -- loose (residual streams dilute each block's error) but derived"""
    mv = model.model_vision
    dv = len(mv.backbone.net.blocks)
    ne, nd = mv.transformer.encoder.num_layers, mv.transformer.decoder.num_layers
    p2 = dv * 12 + 12
    memory = p2 + 4 + ne * 15
    return {"p2": p2, "p6": dv * 12 + 7, "memory": memory, "enc_class": memory + 4 + 2, "pred_logits": memory + 4 + nd * 21 + 3,
            "pred_boxes": memory + 4 + nd * 21 + 6}


def path_bound(model, key, dt=torch.bfloat16):
    return GAIN * U_RMS_BY_DTYPE[dt] * math.sqrt(path_roundings(model)[key])


def run(model, image, text, ref_topk, semantic=None, free_run=True, prompt="name", dt=torch.bfloat16):
    """fp32 teacher run, teacher-forced 16-bit run (dt = bfloat16 | float16), (optionally) free-running 16-bit run of one model on one image.
    -> (forced errors {key: ...}, free-running errors {key: ...} or None, outputs dict)"""
    mv = model.model_vision
    mv.set_compute_dtype(torch.float32)
    teacher = StageTap()
    out32 = mv.forward_single(image, text, forced_topk=ref_topk, stages=teacher, semantic=semantic, prompt=prompt)
    mv.set_compute_dtype(dt)
    forced = StageTap(teacher=teacher)
    out_f = mv.forward_single(image, text, forced_topk=ref_topk, stages=forced, semantic=semantic, prompt=prompt)
    biases, scales = head_biases(model), linear_head_scales(model, teacher)
    ferr = stage_errors(forced, teacher, biases, scales, dt)
    if "enc_cls2" in forced and "enc_cls2" in teacher:          # tokens whose main / ambiguous choice a rounding flipped
        flips = int((forced["enc_cls2"].float().argmax(1) != teacher["enc_cls2"].float().argmax(1).to(forced["enc_cls2"].device)).sum())
        ferr["enc_cls2"]["flips"] = flips
    free_err = out_b = free = None
    if free_run:
        free = StageTap()
        out_b = mv.forward_single(image, text, forced_topk=ref_topk, stages=free, semantic=semantic, prompt=prompt)
        free_err = stage_errors(free, teacher, biases, scales, dt)
    return ferr, free_err, dict(fp32=out32, forced=out_f, free=out_b, teacher=teacher, forced_stages=forced,
                                free_stages=free)


def report(tag, ferr, free_err=None, file=None):
    lines = [f"[{tag}] stage: teacher-forced bf16 error (rms / max, relative unless 'abs') | derived tolerance (R roundings)"
             + (" | free-running bf16 error (rms / max)" if free_err else "")]
    for key in sorted(ferr, key=_rank):
        e = ferr[key]
        line = (f"[{tag}] {key:18s} {e['kind']} rms {e['rms']:.2e}  max {e['max']:.2e} | tol rms {e['tol']:.2e} max {e['tol_max']:.2e} "
                f"(R={e['R']:2d})  {'ok' if e['rms'] <= e['tol'] and e['max'] <= e['tol_max'] else 'EXCEEDS'}")
        if free_err and key in free_err:
            line += f" | free rms {free_err[key]['rms']:.2e}  max {free_err[key]['max']:.2e}"
        if "flips" in e:
            line += f" | main/ambiguous head choice flipped on {e['flips']} of {e['n'] // 2} tokens"
        lines.append(line)
    text = "\n".join(lines)
    print(text, file=file, flush=True)
    return text


def violations(ferr):
    return {k: (e["rms"], e["tol"], e["max"], e["tol_max"]) for k, e in ferr.items() if e["rms"] > e["tol"] or e["max"] > e["tol_max"]}
