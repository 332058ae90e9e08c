#!/usr/bin/env python
"""bench.py -- images/sec of the APE-L_D inference forward at 1024^2 (bf16, 80 COCO classes) on N x MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches it under
`python -m torch.distributed.run --nproc-per-node N ...` (one rank per GPU, RCCL) -- and when it is started WITHOUT that
launcher (`python bench.py --gpus 8`, no RANK in the environment) it re-executes itself under it, so N ranks always run and
the line's `n_gpus` / `config.rccl_ranks` are what `torch.distributed` reports.  Rank 0 prints ONE JSON line.

A "step" = `--images-per-step` (default 2) images per rank.  The ViT runs once over the images of a step (every linear at
B x 4096 rows; rows are independent, so each image's result is bit-identical to its own pass), everything behind it is one
batch-1 forward per image, the B of them parallel branches of one hipGraph; steps are software-pipelined (the graph of step i
holds the ViT of step i's images and the tails of step i-1's), and the last step is flushed inside the timed region, so K
timed steps deliver K x B complete images.  One image goes through the whole hot path: normalise + pad + patchify -> EVA-02 ViT-L ->
SimpleFPN -> neck -> 6x (VL fusion + deformable encoder layer) -> two-stage proposal selection -> 6x decoder ->
heads -> class-wise NMS -> masks of the kept detections (upsample, ROIAlign 128, paste) -> detections on the host.
Inputs (images, text-embedding bank) are resident in HBM before the timed region.  Weights are seeded synthetic
(no checkpoint can be downloaded here); data = synthetic COCO-shaped images.

Multi-GPU: images are independent units -> rank r takes a CONTIGUOUS block of the stream (the reference's
InferenceSampler, ape/data/samplers/distributed_sampler_multi_dataset.py:160-170; ape_amd/dp.py shard_indices); the text bank is
broadcast once from rank 0 (RCCL) and each step's fixed-size detection records -- plus, for N > 1, the masks as COCO run
lengths (--mask-format rle, the default for N > 1) -- are all-gathered (RCCL over xGMI), asynchronously: the host awaits step
i's exchange while step i + 1 computes.  scaling = weak (B images per rank per step).

Extra objects on the JSON line: `roofline` for the dominant kernel (the bf16 MFMA GEMM, measured with HIP events on
the launch stream in an instrumented pass right after the timed region), `cpu_baseline` (the oracle -- a CPU port
of the reference algorithm -- one full-depth image timed on the host cores of the same box, rank 0, N = 1 only) and
`parity` (the bf16 pipeline's heads / detections on image 0 against that oracle forward, and COCO box AP of the bf16
detections against the fp32 pipeline's detections as pseudo ground truth over `--ap-images` seeded images).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)       # 100 two-image steps = a timed region of ~2.2 s
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--size", default="L_D_coco", help="L_D_coco = APE-L_D with the COCO config's top-100 (BASELINE config 2); L_D = top-300")
    ap.add_argument("--classes", type=int, default=80)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--stream-images", type=int, default=8, help="distinct synthetic images cycled per rank")
    ap.add_argument("--images-per-step", type=int, default=2,
                    help="images per rank per step, captured as parallel branches of one hipGraph (each one a batch-1 forward)")
    ap.add_argument("--stream", choices=["square", "coco"], default="square",
                    help="square: S x S images (BASELINE config 2); coco: the synthetic COCO-shaped stream of config 4 -- 1000 sizes, "
                         "long side S, short side U[480, S] (seed 5), served by ONE size-agnostic graph (GraphedForward any_size)")
    ap.add_argument("--semantic", action="store_true",
                    help="BASELINE config 5 flavour: semantic branch on (80 thing + 54 stuff classes incl. the leading 'things' class; the "
                         "text bank has 133 rows), captured with the step; label maps (per-pixel argmax) leave the device")
    ap.add_argument("--no-batch-vit", action="store_true", help="one ViT pass per image instead of one per step")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for --dry)")
    ap.add_argument("--dry", action="store_true",
                    help="launch + exchange check without a GPU: the ranks start, rendezvous (use --backend gloo), broadcast a text "
                         "bank and all-gather K steps of synthetic detection records on the CPU; no forward runs and `value` is null")
    ap.add_argument("--cpu-images", type=int, default=3, help="images the CPU oracle times after its warm-up pass (cpu_baseline)")
    ap.add_argument("--ap-images", type=int, default=16,
                    help="seeded images for the box-AP parity number (bf16 detections vs the fp32 pipeline's as pseudo ground truth)")
    ap.add_argument("--input", choices=["resident", "uint8"], default="resident",
                    help="resident: float32 model-ready images already in HBM (the tier's definition of `value`); uint8: the predictor's "
                         "input side inside the timed region (ape/engine/defaults.py:213-222) -- ORIGINAL uint8 BGR images (1.25 x the model "
                         "size) in pinned host memory -> upload -> ResizeShortestEdge (Pillow-exact resize kernel) + BGR->RGB + float CHW -> forward")
    ap.add_argument("--mask-format", choices=["auto", "bitmask", "rle", "both"], default="auto",
                    help="how masks leave the device: bitmask = [k, H, W] bool on the host (the reference's Instances contract, 105 MB per "
                         "image at k = 100); rle = the evaluators' COCO run lengths only, encoded on the device (a few KB per image); both = the "
                         "bitmasks to this rank's host AND the run lengths, which are all-gathered across ranks with the records.  auto = "
                         "bitmask on 1 GPU, both on N > 1: EVERY rank of an N-GPU run does exactly the work of the 1-GPU run (same masks to its "
                         "host) plus the exchange (north_star: all-gather of boxes / masks), so the driver's 1 -> N efficiency compares like with like")
    ap.add_argument("--n1-value", type=float, default=None,
                    help="N > 1: the images/sec of the N = 1 run of the same workload (e.g. the driver's previous line): efficiency_vs_n1 = "
                         "value / (N x this).  Without it rank 0 measures the reference itself, in the same process, right after the timed "
                         "region (--solo-steps steps of the same step with no exchange while the other ranks wait at a barrier)")
    ap.add_argument("--solo-steps", type=int, default=20, help="N > 1 without --n1-value: steps of the same-process N = 1 reference on rank 0")
    ap.add_argument("--no-second-flavour", action="store_true",
                    help="skip the second timed region of the default bf16 run: the SAME step in IEEE half (--dtype f16's pipeline, the "
                         "reference's own evaluation dtype), --f16-steps steps, reported as value_f16 / ms_per_step_f16 (N = 1 only)")
    ap.add_argument("--f16-steps", type=int, default=0, help="0 = the same number of steps (and the same warm-up) as the bf16 region")
    ap.add_argument("--dry-images", type=int, default=1000, help="--dry: length of the sharded synthetic stream")
    ap.add_argument("--instrumented-only", action="store_true",
                    help="skip the timed region: run only the instrumented pass that produces `roofline` (what tools/gpu_profile.sh puts "
                         "under rocprofv3 --kernel-trace --stats, so that the trace holds exactly the launches the HIP events metered)")
    ap.add_argument("--dtype", choices=["bf16", "f16"], default="bf16",
                    help="16-bit flavour of the timed pipeline: bf16 = BASELINE's dtype (default); f16 = IEEE half operands, the reference's "
                         "own evaluation dtype (tools/train_net.py:642) -- same kernels, v_mfma_f32_16x16x32_f16, 3 more mantissa bits")
    ap.add_argument("--ablate", default="", metavar="KEY=INT[,KEY=INT]",
                    help="EXPERIMENT ONLY (the line is marked invalid): override integer fields of the model configuration, e.g. dec_layers=1 or "
                         "depth=12 -- the marginal cost of a stage inside the pipelined step = the step time with and without it")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="no software pipeline over steps (ViT of step i+1 || tails of step i inside one graph)")
    return ap.parse_args()


def make_images(n, S, seed, device):
    g = torch.Generator().manual_seed(seed)
    return [torch.randint(0, 256, (3, S, S), generator=g).float().to(device) for _ in range(n)]


def make_raw_images(n, S, seed):
    """ORIGINAL images as the predictor receives them (cv2.imread: uint8 [H, W, 3] BGR) in pinned host memory, 1.25 x the model size
    so that ResizeShortestEdge(S, S) really resamples (Pillow's triangle filter over a 2.5-pixel support)"""
    g = torch.Generator().manual_seed(seed)
    H = S * 5 // 4
    return [torch.randint(0, 256, (H, H, 3), generator=g, dtype=torch.uint8).pin_memory() for _ in range(n)]


class GemmMeter:
    """wraps ape_amd.ops.gemm / ffn_fused / conv3x3 and meters every launch they make TWICE:
    * exactly -- the library's launch meter (csrc/meter.cpp, ape_hip_meter_*): while it is on, every kernel of the library goes out
      through hipExtLaunchKernelGGL with its own (start, stop) HIP events, whose elapsed time is the dispatch's begin-to-end duration
      (the number rocprofv3's kernel trace reports for the same launch).  This is what `roofline.achieved` is computed from;
    * with a HIP event pair RECORDED around the call (torch.cuda.Event on the launch stream = torch's current stream), the
      round-1..4 method: it additionally contains the command-processor gaps either side of the kernel (~4-5 us per launch) and is kept
      on the line as `avg_launch_us_event_pair` for continuity.
    The library names the kernel symbol each call launched (ape_hip_gemm_last_kernel)."""

    def __init__(self, ops):
        import ctypes
        from ape_amd import _lib
        self.ops, self.orig, self.orig_ffn, self.orig_conv, self.records = ops, ops.gemm, ops.ffn_fused, ops.conv3x3, []
        self.lib = _lib.load()
        self._ct = ctypes

    def _metered(self, call, name_of, flops, shape):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n0 = self.lib.ape_hip_meter_count()
        s.record()
        out = call()
        e.record()
        self.records.append((name_of(), s, e, flops, shape, n0, self.lib.ape_hip_meter_count()))
        return out

    def __enter__(self):
        last = lambda: self.lib.ape_hip_gemm_last_kernel().decode()

        def wrapped(a, w, *args, **kw):
            return self._metered(lambda: self.orig(a, w, *args, **kw), last, 2.0 * a.shape[0] * a.shape[1] * w.shape[0],
                                 f"{a.shape[0]}x{w.shape[0]}x{a.shape[1]}")

        def wrapped_ffn(x, w1, b1, w2, b2, **kw):
            # the one-kernel FFN (csrc/ffn_fused.hip): two chained contractions, 2 * M * 256 * HID flops each
            return self._metered(lambda: self.orig_ffn(x, w1, b1, w2, b2, **kw),
                                 lambda: "ffn_fused_kernel<true>" if kw.get("w2_permuted") else "ffn_fused_kernel<false>",
                                 2.0 * x.shape[0] * x.shape[1] * w1.shape[0] + 2.0 * x.shape[0] * w1.shape[0] * w2.shape[0],
                                 f"{x.shape[0]}x{x.shape[1]}x{w1.shape[0]}x{w2.shape[0]}")

        def wrapped_conv(x, perm, h, wd, w, bias=None, **kw):
            # the 3 x 3 convolutions as implicit GEMMs (the tile kernel stages A from the shifted input rows): the same 2 M N K flops as
            # the im2col GEMM they replace, M = h * wd, K = 9 C.  When the im2col path is taken instead, its inner ops.gemm is metered.
            if not self.ops.conv3x3_implicit_ok(x, w, h, wd):
                return self.orig_conv(x, perm, h, wd, w, bias, **kw)
            return self._metered(lambda: self.orig_conv(x, perm, h, wd, w, bias, **kw), last, 2.0 * h * wd * w.shape[0] * w.shape[1],
                                 f"{h * wd}x{w.shape[0]}x{w.shape[1]} (implicit 3x3 conv)")
        self.lib.ape_hip_meter_begin()
        self.ops.gemm, self.ops.ffn_fused, self.ops.conv3x3 = wrapped, wrapped_ffn, wrapped_conv
        return self

    def __exit__(self, *exc):
        self.ops.gemm, self.ops.ffn_fused, self.ops.conv3x3 = self.orig, self.orig_ffn, self.orig_conv
        self.lib.ape_hip_meter_end()

    @staticmethod
    def family(name):
        """the eight-wave tile kernel is compiled once per epilogue (third template argument, csrc/gemm_p8.hip) and once more as the
        implicit 3 x 3 convolution (`..., conv3x3>`: same tile, schedule and MFMA stream, the A rows staged from shifted input rows): the
        roofline object is about the tile kernel as a whole = the sum over these instantiations (rocprofv3 lists them as separate rows)"""
        import re
        # round 6: `..., rowstat>` (the 256 x 128 tile kernel computing the folded LayerNorm's row statistics in its main loop) folds into
        # its tile family like the other instantiations do -- which makes the two families nearly equal in total time;
        # `roofline.tile_kernel_families` carries both, whichever is larger
        return re.sub(r"^(gemm_(?:bf16|f16)_p8_kernel<\d+, \w+), (?:\d+|conv3x3|persistent|rowstat)>$", r"\1>", name)

    def launch(self, i):
        """(kernel expression at the launch site, milliseconds) of launch i of the metering session"""
        name, ms = self._ct.c_char_p(), self._ct.c_float()
        if self.lib.ape_hip_meter_read(i, self._ct.byref(name), self._ct.byref(ms)) != 0:
            raise RuntimeError(self.lib.ape_hip_last_error().decode())
        return name.value.decode().strip("()"), float(ms.value)

    def summary(self):
        """per kernel family: [launches, exact seconds, flops, event-pair seconds], sorted by exact time; per exact symbol and per
        (family, shape) alike; `self.kernels` = every library kernel of the session by launch-site expression: [launches, seconds]"""
        torch.cuda.synchronize()
        groups, symbols, shapes = {}, {}, {}
        for name, s, e, fl, shape, n0, n1 in self.records:
            pair = s.elapsed_time(e) * 1e-3
            dt = sum(self.launch(i)[1] for i in range(n0, n1)) * 1e-3 if n1 > n0 else pair
            for table, key in ((groups, self.family(name)), (symbols, name), (shapes, (self.family(name), shape))):
                g = table.setdefault(key, [0, 0.0, 0.0, 0.0])
                g[0] += 1
                g[1] += dt
                g[2] += fl
                g[3] += pair
        self.symbols, self.shapes = symbols, shapes
        self.kernels = {}
        for i in range(self.lib.ape_hip_meter_count()):
            name, ms = self.launch(i)
            k = self.kernels.setdefault(name, [0, 0.0])
            k[0] += 1
            k[1] += ms * 1e-3
        return sorted(groups.items(), key=lambda kv: -kv[1][1])


def pmc_traffic_bytes(kernel):
    """(bytes per launch | None, note): memory-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (tools/gpu_pmc.sh
    -> profiles/r0N_pmc_summary.txt: separate FETCH_SIZE and WRITE_SIZE passes, FETCH_SIZE doubled as MI355X_MICROARCH.md's HBM section
    prescribes for gfx950).  Only a summary measured on THIS build counts: its `# library_digest` header (tools/pmc_summary.py) must
    equal ape_amd.build._digest() of the sources bench.py runs on -- a summary of an older build describes another launch mix and is
    refused (traffic = null, the note says why).  The library reports `gemm_bf16_p8_kernel<256, true>` / `gemm_f16_...`; rocprofv3 lists
    the full instantiations (`gemm_bf16_p8_kernel<256, true, unsigned short, 0, false>`, `..., 0, true>` = the implicit convolution):
    matched by prefix + operand type, launch-weighted over the family's rows."""
    from ape_amd import build as _build
    f16 = kernel.startswith("gemm_f16_")
    prefix = ("gemm_bf16_" + kernel[len("gemm_f16_"):] if f16 else kernel).rstrip(">")
    names = sorted((n for n in os.listdir(os.path.join(ROOT, "profiles")) if n.endswith("_pmc_summary.txt")), reverse=True)
    want = _build._digest()
    for name in names:
        lines = open(os.path.join(ROOT, "profiles", name)).read().splitlines()
        digest = next((l.split()[-1] for l in lines if l.startswith("# library_digest")), None)
        if digest != want:
            return None, (f"profiles/{name} (the newest PMC summary) was measured on library digest {str(digest)[:12]}, this build is "
                          f"{want[:12]}: refused -- re-run tools/gpu_pmc.sh on this build")
        tot_b, tot_n = 0.0, 0
        for line in lines:
            cols = [c.strip() for c in line.split("|")]
            if len(cols) <= 3 or not cols[0].startswith(prefix):
                continue
            rest = cols[0][len(prefix):]
            if rest not in (">", "") and not rest.startswith(","):
                continue                                   # a longer first-arguments match (e.g. <256, true> vs <256, true1>)
            if ("_Float16" in rest) != f16 and rest not in (">", ""):
                continue
            try:
                n = int(cols[1])
                tot_b += float(cols[-1]) * 1024.0 * 1024.0 * n
                tot_n += n
            except ValueError:
                continue
        if tot_n:
            return tot_b / tot_n, f"profiles/{name}, {tot_n} launches, same library digest as this build"
        return None, f"profiles/{name} does not list {kernel}"
    return None, "no PMC summary under profiles/"


def cpu_baseline(model, size, images, text, n_images=3, max_threads=64):
    """The oracle (CPU port of the reference algorithm, oracle/ape_oracle.py) on the host cores of this box: one warm-up pass
    through a depth-reduced copy (thread pool / primitive caches), then `n_images` full-depth fp32 forwards, each timed.  The
    reference itself cannot run here: /root/reference does not exist on the GPU box (its timing on the build container is in
    BASELINE.md).  Returns (cpu_baseline object, [oracle stage tensors per timed image]) -- the latter feed the `parity` object."""
    from oracle import ape_oracle
    from oracle.configs import CONFIGS

    cfg = dict(CONFIGS[size])
    depth, enc, dec = cfg["depth"], cfg["enc_layers"], cfg["dec_layers"]
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    cores = min(os.cpu_count() or 1, max_threads)
    torch.set_num_threads(cores)
    txt = text.cpu()
    warm = dict(cfg, depth=3, enc_layers=1, dec_layers=1)
    sdw = dict(sd)
    for k in list(sd):       # the reduced oracle reads decoder level 0 heads and the encoder-side heads stored at index `dec`
        for sub in ("class_embed.", "bbox_embed."):
            if f"{sub}{dec}." in k:
                sdw[k.replace(f"{sub}{dec}.", f"{sub}1.")] = sd[k]
    ape_oracle.ApeOracle(warm, sdw).forward(images[0].cpu(), txt)
    times, stages, per_stage = [], [], []
    for img in images[:max(1, n_images)]:
        orc = ape_oracle.ApeOracle(cfg, sd)
        t0 = time.perf_counter()
        orc.forward(img.cpu(), txt)
        times.append(time.perf_counter() - t0)
        T = orc.timers
        vit = T.get("vit_win_block", 0) + T.get("vit_glb_block", 0)
        per_stage.append({"vit": vit, "encoder": T.get("enc_layer", 0), "decoder": T.get("dec_layer", 0),
                          "rest (pyramid, heads, selection, masks)": times[-1] - vit - T.get("enc_layer", 0) - T.get("dec_layer", 0)})
        stages.append(orc.stages)
    mean = sum(times) / len(times)
    stage_mean = {k: round(sum(p[k] for p in per_stage) / len(per_stage), 2) for k in per_stage[0]}
    return ({"value": 1.0 / mean, "unit": "images/sec", "cores": cores,
             "kind": "port", "kind_note": "port = oracle/ape_oracle.py, the CPU restatement pinned to the reference's fixtures; the "
                                          "reference itself is absent on the GPU box",
             "seconds_per_image": [round(t, 2) for t in times], "seconds_per_stage": stage_mean,
             "sample": (f"{len(times)} images {tuple(images[0].shape)} timed one by one after a depth-reduced warm-up pass, full depth "
                        f"({depth} ViT blocks, {enc}+{dec} layers), oracle fp32: mean {mean:.1f} s per image; torch {torch.__version__} CPU, "
                        f"{cores} threads")}, stages)


def _detections_of(out):
    keep = out["det_scores"] >= 0
    return out["det_boxes"][keep].float().cpu(), out["det_scores"][keep].float().cpu(), out["det_classes"][keep].cpu()


DTYPES = {"bf16": torch.bfloat16, "f16": torch.float16}


def box_ap_vs_fp32(mv, images, text, flavours=("bf16", "f16")):
    """COCO box AP (IoU 0.50:0.95) of each 16-bit pipeline's detections (bf16 = the timed default; f16 = the reference's eval dtype) against the fp32-kernel pipeline's detections of the same
    images as pseudo ground truth (the fp32 pipeline is pinned to the reference's fixtures to <= 1e-3: tests/test_model_gpu.py
    ::test_L_D_fp32_matches_reference).  Both runs select their own proposals (nothing is teacher forced)."""
    from ape_amd.evaluation import box_ap

    gts, ctl, spread = [], [], []
    dets, same_set = {f: [] for f in flavours}, {f: [] for f in flavours}
    gen = torch.Generator().manual_seed(7001)
    for img in images:
        mv.set_compute_dtype(torch.float32)
        ref = mv.forward_single(img, text, with_masks=False)
        rb, rs, rc = _detections_of(ref)
        # control: the SAME fp32 pipeline on the image + uniform noise of half an 8-bit grey level (below the quantisation of any
        # decoded JPEG): how much of the AP deficit is the metric's sensitivity on this seeded random-weight model
        noisy = img + (torch.rand(img.shape, generator=gen) - 0.5).to(img.device)
        ctl.append(_detections_of(mv.forward_single(noisy, text, with_masks=False)))
        spread.append(float(rs.max() - rs.min()) if rs.numel() else 0.0)
        gts.append((rb, rc))
        a = set(zip(ref["det_query"].tolist(), ref["det_classes"].tolist()))
        for f in flavours:
            mv.set_compute_dtype(DTYPES[f])
            got = mv.forward_single(img, text, with_masks=False)
            dets[f].append(_detections_of(got))
            b = set(zip(got["det_query"].tolist(), got["det_classes"].tolist()))
            same_set[f].append(len(a & b) / max(len(a), 1))
    r = {}
    for f in flavours:
        r[f] = box_ap(dets[f], gts)
        r[f]["same_query_class_pairs"] = sum(same_set[f]) / max(len(same_set[f]), 1)
    r["images"] = len(images)
    c = box_ap(ctl, gts)
    r["control_fp32_on_half_grey_level_input_noise"] = {"AP": c["AP"], "AP50": c["AP50"], "AP75": c["AP75"]}
    r["fp32_detection_score_range"] = sum(spread) / max(len(spread), 1)
    r["reading"] = ("pseudo ground truth = the fp32 pipeline's own top-k detections; with seeded random weights the kept scores span the "
                    "range above, so WHICH (query, class) pairs make the top-k is decided by differences far below bf16's resolution: the "
                    "control (same fp32 arithmetic, input noise below 8-bit quantisation) bounds what this metric can show on this model")
    return r


def same_rounding_parity(model, size, image, text, flavour):
    """T2 of BASELINE.md section 3 for the timed flavour on ONE timed image: the oracle evaluated at the pipeline's own 16-bit rounding
    points (oracle/rounded.py, ~1 min of host time at 1024^2) is the teacher of a teacher-forced run of the HIP pipeline; every stage's
    output against the rounded oracle's -- what remains is accumulation order and the rounding flips it causes (tests/test_same_rounding.py
    asserts <= 2e-3 per stage at the BASELINE configurations)."""
    from ape_amd.stagetap import StageTap
    from oracle import rounded
    from oracle.configs import CONFIGS

    mv = model.model_vision
    dt = DTYPES[flavour]
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    orc = rounded.RoundedApeOracle(dict(CONFIGS[size]), sd, dtype=dt)
    t0 = time.perf_counter()
    orc.forward(image.cpu(), text.cpu())
    sec = time.perf_counter() - t0
    net = mv.backbone.net
    teacher = rounded.teacher_stages(orc, net.token_order(net.img_size // net.patch_size)[0])
    mv.set_compute_dtype(dt)
    forced = StageTap(teacher=teacher)
    mv.forward_single(image, text, forced_topk=orc.hip["topk_proposals"].to(image.device), stages=forced)
    dist = rounded.stage_distances(forced, teacher)
    worst = sorted(dist, key=lambda k: -dist[k][0])
    return {"flavour": flavour, "stages": len(dist), "max_stage_rel_rms": dist[worst[0]][0] if worst else None,
            "worst_stages": {k: round(dist[k][0], 6) for k in worst[:6]},
            "pred_logits_rel_rms": dist.get("pred_logits", (None,))[0], "pred_boxes_rel_rms": dist.get("pred_boxes", (None,))[0],
            "stages_over_2e-3": sorted(k for k in dist if dist[k][0] > 2e-3), "oracle_seconds": round(sec, 1),
            "method": "every stage of the 16-bit HIP pipeline teacher-forced with the input of the oracle evaluated at the SAME rounding "
                      "points (storage dtype of every tensor, 16-bit GEMM operands, half offsets / values, the flash loop's per-tile "
                      "probability rounding); relative rms of the stage outputs"}


def parity_object(mv, images, text, stages_per_image, ap_images, timed="bf16"):
    """both 16-bit flavours of the HIP pipeline (bf16 and f16; `timed` names the one the line's `value` was measured on) against the
    oracle's fp32 forwards of the same images (same weights): head tensors with the oracle's proposal order injected (max-abs
    error / max-abs reference, and rms), detection-level agreement of the free run, box AP against the oracle's detections (the
    timed images) and against the fp32 pipeline's (ap_images seeded images)"""
    res = {"timed_flavour": timed}
    for f in ("bf16", "f16"):
        mv.set_compute_dtype(DTYPES[f])
        res[f] = _parity_one(mv, images, text, stages_per_image)
    for k in ("pred_logits_relerr", "pred_boxes_relerr", "proposal_overlap", "detections_matched"):     # the timed flavour's, at top level
        res[k] = res[timed][k]
    res["against"] = res[timed].pop("against")
    res["f16"].pop("against", None), res["bf16"].pop("against", None)
    res["match_rule"] = "same class, |score| < 0.05, box within 5 %"
    res["note"] = ("16-bit storage / fp32 accumulate vs fp32; f16 = the reference's own evaluation dtype (tools/train_net.py:642), bf16 = "
                   "BASELINE's; max-norm head errors of a random-weight model are single-query outliers of the decoder's refinement "
                   "(profiles/r03_bf16_error_trace.log); the fp32 kernels meet 1e-3 (tests/test_model_gpu.py)")
    if ap_images > 0:
        S = images[0].shape[-1]
        res["box_ap_vs_fp32_pipeline"] = box_ap_vs_fp32(mv, make_images(ap_images, S, seed=7000, device=images[0].device), text)
    mv.set_compute_dtype(DTYPES[timed])
    return res


def _parity_one(mv, images, text, stages_per_image):
    from ape_amd.evaluation import box_ap

    def rel(a, b):
        a, b = a.float().cpu(), b.float().cpu()
        return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))

    def rms(a, b):
        a, b = a.float().cpu(), b.float().cpu()
        return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-12))

    per_image, dets, gts = [], [], []
    for image, O in zip(images, stages_per_image):
        st = {}
        mv.forward_single(image, text, forced_topk=O["topk_proposals"][0].to(image.device), stages=st)
        out = mv.forward_single(image, text)
        ob, os_, oc = O["det_boxes"], O["det_scores"], O["det_classes"]
        gb, gs, gc = out["det_boxes"].cpu(), out["det_scores"].cpu(), out["det_classes"].cpu()
        used, matched = set(), 0
        for j in range(len(os_)):
            cand = ((gc == oc[j]) & ((gs - os_[j]).abs() < 0.05)).nonzero().flatten().tolist()
            for i in cand:
                if i not in used and float((gb[i] - ob[j]).abs().max()) < 0.05 * max(1.0, float(ob[j].abs().max())):
                    used.add(i)
                    matched += 1
                    break
        per_image.append({"pred_logits_relerr": rel(st["pred_logits"], O["pred_logits"][0]), "pred_boxes_relerr": rel(st["pred_boxes"], O["pred_boxes"][0]),
                          "pred_logits_rms": rms(st["pred_logits"], O["pred_logits"][0]), "pred_boxes_rms": rms(st["pred_boxes"], O["pred_boxes"][0]),
                          "proposal_overlap": len(set(out["topk_proposals"].cpu().tolist()) & set(O["topk_proposals"][0].tolist())) / float(O["topk_proposals"].shape[-1]),
                          "detections_matched": matched / max(len(os_), 1)})
        dets.append(_detections_of(out))
        gts.append((ob.float(), oc))
    first = per_image[0]
    res = {"against": f"oracle fp32 (CPU) on {len(per_image)} image(s), same seeded weights; per-stage teacher-forced bounds: tests/test_teacher_forced.py",
           "pred_logits_relerr": first["pred_logits_relerr"], "pred_boxes_relerr": first["pred_boxes_relerr"],
           "proposal_overlap": first["proposal_overlap"], "detections_matched": first["detections_matched"],
           "per_image": per_image, "box_ap_vs_oracle": box_ap(dets, gts)}
    return res


def workload_string(size, B, classes, semantic=False):
    """the workload a rank runs per step, as the line's config names it -- a function of the model configuration and the step size ONLY
    (no rank count in it: the driver compares the N = 1 and N = 8 lines of the SAME workload; tests/test_bench_launch.py)"""
    from ape_amd.modeling.build import SIZES
    S, topk = SIZES[size]["img_size"], SIZES[size]["topk_eval"]
    return (f"APE-L_D forward (size key {size}), {B}x{S}x{S} images per rank per step: one ViT pass over the {B} images, everything after "
            f"it one batch-1 forward per image ({B} parallel branches of one hipGraph); {classes} classes (name prompt), masks on, "
            f"top-{topk} detections per image incl. their full-resolution masks on the host; seeded synthetic weights"
            + ("; semantic branch on (54 stuff columns), label maps on the host" if semantic else ""))


def pin_to_gpu_numa_node(local):
    """bind this rank's host threads to the CPUs of its GPU's NUMA node (the pinned staging buffers and the launch thread then sit next
    to the device's PCIe root: 8 ranks launching ~1400 kernels per step from remote sockets is the classic reason a 1 -> 8 curve bends).
    Reads the device's PCI address from torch and its `local_cpulist` from sysfs; returns what it did (goes on the JSON line)."""
    try:
        p = torch.cuda.get_device_properties(local)
        bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        base = f"/sys/bus/pci/devices/{bdf}"
        node = int(open(f"{base}/numa_node").read().strip())
        cpus = set()
        for part in open(f"{base}/local_cpulist").read().strip().split(","):
            if part:
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
        allowed = os.sched_getaffinity(0)
        cpus = cpus & allowed
        if node < 0 or not cpus or cpus == allowed:
            return {"pci": bdf, "numa_node": node, "pinned": False, "cpus": len(allowed)}
        os.sched_setaffinity(0, cpus)
        return {"pci": bdf, "numa_node": node, "pinned": True, "cpus": len(cpus)}
    except Exception as exc:           # no sysfs / no permission: run unpinned, say so
        return {"pinned": False, "error": f"{type(exc).__name__}: {exc}"[:120]}


def shards_digest(n_items, world):
    """contiguous InferenceSampler shards of an n_items stream over `world` ranks: [(first, last, count)] per rank"""
    from ape_amd.dp import shard_indices
    out = []
    for r in range(world):
        idx = shard_indices(n_items, r, world)
        out.append([idx[0], idx[-1], len(idx)])
    return out


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher in the environment: run N ranks under torch.distributed.run on this
    node (127.0.0.1 rendezvous) and pass their output through.  The ranks see RANK / WORLD_SIZE and take the normal path."""
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_run(args, rank, world):
    """the distributed plumbing without a device: rendezvous, text-bank broadcast from rank 0, K all-gathers of [B, k, 6] records"""
    import torch.distributed as dist
    from ape_amd.dp import DataParallelRunner, shard_indices

    dev = torch.device("cpu")
    k = 100

    class Fake:
        def submit(self, image, text, height=None, width=None, prompt="name"):
            from types import SimpleNamespace
            return SimpleNamespace(rec6=torch.full((args.images_per_step, k, 6), float(rank)), runs=None)

        def result(self, ticket):
            return None, ticket.rec6

    dp = DataParallelRunner(Fake(), k, dev, lag=1)
    bank = torch.randn(args.classes, 1024, generator=torch.Generator().manual_seed(3)) if rank == 0 else None
    text = dp.broadcast_text_bank(bank, args.classes, 1024)
    mine = shard_indices(args.dry_images, rank, world)
    steps = args.steps if args.stream != "coco" else (len(mine) + args.images_per_step - 1) // args.images_per_step
    dist.barrier()
    t0 = time.perf_counter()
    gathered, collected = None, 0
    for i in range(steps):
        g = dp.result(dp.submit(None, text))[1]                 # lag 1: the records of the PREVIOUS step (None at the first)
        collected += g is not None
        gathered = g if g is not None else gathered
    g = dp.drain()                                              # the last step's records
    collected += g is not None
    gathered = g if g is not None else gathered                 # [world, B, k, 6]: every rank's records of one step
    dist.barrier()
    elapsed = time.perf_counter() - t0
    sums = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(sums, text.double().sum().reshape(1))
    if rank == 0:
        print(json.dumps({"metric": "dry run: launch + exchanges only", "value": None, "unit": "images/sec", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(args.steps, 1),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "dry-run (no forward)",
                          "config": {"workload": workload_string(args.size, args.images_per_step, args.classes, args.semantic),
                                     "dry": True, "shards": shards_digest(args.dry_images, world),
                                     "parallelism": f"dp{world}", "rccl_ranks": dist.get_world_size(),
                                     "backend": dist.get_backend(), "shard_of_rank0": [mine[0], mine[-1]], "steps_run": steps,
                                     "record_sets_collected": collected, "shard_sizes": [len(shard_indices(args.dry_images, r, world)) for r in range(world)],
                                     "text_bank_identical_on_all_ranks": bool(all(abs(float(x) - float(sums[0])) < 1e-9 for x in sums)),
                                     "gathered_shape": list(gathered.shape) if torch.is_tensor(gathered) else None,
                                     "gathered_rank_ids": sorted({int(v) for v in gathered[:, 0, 0, 0].tolist()}) if torch.is_tensor(gathered) else None}}),
              flush=True)


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        # started without a launcher: become the launcher (N ranks under torch.distributed.run), never a silent 1-GPU run
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry:
        import torch.distributed as dist
        if "RANK" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"), RANK="0", WORLD_SIZE="1")
        dist.init_process_group(args.backend)
        dry_run(args, rank, world)
        dist.barrier()
        dist.destroy_process_group()
        return None
    if os.environ.get("APE_BENCH_SHARE_GPU") == "1":
        # smoke of the N > 1 code path on a box with fewer GPUs than ranks (ranks share a device; use --backend gloo: RCCL refuses two
        # ranks on one GPU).  Throughput of such a run means nothing; the exchange, the lagged collection and the timing protocol run
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    affinity = pin_to_gpu_numa_node(local) if world > 1 or os.environ.get("APE_BENCH_PIN") == "1" else {"pinned": False, "note": "1 rank: not pinned"}
    dist = None
    if world > 1 or "RANK" in os.environ:   # under torch.distributed.run the RCCL path is exercised even for N = 1
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(args.backend, device_id=dev if args.backend == "nccl" else None)

    import ape_amd.ops as ops
    from ape_amd.modeling.build import build_ape, init_synthetic
    from ape_amd.runtime import GraphedForward

    ablate = {k: int(v) for k, v in (kv.split("=") for kv in args.ablate.split(",") if kv)}
    model = init_synthetic(build_ape(args.size, **ablate), seed=0).to(dev)
    mv = model.model_vision
    mv.set_compute_dtype(DTYPES[args.dtype])
    S = mv.backbone.padding_constraints["square_size"]
    from ape_amd.dp import DataParallelRunner

    B = args.images_per_step
    sem_meta = None
    if args.semantic:
        things, stuff = [f"thing{i}" for i in range(80)], ["things"] + [f"stuff{i}" for i in range(53)]
        mv.semantic_on = True
        mv.set_metadata(0, name="bench_thing_stuff", thing_classes=things, stuff_classes=stuff)
        sem_meta = dict(mv.metadata_list[-1], entity=mv.dataset_entities[-1])
        args.classes = len(things) + len(stuff) - 1
    # one contract for every N: each rank delivers its images' bitmasks to its own host; N > 1 ADDS the run-length encoding and the
    # all-gather of records + runs (a rank of an N-GPU run never does less than the 1-GPU run)
    mask_format = args.mask_format if args.mask_format != "auto" else ("both" if world > 1 else "bitmask")
    graphed = GraphedForward(mv, use_graph=not args.no_graph, images_per_step=B, batch_vit=not args.no_batch_vit,
                             pipeline=not args.no_pipeline, any_size=args.stream == "coco", semantic=sem_meta, mask_format=mask_format,
                             input_resize=(S, S), input_format="RGB")
    # N > 1: masks travel as COCO run lengths and are all-gathered with the records (north_star: "all-gather of boxes / masks over
    # xGMI"); the exchange of step i is awaited one step later and only by the host (lag = 1): a slow rank never stalls another GPU
    dp = DataParallelRunner(graphed, mv.test_topk_per_image, dev, gather_masks=mask_format in ("rle", "both") and world > 1, lag=1 if world > 1 else 0)
    # text-embedding bank: produced once on rank 0 (the CLIP text tower's output contract [K,1024]) and broadcast (RCCL)
    bank = torch.randn(args.classes, 1024, generator=torch.Generator().manual_seed(3)) if rank == 0 else None
    text = dp.broadcast_text_bank(bank, args.classes, 1024)
    # rank r owns its own block of the synthetic stream (contiguous shards like the reference's InferenceSampler: dp.shard_indices)
    from ape_amd.dp import shard_indices
    images = make_images(args.stream_images, S, seed=100 + rank, device=dev)
    raw_images = make_raw_images(args.stream_images, S, seed=100 + rank) if args.input == "uint8" else None
    if raw_images is not None and args.stream == "coco":
        raise SystemExit("--input uint8 is implemented for --stream square")
    if args.stream == "coco":
        # 1000 sizes, long side S, short side U[480, S] rounded, either orientation (seed 5); image i of the
        # stream = the top-left (h_i, w_i) crop of a base image (views: no extra memory); rank r takes its contiguous block
        g5 = torch.Generator().manual_seed(5)
        short = torch.randint(480, S + 1, (1000,), generator=g5).tolist()
        tall = (torch.rand(1000, generator=g5) < 0.3).tolist()
        sizes = [(S, sh) if t else (sh, S) for sh, t in zip(short, tall)]
        mine = [sizes[i] for i in shard_indices(len(sizes), rank, world)]
        base = images
        images = [base[i % len(base)][:, :h, :w] for i, (h, w) in enumerate(mine)]

    # One step = one image per rank, submitted to the double-buffered runtime: the forward of image i is enqueued (hipGraph
    # replay + mask paste + record all-gather), then the host collects image i-1, whose device->host transfer ran on the
    # copy stream meanwhile.  K timed steps = K submits + K collected results (the last one is flushed inside the timed
    # region), so `value` counts K complete images per rank, masks on the host included.
    # the host collects a ticket `depth` steps after submitting it: 1 = while the next image computes (its transfer ran on the
    # copy stream meanwhile); 2 with the software pipeline, where a ticket's detections are produced by the next step's replay
    depth = 1 if args.no_pipeline else 2

    def timed_region(runner, warmup, steps, collective=True, trace=None):
        """`warmup` untimed steps, then EXACTLY `steps` steps (the last one flushed inside the region) between barrier +
        synchronize on both sides; returns this rank's seconds.  runner: the DataParallelRunner (records / runs exchanged), or the
        GraphedForward itself (the same step with no exchange: the same-process N = 1 reference of an N-rank run)."""
        queue = []
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(warmup + steps + 2)] if trace is not None else None

        def step(i):
            if raw_images is not None:
                # the masks are delivered in the S x S frame of the resident-input workload (the predictor's default is the ORIGINAL
                # size): the two numbers then differ by the input side only
                batch = [raw_images[(i * B + b) % len(raw_images)] for b in range(B)]
                queue.append(runner.submit(batch if B > 1 else batch[0], text, height=S, width=S))
            else:
                batch = [images[(i * B + b) % len(images)] for b in range(B)]
                queue.append(runner.submit(batch if B > 1 else batch[0], text))
            return runner.result(queue.pop(0)) if len(queue) > depth else None

        def flush():
            while queue:
                runner.result(queue.pop(0))
            if hasattr(runner, "drain"):
                runner.drain()

        # `trace` (a dict): stream-side milliseconds of every step, warm-up included -- an event recorded on the compute stream behind each
        # submit (no host wait: the region is timed exactly as without it); what the driver's short command pays per step, in order
        if marks is not None:
            marks[0].record()
        for i in range(warmup):
            step(i)
            if marks is not None:
                marks[1 + i].record()
        flush()
        torch.cuda.synchronize()
        if dist is not None and collective:
            dist.barrier()
        t0 = time.perf_counter()
        if marks is not None:
            marks[warmup + 1].record()
        for i in range(steps):
            step(warmup + i)
            if marks is not None:
                marks[warmup + 2 + i].record()
        flush()
        torch.cuda.synchronize()
        if dist is not None and collective:
            dist.barrier()
        sec = max(time.perf_counter() - t0, 1e-9)
        if marks is not None:
            trace["warmup_ms"] = [round(marks[i].elapsed_time(marks[i + 1]), 3) for i in range(warmup)]
            trace["timed_ms"] = [round(marks[warmup + 1 + i].elapsed_time(marks[warmup + 2 + i]), 3) for i in range(steps)]
            trace["note"] = ("stream-side ms between the events recorded behind consecutive submits (the first timed entry includes the ViT-only "
                             "first step, the flush of the last step's tails comes after the last entry): sum(timed_ms) + flush = the timed region")
        return sec

    if args.instrumented_only:
        args.warmup = args.steps = 0
    step_trace = {} if (args.steps and args.steps + args.warmup <= 400) else None
    elapsed_rank = elapsed = timed_region(dp, args.warmup, args.steps, trace=step_trace)
    per_rank = None
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if args.steps:
            mine = torch.tensor([args.steps * B / elapsed_rank], device=dev, dtype=torch.float64)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            per_rank = [round(float(v.item()), 2) for v in every]     # each rank's own images/sec between the two barriers

    # N > 1: the N = 1 reference of the efficiency -- given (--n1-value, e.g. the driver's N = 1 line) or measured here: rank 0 alone
    # runs the SAME step with no exchange (every other GPU idle at a barrier), right after the timed region, same process, warm
    n1 = None
    if world > 1 and args.steps:
        if args.n1_value:
            n1 = {"value": float(args.n1_value), "source": "--n1-value"}
        else:
            if rank == 0:
                sec = timed_region(graphed, 2, args.solo_steps, collective=False)
                n1 = {"value": args.solo_steps * B / sec, "source": f"same process: rank 0 alone, {args.solo_steps} steps of the same step with no "
                                                                    "exchange, the other ranks idle at a barrier"}
            dist.barrier()

    # second flavour of the default run: the same step in IEEE half (the reference's own evaluation dtype), timed in the same invocation
    f16 = None
    if world == 1 and args.dtype == "bf16" and args.steps and not args.no_second_flavour and not args.instrumented_only:
        mv.set_compute_dtype(torch.float16)
        g16 = GraphedForward(mv, use_graph=not args.no_graph, images_per_step=B, batch_vit=not args.no_batch_vit,
                             pipeline=not args.no_pipeline, any_size=args.stream == "coco", semantic=sem_meta, mask_format=mask_format,
                             input_resize=(S, S), input_format="RGB")
        # the SAME protocol as the bf16 region (round 5 timed 20 steps behind 3 warm-up steps next to 100 behind 10: with one extra
        # replay per region for the flush, that alone read 4 % low): same warm-up, same number of steps, its own step trace
        n16 = args.f16_steps or args.steps
        trace16 = {} if step_trace is not None else None
        sec = timed_region(g16, args.warmup, n16, collective=False, trace=trace16)
        f16 = {"value_f16": n16 * B / sec, "ms_per_step_f16": 1e3 * sec / n16, "steps_f16": n16, "warmup_f16": args.warmup}
        if trace16:
            f16["step_trace_f16"] = trace16
        del g16
        mv.set_compute_dtype(DTYPES[args.dtype])

    result = None
    if rank == 0:
        # instrumented pass (eager, NOT part of the timed region): the SAME step composition as the timed steps -- one batched
        # ViT pass over B images, then B tails -- with every branch inline, so each GEMM runs alone between its HIP events;
        # grouped by the kernel symbol the library reports.  The roofline object is about the symbol with the largest total.
        reps = 3
        n_tok = (mv.backbone.net.img_size // mv.backbone.net.patch_size) ** 2

        def eager_step(i):
            batch = [images[(i * B + b) % len(images)].contiguous() for b in range(B)]
            x = mv.backbone.net.forward_tokens(batch if B > 1 else batch[0], mv._mean, mv._std)
            for b in range(B):
                out = mv.forward_single(batch[b], text, vit_feat=x[b * n_tok:(b + 1) * n_tok], semantic=sem_meta)
                hh, ww = batch[b].shape[-2:]
                mv.postprocess_instance(out, (hh, ww), hh, ww)

        with ops.inline_forks():
            # one unmetered pass first: the timed region replayed a graph (private memory pool), so the first EAGER pass makes the
            # caching allocator hipMalloc its blocks -- a host stall of tens of ms that would land between an event pair while
            # the GPU sits idle (seen once: one K = 256 GEMM "took" 44 ms and became the dominant symbol)
            eager_step(0)
            torch.cuda.synchronize()
            with GemmMeter(ops) as meter:
                for i in range(reps):
                    eager_step(i)
                groups = meter.summary()
        reps = reps * B                                     # images in the instrumented pass
        dom_name, (dom_n, dom_t, dom_fl, dom_pair) = groups[0]
        all_t, all_fl = sum(g[1][1] for g in groups), sum(g[1][2] for g in groups)
        achieved = dom_fl / dom_t / 1e12
        traffic, traffic_note = pmc_traffic_bytes(dom_name)
        result = {
            "metric": f"images/sec @{S}^2 APE-L_D fwd", "value": (world * args.steps * B / elapsed) if args.steps else None, "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": (1e3 * elapsed / args.steps) if args.steps else None,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": workload_string(args.size, B, args.classes, args.semantic) + (f" -- ABLATED {ablate}: NOT the benchmark" if ablate else ""),
                       "parallelism": f"dp{world}", "rccl_ranks": dist.get_world_size() if dist is not None else 1,
                       "cpu_affinity": affinity,
                       "graph": not args.no_graph, "pipelined_d2h": True, "images_per_step": B, "input": args.input,
                       "input_note": ("uint8 BGR originals (%d x %d) in pinned host memory -> H2D -> resize kernel (Pillow-exact) + BGR->RGB + "
                                      "float CHW -> forward, all inside the timed region" % (S * 5 // 4, S * 5 // 4)) if args.input == "uint8"
                                     else "float32 model-ready images resident in HBM (PCIe-exclusive, the tier's definition of `value`)",
                       "mask_format": mask_format,
                       "mask_format_note": "every rank delivers its images' [k, H, W] bitmasks to its own host (the 1-GPU contract); N > 1 adds the "
                                           "device-side run-length encoding and the all-gather of records + runs" if mask_format in ("bitmask", "both")
                                           else "run lengths only (the evaluators' wire format); NOT comparable with a bitmask line",
                       "records_exchange": "all-gather of records + mask run lengths, awaited one step late by the host" if world > 1 else "none (1 rank)",
                       "host_MB_per_s_per_rank": round((args.steps * B / elapsed if args.steps else 0.0) * (
                           (mv.test_topk_per_image * S * S if mask_format in ("bitmask", "both") else 0)
                           + (mv.test_topk_per_image * graphed.rle_cap * 4 if mask_format in ("rle", "both") else 0)
                           + mv.test_topk_per_image * 32) / 1e6, 1),
                       "batched_vit": not args.no_batch_vit, "stream": args.stream,
                       "software_pipeline": (not args.no_pipeline) and "ViT of step i+1 overlaps the tails of step i; the last step is flushed inside the timed region"},
            "roofline": {"bound": "mfma", "kernel": dom_name, "achieved": achieved,
                         "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_BF16_PEAK_TFLOPS,
                         "traffic": traffic, "traffic_source": traffic_note, "launches_per_image": dom_n / reps,
                         "avg_launch_us": 1e6 * dom_t / max(dom_n, 1), "kernel_ms_per_image": 1e3 * dom_t / reps,
                         "flops_per_launch": dom_fl / max(dom_n, 1),
                         # every family of the eight-wave tile kernel (template gemm_bf16_p8_kernel, folded by tile width): the figures of the
                         # family the object is NOT about stay on the line
                         "tile_kernel_families": {k: {"kernel_ms_per_image": round(1e3 * v[1] / reps, 3), "launches_per_image": round(v[0] / reps, 2),
                                                      "avg_launch_us": round(1e6 * v[1] / max(v[0], 1), 2), "tflops": round(v[2] / v[1] / 1e12, 1),
                                                      "frac": round(v[2] / v[1] / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)}
                                                  for k, v in groups if "_p8_kernel<" in k},
                         "avg_launch_us_event_pair": 1e6 * dom_pair / max(dom_n, 1), "frac_event_pair": dom_fl / dom_pair / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                         "metering": "an eager pass of the step's composition, every branch inline (one kernel at a time, stand-alone launch "
                                     "durations); each launch carries its OWN (start, stop) HIP events (hipExtLaunchKernelGGL through the library's "
                                     "launch meter, csrc/meter.cpp): their elapsed time is the dispatch's begin-to-end duration, the number "
                                     "rocprofv3's kernel trace reports for the same launch (profiles/).  `avg_launch_us_event_pair` / "
                                     "`frac_event_pair` = the round-1..4 method, an event pair RECORDED around the call: it adds the "
                                     "command-processor gaps either side of the kernel",
                         # the dominant family per problem shape M x N x K (launches per image, average us, TFLOP/s)
                         "by_shape": {sh: {"launches_per_image": round(v[0] / reps, 2), "avg_launch_us": round(1e6 * v[1] / max(v[0], 1), 1),
                                           "tflops": round(v[2] / v[1] / 1e12, 1)}
                                      for (fam, sh), v in sorted(meter.shapes.items(), key=lambda kv: -kv[1][1]) if fam == dom_name},
                         "all_gemm_kernels": {"ms_per_image": 1e3 * all_t / reps, "tflops": all_fl / all_t / 1e12,
                                              "flops_per_image": all_fl / reps,
                                              "by_kernel_ms_per_image": {k: round(1e3 * v[1] / reps, 3) for k, v in groups},
                                              "by_kernel_tflops": {k: round(v[2] / v[1] / 1e12, 1) for k, v in groups},
                                              "by_kernel_launches_per_image": {k: round(v[0] / reps, 2) for k, v in groups},
                                              # every (kernel family, M x N x K) with >= 0.05 ms per image: launches, average us, TFLOP/s
                                              # every kernel the library launched in the instrumented pass, by launch-site expression
                                              # (non-GEMM kernels included): [launches per image, average us, ms per image], >= 0.05 ms
                                              "library_kernels": {k: [round(v[0] / reps, 2), round(1e6 * v[1] / max(v[0], 1), 1), round(1e3 * v[1] / reps, 3)]
                                                                  for k, v in sorted(meter.kernels.items(), key=lambda kv: -kv[1][1])
                                                                  if 1e3 * v[1] / reps >= 0.05},
                                              "by_kernel_and_shape": {f"{fam} {sh}": [round(v[0] / reps, 2), round(1e6 * v[1] / max(v[0], 1), 1), round(v[2] / v[1] / 1e12, 1)]
                                                                      for (fam, sh), v in sorted(meter.shapes.items(), key=lambda kv: -kv[1][1])
                                                                      if 1e3 * v[1] / reps >= 0.05}}},
        }
        if args.steps:
            result["per_rank_images_per_s"] = per_rank if per_rank is not None else [round(result["value"], 2)]
        if n1 is not None:
            # T_1 / (n T_n) of SURVEY 8d config 4, as throughput: whole-job images/sec over N x the 1-GPU images/sec of the same workload
            result["n1_reference"] = n1
            result["efficiency_vs_n1"] = result["value"] / (world * n1["value"])
        if f16 is not None:
            result.update(f16)
        if step_trace:
            result["step_trace"] = step_trace
        if world == 1 and not args.no_cpu_baseline:
            try:
                timed = [im.contiguous() for im in images[:max(1, args.cpu_images)]]
                result["cpu_baseline"], O = cpu_baseline(model, args.size, timed, text, n_images=args.cpu_images)
                result["parity"] = parity_object(mv, timed, text, O, args.ap_images, timed=args.dtype)
                try:
                    result["parity"]["vs_same_rounding_oracle"] = same_rounding_parity(model, args.size, timed[0], text, args.dtype)
                except NotImplementedError as exc:        # the rounded oracle covers the APE-L_D path (BASELINE's configurations)
                    result["parity"]["vs_same_rounding_oracle"] = {"skipped": str(exc)}
                mv.set_compute_dtype(DTYPES[args.dtype])
            except Exception as exc:  # the baseline must never take the GPU number down with it
                import traceback
                result.setdefault("cpu_baseline", {"value": None, "unit": "images/sec", "cores": os.cpu_count(), "kind": "port",
                                                   "sample": f"failed: {type(exc).__name__}: {exc}"})
                result.setdefault("parity", {"failed": f"{type(exc).__name__}: {exc}", "trace": traceback.format_exc()[-600:]})
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
