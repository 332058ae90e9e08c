"""hipGraph-replayed, double-buffered forward: the per-image launch sequence (~700 kernels) is captured once per
(image size, vocabulary, prompt mode) -- or once per (vocabulary, prompt mode) with `any_size=True` -- and replayed, and the
results of image i travel to the host on a copy stream while image i+1 is being computed.

The forward of DeformableDETRSegmVL.forward_single is a fixed sequence of launches with fixed shapes and no host
synchronisation (data-dependent selection / NMS / top-k are fixed-shape device code), which makes it capturable with
torch.cuda.CUDAGraph (a hipGraph on ROCm): our C-ABI launchers enqueue on torch's current stream, which is the
capturing stream inside `torch.cuda.graph`.

Results leave the device through TWO slots (device staging + pinned host buffers).  The last kernel of the forward
(paste_bits, which writes the [k, H, W] instance masks: 105 MB for 100 detections at 1024^2) runs outside the graph and
writes straight into the slot's staging buffer; a copy stream then moves the slot to pinned memory behind an event.
`submit()` enqueues an image and returns a ticket, `result(ticket)` waits for that slot's copy only -- so the ~2 ms
PCIe transfer of one image overlaps the compute of the next.  `__call__` = result(submit(...)) is the synchronous form.

`images_per_step = B > 1`: the ViT runs ONCE over the B images (every linear at B x 4096 rows: the 256 x 256-tile GEMM
kernels need that many rows to fill the chip; rows are independent, nothing changes numerically) and everything behind it
is one batch-1 forward per image, the B of them PARALLEL BRANCHES of one graph (ops.fork).

`pipeline=True` (software pipeline over steps): the graph of a step holds the ViT of the NEW images as one branch and the
rest of the forward ("tails": FPN, encoder, proposal selection, decoder, heads, NMS, masks) of the PREVIOUS step's images as
B more branches, reading the ViT features the previous replay left in a persistent buffer.  The GEMM-bound ViT of one batch
then overlaps the latency-bound tails of another instead of all images sitting in the same phase at the same time; a
ticket's results exist after the NEXT submit (or a flush).

`any_size=True` (mixed-size streams, BASELINE config 4): the graph is SIZE-AGNOSTIC.  Everything that depends on an image's
(h, w) inside the square pad is data, not launch geometry: the image is written into a fixed S x S canvas (pixels outside
the image = the per-channel mean, which normalise to the exact zeros the reference pads with), the per-size constants
(padding masks, sine position embeddings + level embeddings, valid ratios, encoder reference points, anchors, box limits)
are copied into fixed `StaticGeometry` buffers before the replay, and the rescale to the output frame reads a device
vector.  One graph then serves every size -- including different sizes inside one step -- with the results of the
per-size path (tests/test_model_gpu.py::test_any_size_runtime).
"""
import ctypes
import os
import queue
import threading
import weakref
from types import SimpleNamespace

import torch

from .structures import make_instances


# Captured graphs are never destroyed while the process lives: on this stack (ROCm 7.2, torch 2.10) destroying a hipGraph
# whose capture used side streams, then capturing and launching further graphs, crashed hipGraphLaunch (reproduced:
# tests/test_model_gpu.py in suite order; keeping the graph objects alive removes it).  Evicted / orphaned graphs are parked
# here instead -- their private memory pools stay allocated, which is why a mixed-size stream should use `any_size=True`
# (one graph) instead of churning through per-size graphs.
_RETIRED = []                    # [(graph object, bytes of its private pool)]
# ... but not without bound: the parked pools are HBM a long-running server never gets back.  Parking more than this many bytes raises
# (with what to do about it) instead of letting the process creep towards an out-of-memory: APE_GRAPH_RETIRE_LIMIT_GB, default 64 GB of
# the 288 GB -- about 20 evicted full-size two-image step graphs of ~3 GB each.
RETIRE_LIMIT_BYTES = int(float(os.environ.get("APE_GRAPH_RETIRE_LIMIT_GB", "64")) * (1 << 30))


def retired_graphs():
    """(number of parked graphs, bytes of HBM their private pools hold) -- process wide"""
    return len(_RETIRED), sum(b for _, b in _RETIRED)


class _DmaTransfer:
    """Helper thread that moves finished result slots to pinned host memory on a DMA ENGINE (csrc/hostcopy.cpp ape_hip_sdma_d2h).

    hipMemcpyAsync (what `tensor.copy_(..., non_blocking=True)` issues) runs a device -> pinned-host copy as a shader blit on this stack:
    a PCIe-bound kernel of ~1.9 ms per 105 MB image that keeps a few CUs occupied, so every GEMM launch with one workgroup per CU
    pays a second round while it runs (profiles/r06_d2h_blit_vs_sdma_probe.txt: +16-18 % on a GEMM loop with copies in flight, +0.2-1.1 %
    with the same bytes on the DMA engine).  The HSA copy is not stream-ordered, so this thread waits for the slot's `computed` event on
    the host, issues the blocking copies (ctypes releases the GIL), and sets `done`; `GraphedForward.result()` waits for it."""

    def __init__(self, device):
        self.device = device
        self.q = queue.SimpleQueue()
        self.thread = threading.Thread(target=self._run, name="ape-dma-transfer", daemon=True)
        self.thread.start()

    def _run(self):
        from . import _lib
        lib = _lib.load()
        torch.cuda.set_device(self.device)
        while True:
            job = self.q.get()
            if job is None:
                return
            ev, copies, done, err = job
            try:
                ev.synchronize()                              # the paste kernels of this slot have finished (host-side wait)
                # every image's copy in flight at once, each in pieces over the free copy engines (one blocking copy after the other keeps
                # one engine busy: the 2 x 1.18 GB per step of the 1536^2 / top-500 configuration then take longer than the step's kernels)
                m = len(copies)
                if os.environ.get("APE_SDMA_MULTI") == "0":   # A/B: one blocking copy after the other (the first round-6 form)
                    for dst, src, n in copies:
                        if lib.ape_hip_sdma_d2h(dst, src, n) != 0:
                            raise RuntimeError("ape_amd.runtime: " + (lib.ape_hip_last_error() or b"ape_hip_sdma_d2h failed").decode())
                    continue
                # pieces placed on the copy engines the runtime reports free for device -> host (the form measured inside the pipeline:
                # profiles/r06_config5_transfer.txt); APE_SDMA_ENGINES=0 leaves the placement of the concurrent copies to the runtime
                # ... in a single-process run.  With several ranks on one node (torch.distributed) the copies stay unplaced unless the
                # environment says otherwise: explicit engine picks were only ever measured on 1-GPU boxes, and at 1024^2 (105 MB per
                # image) one engine is enough
                multi = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
                os.environ.setdefault("APE_SDMA_ENGINES", "0" if multi else "1")
                dsts = (ctypes.c_void_p * m)(*[c[0] for c in copies])
                srcs = (ctypes.c_void_p * m)(*[c[1] for c in copies])
                sizes = (ctypes.c_size_t * m)(*[c[2] for c in copies])
                if lib.ape_hip_sdma_d2h_multi(m, dsts, srcs, sizes, 0) != 0:
                    raise RuntimeError("ape_amd.runtime: " + (lib.ape_hip_last_error() or b"ape_hip_sdma_d2h_multi failed").decode())
            except BaseException as exc:                      # surfaced by result()
                err.append(exc)
            finally:
                done.set()

    def submit(self, event, copies):
        done, err = threading.Event(), []
        self.q.put((event, copies, done, err))
        return done, err

    def close(self):
        self.q.put(None)


class _Ticket:
    """handle of a submitted step (weak-referenceable, unlike SimpleNamespace)"""
    __slots__ = ("entry", "slot", "ready", "single", "rec6", "records", "frames", "model_hw", "runs", "mask_runs", "sem_labels", "panoptic", "__weakref__")

    def __init__(self, entry, single, frames, model_hw=None):
        self.entry, self.single, self.frames = entry, single, frames      # frames: [(height, width)] of the output masks
        self.model_hw = model_hw                      # [(h, w)] of the model inputs (any_size: where each image's valid region ends)
        self.slot, self.ready, self.rec6, self.records = None, False, None, None
        self.sem_labels = None                        # GraphedForward(semantic=...): per-image label maps [fh, fw] int16 (host)
        self.panoptic = None                          # GraphedForward(panoptic=...): per-image (panoptic_seg int32 [fh, fw], segments_info)
        self.runs, self.mask_runs = None, None        # mask_format="rle": (device run lengths [B,k,cap], run counts [B,k]); all ranks'



class GraphedForward:
    SLOTS = 2

    def __init__(self, model_vision, use_graph=True, max_graphs=16, with_masks=True, images_per_step=1, batch_vit=True,
                 pipeline=False, any_size=False, max_out_pixels=None, mask_format="bitmask", rle_cap=4096, semantic=None, panoptic=None,
                 input_resize=(1024, 1024), input_format="RGB"):
        self.mv = model_vision
        # uint8 inputs (the predictor's contract, ape/engine/defaults.py:213-222: the ORIGINAL BGR image): submit() takes uint8
        # [H, W, 3] tensors -- pinned host memory or device -- and runs upload + ResizeShortestEdge(short_edge_length, max_size)
        # + BGR->RGB + float CHW (csrc/imageio.hip resize_u8_kernel, bit exact with Pillow) straight into the step's static
        # image buffer, stream-ordered in front of the replay; the output frame defaults to the ORIGINAL (H, W)
        self.input_resize = (int(input_resize[0]), int(input_resize[1]))
        self.input_flip = input_format == "RGB"
        self.batch_vit = batch_vit
        self.pipeline = bool(pipeline)
        self.any_size = bool(any_size)
        self.use_graph = use_graph
        self.max_graphs = max_graphs
        self.with_masks = with_masks
        self.B = int(images_per_step)
        self.max_out_pixels = max_out_pixels          # any_size: largest output frame (height * width) a slot can hold
        # "bitmask": pred_masks [n, H, W] bool on the host (the reference's output contract, ~1 MB per mask over PCIe);
        # "rle": the evaluators' wire format instead -- the pasted masks are run-length encoded on the device
        # (ops.rle_encode) and only the runs travel; instances carry `pred_masks_rle` (ape_amd/evaluation.py);
        # "both": the bitmasks reach this rank's host exactly as with "bitmask" AND the run lengths are produced next to them
        # (`ticket.runs`, what dp.DataParallelRunner(gather_masks=True) all-gathers across ranks): a data-parallel rank then does
        # everything a single-GPU run does, plus the exchange
        if mask_format not in ("bitmask", "rle", "both"):
            raise ValueError("GraphedForward: mask_format is 'bitmask', 'rle' or 'both'")
        self.mask_format, self.rle_cap = mask_format, int(rle_cap)
        # semantic branch inside the captured step (deformable_detr_segm_vl.py:628-666 + sem_seg_postprocess :875-918): `semantic`
        # = the dataset's metadata dict (thing_classes / stuff_classes / entity, as model.forward builds it).  The [K', H, W] score
        # volume stays on the device (1.3 GB per 1536^2 image with 134 classes); what leaves is its per-pixel argmax -- the
        # label map every semantic evaluator reduces the scores to -- as int16 [H, W] (`ticket.sem_labels`).  With any_size the
        # captured step produces the class scores over the whole S x S pad (fixed shape); the crop to the image's own (h, w), the
        # resize to its output frame and the argmax run behind the replay with the ticket's sizes, like the mask paste does.
        self.semantic = semantic
        # panoptic branch inside the captured step (:671-690 + _postprocess_panoptic :921-998): `panoptic` = the evaluation dataset's
        # metadata dict (thing_classes, stuff_classes, thing_dataset_id_to_contiguous_id).  The merge runs on the device without a host
        # round trip (csrc/masks.hip panoptic_*), so it is part of the graph; `ticket.panoptic` = [(panoptic_seg, segments_info)].
        # With any_size the captured step produces the panoptic queries' mask logits over the whole S x S pad and their class
        # logits (fixed shapes); the merge itself -- crop to the image's own (h, w), resize to its output frame, the walk over
        # the queries: three launches, no host round trip -- runs behind the replay with the ticket's sizes, like the mask paste.
        self.panoptic = panoptic
        # pipelined steps: start the ViT branch behind the tails' encoders (see _run_entry); APE_PIPE_LATE_VIT=0|1 overrides
        self.late_vit = os.environ.get("APE_PIPE_LATE_VIT", "0") == "1"
        # pipelined steps: the first step of a stream / the flush behind its last step run as ViT-only / tails-only graphs (see
        # _build); APE_PIPE_PARTIAL=0 replays the full step there, as rounds 2-5 did
        self.partial_graphs = os.environ.get("APE_PIPE_PARTIAL", "1") != "0"
        self._graphs = {}
        self._copy_stream = None
        # mask transfer: "sdma" = the library's DMA-engine copy on a helper thread (default), "blit" = hipMemcpyAsync on the copy stream
        # (a shader blit on this stack; rounds 1-5) -- APE_D2H=blit|sdma
        self.d2h = os.environ.get("APE_D2H", "sdma")
        if self.d2h not in ("sdma", "blit"):
            raise ValueError("APE_D2H is 'sdma' or 'blit'")
        self._dma = None

    def _retire(self, entry, strict=True):
        if getattr(entry, "graph", None) is None:
            return
        for name in ("graph_vit", "graph_tails"):         # the partial graphs of a pipelined entry share its fate (and its byte count)
            if getattr(entry, name, None) is not None:
                _RETIRED.append((getattr(entry, name), 0))
        nbytes = int(getattr(entry, "pool_bytes", 0))
        n, held = retired_graphs()
        if strict and held + nbytes > RETIRE_LIMIT_BYTES:
            raise RuntimeError(
                f"GraphedForward: evicting this graph would park {(held + nbytes) / (1 << 30):.1f} GB of HBM in {n + 1} retired hipGraphs "
                f"(limit {RETIRE_LIMIT_BYTES / (1 << 30):.0f} GB, APE_GRAPH_RETIRE_LIMIT_GB).  Captured graphs are never destroyed on this "
                "stack, so a stream that keeps changing (image size, vocabulary, frame) grows without bound: use any_size=True (one "
                f"size-agnostic graph), raise max_graphs (now {self.max_graphs}) above the number of distinct keys, or raise the limit.")
        _RETIRED.append((entry.graph, nbytes))

    def memory_report(self):
        """{"live_graphs", "live_bytes", "retired_graphs", "retired_bytes", "retire_limit_bytes"}: HBM held by captured steps"""
        n, held = retired_graphs()
        return {"live_graphs": len(self._graphs), "live_bytes": sum(int(getattr(e, "pool_bytes", 0)) for e in self._graphs.values()),
                "retired_graphs": n, "retired_bytes": held, "retire_limit_bytes": RETIRE_LIMIT_BYTES}

    def __del__(self):
        try:
            if self._dma is not None:
                self._dma.close()
            for e in self._graphs.values():
                self._retire(e, strict=False)     # a destructor parks unconditionally
        except Exception:       # interpreter shutdown
            pass

    # ------------------------------------------------------------------ device work of one image
    def _device_part(self, image, text, height, width, frame, prompt="name", vit_feat=None, geo=None, encoder_done=None):
        """everything up to (excluding) the mask paste: (record [k,8], 128x128 masks or None, boxes in the output frame).
        frame: device vector [8] = (sx, sy, sx, sy, width, height, width, height) of the output frame (rewritten per image
        with any_size, constant otherwise)."""
        import math
        from . import ops
        mv = self.mv
        out = mv.forward_single(image, text, with_masks=self.with_masks, prompt=prompt, vit_feat=vit_feat, geo=geo,
                                encoder_done=encoder_done, semantic=self.semantic, panoptic=self.panoptic is not None)
        labels = None
        if self.semantic is not None:
            # any_size: (height, width) = the pad; the scores stay [K', S, S] here and become labels in _replay (ticket sizes)
            labels = out["sem_seg"] if self.any_size else self._sem_labels(out["sem_seg"], height, width)
        pan = None
        if self.panoptic is not None:
            # any_size: (height, width) = the pad; the pieces stay fixed-shape here and are merged in _replay (ticket sizes)
            pan = ({k: out.get(k) for k in ("pan_masks", "pan_cls", "pan_valid_score")} if self.any_size
                   else mv.panoptic_device(out, height, width, self.panoptic))
        # boxes in the output frame, keep flags, records with the kept detections first (stable) -- the host then takes PREFIX
        # views of the pinned buffers instead of gathering ~1 MB per mask with a boolean index; dropped rows (empty slots, empty
        # boxes after the rescale) carry score -1, so the 6-column view that is all-gathered across ranks tells kept from dropped
        # without the keep column.  One launch (csrc/boxes.hip) + one row gather of the 128 x 128 masks.
        rec, boxes, order = ops.det_records(out["det_boxes"], out["det_scores"], out["det_classes"], out["det_query"], frame)
        masks128 = out.get("det_masks128")
        if masks128 is not None:
            n = masks128.shape[0]
            masks128 = ops.gather_rows(masks128.view(n, -1).view(torch.float32), order).view(torch.uint8).view(masks128.shape)
        return rec, masks128, boxes, labels, pan

    def _sem_labels(self, sem, height, width):
        """sem_seg_postprocess (:875-918) on the class scores [K', h, w] of the image's own region -> int16 labels [height, width]"""
        import math
        from . import ops
        mv = self.mv
        r = ops.bilinear_resize(sem, height, width)                             # (:916)
        meta = self.semantic
        class0 = None
        if (mv.eval_dataset_id >= 0 and meta.get("entity") == "stuff" and (meta.get("stuff_classes") or [""])[0] == "things"
                and mv.stuff_prob_thing > 0):                                                # (:654-663)
            class0 = math.log(mv.stuff_prob_thing / (1 - mv.stuff_prob_thing))
        return ops.argmax_labels(r, class0)                                                  # library launch (round 5: torch argmax + cast)

    def _tail(self, e, b, vit_feat, encoder_done=None):
        height, width = e.size
        return self._device_part(e.images[b], e.text, height, width, e.frame[b], e.prompt, vit_feat,
                                 e.sgeo[b] if self.any_size else None, encoder_done)

    def _run_entry(self, e, part="full"):
        """the device work of one step of entry `e` on its static buffers: (ViT of the images) + B tails.  Pipelined mode, `part`:
        "full" = ViT of the new images || tails of the previous step's; "vit" = the ViT branch only (the FIRST step of a stream: no
        previous images whose tails could run); "tails" = the tails only (the FLUSH behind the last step: no new images)."""
        from . import ops
        mv = self.mv
        if e.prompt == "expression" and mv.test_topk_per_image != 1:      # (:183-194) forward() applies the same rule
            saved = mv.test_topk_per_image
            mv.test_topk_per_image = 1
            try:
                return self._run_entry(e, part)
            finally:
                mv.test_topk_per_image = saved
        B = len(e.images)
        net = mv.backbone.net
        n_tok = (net.img_size // net.patch_size) ** 2
        if self.pipeline:
            # ViT of the NEW images || tails of the features the previous replay left in e.feat.  Branches do not fork again
            # (nested fork/join inside image branches made hipStreamEndCapture crash on this ROCm).
            with ops.inline_forks():
                feats = [e.feat[b * n_tok:(b + 1) * n_tok] for b in range(B)]
                if part == "vit":
                    e.feat.copy_(net.forward_tokens(e.images if B > 1 else e.images[0], mv._mean, mv._std))
                    return None
                if part == "tails":
                    jobs = [ops.fork(lambda b=b: self._tail(e, b, feats[b]), force=True) for b in range(1, B)]
                    return [self._tail(e, 0, feats[0])] + [j.join() for j in jobs]
                if self.late_vit:
                    # ViT of the new images ordered BEHIND the encoders of the tails: the tails' GEMM-bound first half (FPN,
                    # encoder) then has the chip to itself, and the GEMM-bound ViT overlaps their latency-bound second half
                    # (selection, decoder, heads, NMS, masks) instead of both heavy phases fighting for the CUs first and
                    # the light chains running alone at the end.  Every tail on a side stream, so the ViT branch depends on
                    # nothing but the encoder events.
                    done = [torch.cuda.Event() for _ in range(B)]
                    jobs = [ops.fork(lambda b=b: self._tail(e, b, feats[b], done[b]), force=True) for b in range(B)]

                    def vit():
                        cur = torch.cuda.current_stream()
                        for ev in done:
                            cur.wait_event(ev)
                        return net.forward_tokens(e.images if B > 1 else e.images[0], mv._mean, mv._std)

                    vjob = ops.fork(vit, force=True)
                    outs = [j.join() for j in jobs]
                else:
                    vjob = ops.fork(lambda: net.forward_tokens(e.images if B > 1 else e.images[0], mv._mean, mv._std), force=True)
                    jobs = [ops.fork(lambda b=b: self._tail(e, b, feats[b]), force=True) for b in range(1, B)]
                    outs = [self._tail(e, 0, feats[0])] + [j.join() for j in jobs]
                e.feat.copy_(vjob.join())             # behind every tail: the next replay reads the new features
            return outs
        if B == 1:
            return [self._tail(e, 0, None)]
        with ops.inline_forks():
            feats = [None] * B
            if self.batch_vit:
                x = net.forward_tokens(e.images, mv._mean, mv._std)
                feats = [x[b * n_tok:(b + 1) * n_tok] for b in range(B)]
            jobs = [ops.fork(lambda b=b: self._tail(e, b, feats[b]), force=True) for b in range(1, B)]
            return [self._tail(e, 0, feats[0])] + [j.join() for j in jobs]

    # ------------------------------------------------------------------ inputs: float CHW model images or uint8 HWC originals
    @staticmethod
    def _is_raw(im):
        return im.dtype == torch.uint8 and im.dim() == 3 and im.shape[-1] == 3

    def _model_hw(self, im):
        """(h, w) of the model input an image becomes"""
        if self._is_raw(im):
            from .engine import shortest_edge_size
            return shortest_edge_size(int(im.shape[0]), int(im.shape[1]), *self.input_resize)
        return int(im.shape[-2]), int(im.shape[-1])

    def _write_input(self, dst, im):
        """im -> dst (float32 [3, h, w] view of a static image buffer), stream-ordered: a copy for model-ready images; upload
        (pinned source: asynchronous) + the resize kernel for uint8 originals"""
        if not self._is_raw(im):
            dst.copy_(im, non_blocking=True)
            return
        from . import ops
        raw = im if im.is_cuda else im.to(dst.device, non_blocking=True)
        ops.resize_bilinear_u8(raw.contiguous(), dst.shape[-2], dst.shape[-1], out=dst, float_chw=True, flip=self.input_flip)

    # ------------------------------------------------------------------ per-image inputs of the size-agnostic graph
    def _level_shapes(self):
        S = self.mv.backbone.padding_constraints.get("square_size", 0)
        strides = [self.mv.backbone._out_feature_strides[f] for f in self.mv.neck.in_features]
        return S, [(S // st, S // st) for st in strides]

    def _geometry_into(self, sgeo, S, h, w):
        """the constants of an (h, w) image, generated on the device into the graph's fixed buffers (one launch)"""
        mv = self.mv
        if not mv.position_embedding.normalize:
            raise NotImplementedError("ape_amd: the geometry kernel implements the normalised sine embedding of the APE configs")
        sgeo.generate(S, (h, w), mv.pos_cfg(), mv.transformer.packed(mv.compute_dtype)["level_embeds"])

    def _load_inputs(self, e, images, frames):
        """any_size: write image b into its canvas, the constants of its (h, w) into its StaticGeometry, and its output frame
        into the frame vector -- all stream-ordered in front of the replay"""
        mv = self.mv
        S, shapes = self._level_shapes()
        dt = mv.compute_dtype
        vals = []
        for b, im in enumerate(images):
            h, w = self._model_hw(im)
            if h > S or w > S:
                raise ValueError(f"GraphedForward(any_size): image {h}x{w} does not fit the {S}x{S} pad")
            e.images[b].copy_(e.mean_canvas, non_blocking=True)
            self._write_input(e.images[b][:, :h, :w], im)
            self._geometry_into(e.sgeo[b], S, h, w)
            height, width = frames[b]
            sx, sy = width / w, height / h
            vals.append([sx, sy, sx, sy, width, height, width, height])
        e.frame.copy_(torch.tensor(vals, dtype=torch.float32))              # pageable source: staged before the call returns

    # ------------------------------------------------------------------ capture
    def _build(self, images, text, height, width, prompt):
        from .modeling.ape_deta.geometry import StaticGeometry
        mv = self.mv
        dev = next(mv.parameters()).device
        B = len(images)
        e = SimpleNamespace()
        e.text = text                 # keeps the bank alive: the graph key holds its address
        e.size, e.prompt = (height, width), prompt
        # warm-up and capture run fusion_tokens, which (phrase / expression prompts, persistent bank) shifts the phrase bank
        # in place: snapshot it and restore it afterwards so that building a graph does not count as three extra images
        bank = mv.features_phrase_bank.clone() if getattr(mv, "text_feature_bank", False) else None
        if self.any_size:
            S, shapes = self._level_shapes()
            dt = mv.compute_dtype
            e.mean_canvas = torch.tensor(mv._mean, dtype=torch.float32, device=dev).view(3, 1, 1).expand(3, S, S).contiguous()
            e.images = [e.mean_canvas.clone() for _ in range(B)]
            g0 = mv.geometry((S, S), shapes)
            lp = mv.transformer.lvl_pos(g0, dt)
            key = next(iter(g0._lvl_pos))
            e.sgeo = [StaticGeometry(g0, key, lp) for _ in range(B)]
            e.frame = torch.zeros((B, 8), dtype=torch.float32, device=dev)
            self._load_inputs(e, images, [self._model_hw(im) for im in images])
            e.size = (S, S)
        else:
            e.images = []
            vals = []
            for im in images:
                mh, mw = self._model_hw(im)
                e.images.append(torch.empty((3, mh, mw), dtype=torch.float32, device=dev))
                self._write_input(e.images[-1], im)
                sx, sy = width / mw, height / mh
                vals.append([sx, sy, sx, sy, width, height, width, height])
            e.frame = torch.tensor(vals, dtype=torch.float32).to(dev)
        if self.pipeline:
            net = mv.backbone.net
            n_tok = (net.img_size // net.patch_size) ** 2
            e.feat = torch.zeros((B * n_tok, net.embed_dim), dtype=mv.compute_dtype, device=dev)
            e.pending = None          # weakref of the ticket whose ViT features sit in e.feat, tails not run yet
        # (no closure over `e` is stored on it: a reference cycle would leave the captured graph to the garbage collector)
        for _ in range(2):            # warm every cache (weight packing, geometry, text side) outside the capture
            self._run_entry(e)
        torch.cuda.synchronize()
        if self.use_graph:
            reserved = torch.cuda.memory_reserved()
            e.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(e.graph):
                e.outs = self._run_entry(e)
            # pipelined: the two ends of a stream as their own graphs -- the first step has no previous images (ViT branch only), the
            # flush behind the last step has no new ones (tails only).  Replaying the FULL step there ran one ViT pass and B tails
            # nobody consumed per stream: K steps cost K + 1 full replays (5 % of a 20-step run).  Same launches, same static
            # buffers, the part's own output tensors.
            e.graph_vit = e.graph_tails = None
            if self.pipeline and self.partial_graphs:
                e.graph_vit = torch.cuda.CUDAGraph()
                with torch.cuda.graph(e.graph_vit):
                    self._run_entry(e, "vit")
                e.graph_tails = torch.cuda.CUDAGraph()
                with torch.cuda.graph(e.graph_tails):
                    e.outs_tails = self._run_entry(e, "tails")
            e.pool_bytes = max(0, torch.cuda.memory_reserved() - reserved)      # the captures' private pools (what eviction parks)
        else:
            e.graph = e.graph_vit = e.graph_tails = None
        if bank is not None:
            mv.features_phrase_bank.copy_(bank)
        k = 1 if prompt == "expression" else mv.test_topk_per_image       # (:183-194) one box per referring expression
        has_masks = self.with_masks and mv.test_mask_on
        e.k = k
        e.maxpix = (self.max_out_pixels or e.size[0] * e.size[1]) if self.any_size else height * width
        e.slots = []
        # rows of the panoptic segment table = the step's panoptic queries (the class-wise NMS keeps test_topk_per_image of them;
        # all queries without panoptic_post_nms)
        pq = (k if mv.panoptic_post_nms else mv.num_queries) if self.panoptic is not None else 0
        for _ in range(self.SLOTS):
            s = SimpleNamespace()
            s.d_rec = torch.empty((B, k, 8), dtype=torch.float32, device=dev)
            s.h_rec = torch.empty((B, k, 8), dtype=torch.float32, pin_memory=True)
            s.d_masks = torch.empty((B, k * e.maxpix), dtype=torch.uint8, device=dev) if has_masks else None
            rle = has_masks and self.mask_format in ("rle", "both")
            s.h_masks = torch.empty((B, k * e.maxpix), dtype=torch.uint8, pin_memory=True) if has_masks and self.mask_format != "rle" else None
            s.d_runs = torch.empty((B, k, self.rle_cap), dtype=torch.int32, device=dev) if rle else None
            s.d_nruns = torch.zeros((B, k), dtype=torch.int32, device=dev) if rle else None
            s.h_runs = torch.empty((B, k, self.rle_cap), dtype=torch.int32, pin_memory=True) if rle else None
            s.h_nruns = torch.zeros((B, k), dtype=torch.int32, pin_memory=True) if rle else None
            s.d_sem = torch.empty((B, e.maxpix), dtype=torch.int16, device=dev) if self.semantic is not None else None
            s.h_sem = torch.empty((B, e.maxpix), dtype=torch.int16, pin_memory=True) if self.semantic is not None else None
            # panoptic: the map + the segment table of the step's panoptic queries (one extra row carries the segment count)
            s.d_pan = torch.empty((B, e.maxpix), dtype=torch.int32, device=dev) if self.panoptic is not None else None
            s.h_pan = torch.empty((B, e.maxpix), dtype=torch.int32, pin_memory=True) if self.panoptic is not None else None
            s.d_paninfo = torch.zeros((B, pq + 1, 3), dtype=torch.int32, device=dev) if self.panoptic is not None else None
            s.h_paninfo = torch.zeros((B, pq + 1, 3), dtype=torch.int32, pin_memory=True) if self.panoptic is not None else None
            s.computed, s.copied = torch.cuda.Event(), torch.cuda.Event()
            s.busy = False
            e.slots.append(s)
        e.next_slot = 0
        return e

    # ------------------------------------------------------------------ pipelined interface
    @torch.no_grad()
    def submit(self, image, text, height=None, width=None, prompt="name"):
        """enqueue one image [3,h,w] (fp32, device) -- or a list of `images_per_step` images -- against the text bank [K, D]
        (device); returns a ticket.  height / width: the output frame (ints, or per-image lists with any_size); default =
        the image's own size.  Without any_size the images of one step must share a size.  At most SLOTS completed tickets
        may be outstanding per entry."""
        images = list(image) if isinstance(image, (list, tuple)) else [image]
        if len(images) != self.B:
            raise ValueError(f"GraphedForward.submit: expected {self.B} image(s) per step, got {len(images)}")
        single = not isinstance(image, (list, tuple))
        hs = height if isinstance(height, (list, tuple)) else [height] * self.B
        ws = width if isinstance(width, (list, tuple)) else [width] * self.B
        # output frame: what the caller asked for; else the image's own size (for a uint8 original: its ORIGINAL size, :222)
        own = [(int(im.shape[0]), int(im.shape[1])) if self._is_raw(im) else (int(im.shape[-2]), int(im.shape[-1])) for im in images]
        frames = [(int(hs[b] or own[b][0]), int(ws[b] or own[b][1])) for b in range(len(images))]
        mhw = [self._model_hw(im) for im in images]
        # the classifier's text side is a per-vocabulary constant baked into the capture: an in-place update of the bank
        # (text._version) must rebuild the graph, exactly like a new bank
        tkey = (text.data_ptr(), text._version, tuple(text.shape), prompt)
        if self.any_size:
            key = ("any",) + tkey
        else:
            h, w = mhw[0]
            if any(m != (h, w) for m in mhw) or any(f != frames[0] for f in frames):
                raise ValueError("GraphedForward.submit: the images of one step must share a size (or use any_size=True)")
            key = (h, w) + frames[0] + tkey
        e = self._graphs.get(key)
        if e is None:
            if len(self._graphs) >= self.max_graphs:
                oldest = next(iter(self._graphs))
                self.flush(self._graphs[oldest])       # a ticket waiting for its tails keeps its entry alive through ticket.entry
                # park FIRST, pop afterwards: when the retire limit refuses, the entry must stay live in _graphs -- popped and
                # not parked, its hipGraph would be destroyed by the garbage collector, the one thing this stack does not survive
                self._retire(self._graphs[oldest])
                del self._graphs[oldest]
            e = self._graphs[key] = self._build(images, text, frames[0][0], frames[0][1], prompt)
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=next(self.mv.parameters()).device)
        if any(f[0] * f[1] > e.maxpix for f in frames):
            raise ValueError(f"GraphedForward.submit: output frame larger than max_out_pixels={e.maxpix}")
        t = _Ticket(e, single, frames, mhw)
        if not self.pipeline:
            self._replay(e, images, frames, completes=t)
            return t
        # pipelined: this replay runs the tails of the PREVIOUS ticket and the ViT of these images
        self._replay(e, images, frames, completes=e.pending() if e.pending is not None else None)
        e.pending = weakref.ref(t)        # weak: ticket -> entry -> ticket would be a cycle; a dropped ticket's detections
        return t                          # are simply not delivered

    def _replay(self, e, images, frames, completes):
        """enqueue one step (graph replay or eager run) and, for the ticket whose detections it produces, the mask paste into a
        result slot and the slot's transfer to pinned memory on the copy stream"""
        from . import ops
        s = None
        if completes is not None:
            s = e.slots[e.next_slot]
            if s.busy:
                raise RuntimeError("GraphedForward.submit: every result slot is outstanding -- call result() on an earlier ticket first")
            e.next_slot = (e.next_slot + 1) % self.SLOTS
        cur = torch.cuda.current_stream()
        if images is not None and not (self.pipeline and self.any_size):
            if self.any_size:
                self._load_inputs(e, images, frames)
            else:
                for buf, im in zip(e.images, images):
                    self._write_input(buf, im)
        elif images is not None:
            # pipelined + any_size: the canvases feed the ViT of THIS replay (new images), the static geometry / frame feed its
            # tails (previous images).  Canvases are loaded now; geometry and frame of the new images after the replay.
            S = e.size[0]
            for b, im in enumerate(images):
                h, w = self._model_hw(im)
                if h > S or w > S:
                    raise ValueError(f"GraphedForward(any_size): image {h}x{w} does not fit the {S}x{S} pad")
                e.images[b].copy_(e.mean_canvas, non_blocking=True)
                self._write_input(e.images[b][:, :h, :w], im)
        part = "full"
        if self.pipeline and self.partial_graphs:
            part = "tails" if images is None else ("vit" if completes is None else "full")
        if e.graph is not None:
            g = {"full": e.graph, "vit": e.graph_vit, "tails": e.graph_tails}[part]
            if g is None:
                g, part = e.graph, "full"
            g.replay()
            outs = e.outs_tails if part == "tails" else e.outs
        else:
            outs = self._run_entry(e, part)
        if completes is not None:
            cur.wait_event(s.copied)                      # the slot's previous transfer has left the staging buffers
            has_masks = s.d_masks is not None and outs[0][1] is not None
            k = e.k
            for b, (rec, masks128, boxes, labels, pan) in enumerate(outs):
                s.d_rec[b].copy_(rec, non_blocking=True)
                if labels is not None:
                    if self.any_size:           # class scores over the pad -> this image's region -> its output frame
                        mh, mw = completes.model_hw[b]
                        labels = self._sem_labels(labels[:, :mh, :mw], *completes.frames[b])
                    s.d_sem[b, : labels.numel()].copy_(labels.reshape(-1), non_blocking=True)
                if pan is not None:
                    if self.any_size:           # mask logits over the pad -> this image's region -> merged in its output frame
                        mh, mw = completes.model_hw[b]
                        pan = self.mv.panoptic_device(dict(pan, pan_masks=pan["pan_masks"][:, :mh, :mw]), *completes.frames[b], self.panoptic)
                    s.d_pan[b, : pan[0].numel()].copy_(pan[0].reshape(-1), non_blocking=True)
                    s.d_paninfo[b, : pan[1].shape[0]].copy_(pan[1], non_blocking=True)
                    s.d_paninfo[b, -1, 0:1].copy_(pan[2], non_blocking=True)        # last row, column 0: the segment count
                if has_masks:
                    fh, fw = completes.frames[b]
                    pasted = ops.paste_bits(masks128, boxes, fh, fw, out=s.d_masks[b, : k * fh * fw].view(k, fh, fw))   # detector_postprocess (:869-871)
                    if s.d_runs is not None:
                        ops.rle_encode(pasted, cap=self.rle_cap, counts=s.d_runs[b], nruns=s.d_nruns[b])
            s.computed.record(cur)
            with torch.cuda.stream(self._copy_stream):
                self._copy_stream.wait_event(s.computed)
                s.h_rec.copy_(s.d_rec, non_blocking=True)
                if s.d_sem is not None:
                    s.h_sem.copy_(s.d_sem, non_blocking=True)
                if s.d_pan is not None:
                    s.h_pan.copy_(s.d_pan, non_blocking=True)
                    s.h_paninfo.copy_(s.d_paninfo, non_blocking=True)
                if has_masks and s.d_runs is not None:
                    s.h_runs.copy_(s.d_runs, non_blocking=True)
                    s.h_nruns.copy_(s.d_nruns, non_blocking=True)
                if has_masks and s.h_masks is not None and self.d2h == "blit":
                    for b in range(len(outs)):
                        fh, fw = completes.frames[b]
                        s.h_masks[b, : k * fh * fw].copy_(s.d_masks[b, : k * fh * fw], non_blocking=True)
                s.copied.record(self._copy_stream)
            s.masks_done = s.masks_err = None
            if has_masks and s.h_masks is not None and self.d2h == "sdma":
                # the [k, H, W] masks (105 MB per 1024^2 image) leave on a DMA engine: the helper thread waits for `computed` on the host.
                # The slot is not reused before result() has waited for this transfer (s.busy), so no stream-side ordering is needed.
                if self._dma is None:
                    self._dma = _DmaTransfer(s.d_masks.device)
                copies = [(s.h_masks[b].data_ptr(), s.d_masks[b].data_ptr(), k * completes.frames[b][0] * completes.frames[b][1])
                          for b in range(len(outs))]
                s.masks_done, s.masks_err = self._dma.submit(s.computed, copies)
            s.busy = True
            s.has_masks = has_masks
            completes.slot, completes.ready = s, True
            completes.rec6 = s.d_rec[:, :, :6] if self.B > 1 else s.d_rec[0, :, :6]
            if has_masks and s.d_runs is not None:
                completes.runs = (s.d_runs, s.d_nruns)
        if images is not None and self.pipeline and self.any_size:
            # now the static geometry / frame may take the NEW images' constants (stream-ordered behind the replay)
            mv = self.mv
            S, shapes = self._level_shapes()
            vals = []
            for b, im in enumerate(images):
                h, w = self._model_hw(im)
                self._geometry_into(e.sgeo[b], S, h, w)
                fh, fw = frames[b]
                sx, sy = fw / w, fh / h
                vals.append([sx, sy, sx, sy, fw, fh, fw, fh])
            e.frame.copy_(torch.tensor(vals, dtype=torch.float32))          # pageable source: staged before the call returns

    def flush(self, entry=None):
        """pipelined mode: run the tails of the ticket whose ViT features are waiting (one replay of the tails-only graph)"""
        for e in ([entry] if entry is not None else list(self._graphs.values())):
            if self.pipeline and e.pending is not None:
                t, e.pending = e.pending(), None
                self._replay(e, None, None, completes=t)

    def result(self, ticket):
        """wait for the ticket's transfer; returns (instances on the host, device record view [k,6]) -- lists / [B,k,6] when the
        step was submitted as a list.  pred_masks is a zero-copy view of the slot's pinned buffer: valid until SLOTS further
        completed steps.  Pipelined mode: a ticket completes with the NEXT submit; asking earlier flushes the pipeline."""
        e = ticket.entry
        if not ticket.ready:
            self.flush(e)
        s = ticket.slot
        s.copied.synchronize()
        if s.masks_done is not None:                        # the DMA-engine transfer of the masks (helper thread)
            s.masks_done.wait()
            if s.masks_err:
                raise s.masks_err[0]
        s.busy = False
        insts = []
        k = e.k
        for b in range(self.B):
            hr = s.h_rec[b]
            n = int((hr[:, 7] > 0.5).sum())                 # kept detections are a prefix (sorted on the device)
            fh, fw = ticket.frames[b]
            extra = {}
            masks = None
            if s.has_masks and s.h_runs is not None:
                extra["pred_masks_rle"] = self._rles(s, b, n, fh, fw)
            if s.has_masks and s.h_masks is not None:
                masks = s.h_masks[b, : k * fh * fw].view(k, fh, fw)[:n].view(torch.bool)   # zero-copy
            insts.append(make_instances((fh, fw), hr[:n, :4].clone(), hr[:n, 4].clone(), hr[:n, 5].long(), masks,
                                        query_index=hr[:n, 6].long(), **extra))
        if s.h_sem is not None:       # label maps (views of the slot's pinned buffer, valid like pred_masks)
            ticket.sem_labels = [s.h_sem[b, : fh * fw].view(fh, fw) for b, (fh, fw) in enumerate(ticket.frames)]
        if s.h_pan is not None:       # (panoptic_seg view of the slot's pinned buffer, segments_info) per image
            ticket.panoptic = [(s.h_pan[b, : fh * fw].view(fh, fw), self.mv.segments_info(s.h_paninfo[b, :-1], s.h_paninfo[b, -1, 0:1]))
                               for b, (fh, fw) in enumerate(ticket.frames)]
        if ticket.single:
            return insts[0], ticket.rec6
        return insts, ticket.rec6

    def _rles(self, s, b, n, fh, fw):
        """COCO RLE dicts of the first n (kept) detections of image b of slot s; a mask with more runs than the slot's buffer
        holds is encoded again from the slot's device masks (still this ticket's until the slot is reused)"""
        from . import evaluation
        rles = evaluation.rles_from_runs(s.h_runs[b, :n], torch.clamp(s.h_nruns[b, :n], max=self.rle_cap), (fh, fw))
        over = (s.h_nruns[b, :n] > self.rle_cap).nonzero().flatten().tolist()
        if over:
            k = s.d_rec.shape[1]
            dm = s.d_masks[b, : k * fh * fw].view(k, fh, fw)
            for i, r in zip(over, evaluation.encode_masks(dm[torch.tensor(over, device=dm.device)], cap=self.rle_cap * 8)):
                rles[i] = r
        return rles

    def __call__(self, image, text, height=None, width=None, prompt="name"):
        """synchronous form: image -> (instances on the host, device record [k,6])"""
        return self.result(self.submit(image, text, height, width, prompt))
