"""hipGraph-replayed, double-buffered forward: the per-image launch sequence (~700 kernels) is captured once per
(image size, vocabulary, prompt mode) and replayed, and the results of image i travel to the host on a copy stream
while image i+1 is being computed.

The forward of DeformableDETRSegmVL.forward_single is a fixed sequence of launches with fixed shapes and no host
synchronisation (data-dependent selection / NMS / top-k are fixed-shape device code), which makes it capturable with
torch.cuda.CUDAGraph (a hipGraph on ROCm): our C-ABI launchers enqueue on torch's current stream, which is the
capturing stream inside `torch.cuda.graph`.

Results leave the device through TWO slots (device staging + pinned host buffers).  The last kernel of the forward
(paste_bits, which writes the [k, H, W] instance masks: 105 MB for 100 detections at 1024^2) runs outside the graph and
writes straight into the slot's staging buffer; a copy stream then moves the slot to pinned memory behind an event.
`submit()` enqueues an image and returns a ticket, `result(ticket)` waits for that slot's copy only -- so the ~2 ms
PCIe transfer of one image overlaps the compute of the next.  `__call__` = result(submit(...)) is the synchronous form.

`pipeline=True` (software pipeline over steps): the graph of a step holds the ViT of the NEW images (one batched pass, every
linear at B x 4096 rows) as one branch and the rest of the forward ("tails": FPN, encoder, proposal selection, decoder,
heads, NMS, masks) of the PREVIOUS step's images as B more branches, reading the ViT features the previous replay left in
a persistent buffer.  The GEMM-bound ViT of one batch then overlaps the latency-bound tails of another instead of all
images sitting in the same phase at the same time; a ticket's results exist after the NEXT submit (or a flush).

`images_per_step = B > 1`: the forwards of B images are captured as PARALLEL BRANCHES of one graph (ops.fork), so the
latency-bound phases of one image (proposal selection, decoder, NMS: mostly idle CUs) run next to the GEMM-heavy phases
of another.  Every image still goes through the batch-1 pipeline; nothing is batched numerically.
"""
import os
import weakref
from types import SimpleNamespace

import torch

from .structures import make_instances


# Captured graphs are never destroyed while the process lives: on this stack (ROCm 7.2, torch 2.10) destroying a hipGraph
# whose capture used side streams, then capturing and launching further graphs, crashed hipGraphLaunch (reproduced:
# tests/test_model_gpu.py in suite order; keeping the graph objects alive removes it).  Evicted / orphaned graphs are parked
# here instead -- their private memory pools stay allocated, which is why GraphedForward serves every image size of a stream
# from few graphs instead of churning through them.
_RETIRED = []


class _Ticket:
    """handle of a submitted step (weak-referenceable, unlike SimpleNamespace)"""
    __slots__ = ("entry", "slot", "ready", "single", "rec6", "records", "__weakref__")

    def __init__(self, entry, single):
        self.entry, self.single = entry, single
        self.slot, self.ready, self.rec6, self.records = None, False, None, None


class GraphedForward:
    SLOTS = 2

    def __init__(self, model_vision, use_graph=True, max_graphs=16, with_masks=True, images_per_step=1, batch_vit=True,
                 pipeline=False):
        self.mv = model_vision
        self.batch_vit = batch_vit
        self.pipeline = bool(pipeline)
        self.use_graph = use_graph
        self.max_graphs = max_graphs
        self.with_masks = with_masks
        self.B = int(images_per_step)
        self._graphs = {}
        self._copy_stream = None

    def _retire(self, entry):
        if getattr(entry, "graph", None) is not None:
            _RETIRED.append(entry.graph)

    def __del__(self):
        try:
            for e in self._graphs.values():
                self._retire(e)
        except Exception:       # interpreter shutdown
            pass

    # ------------------------------------------------------------------ device work of one image
    def _device_part(self, image, text, height, width, prompt="name", vit_feat=None):
        """everything up to (excluding) the mask paste: (record [k,8], 128x128 masks or None, boxes in the output frame)"""
        mv = self.mv
        h, w = image.shape[-2:]
        out = mv.forward_single(image, text, with_masks=self.with_masks, prompt=prompt, vit_feat=vit_feat)
        boxes = out["det_boxes"].clone()
        boxes[:, 0::2] = (boxes[:, 0::2] * (width / w)).clamp(0, width)
        boxes[:, 1::2] = (boxes[:, 1::2] * (height / h)).clamp(0, height)
        keep = (out["det_scores"] >= 0) & ((boxes[:, 2] - boxes[:, 0]) > 0) & ((boxes[:, 3] - boxes[:, 1]) > 0)
        # dropped rows (empty slots, empty boxes after the rescale) carry score -1 in the record, so the 6-column view that
        # is all-gathered across ranks tells kept from dropped without the keep column
        score = torch.where(keep, out["det_scores"], torch.full_like(out["det_scores"], -1.0))
        rec = torch.cat([boxes, score[:, None], out["det_classes"][:, None].float(),
                         out["det_query"][:, None].float(), keep[:, None].float()], 1)                 # [k, 8]
        # kept detections first (stable): the host then takes PREFIX views of the pinned buffers instead of gathering
        # ~1 MB per mask with a boolean index (a 105 MB host copy per image whenever one detection is dropped)
        order = torch.sort((~keep).to(torch.int8), stable=True)[1]
        masks128 = out.get("det_masks128")
        return rec[order].contiguous(), (masks128[order].contiguous() if masks128 is not None else None), boxes[order].contiguous()

    def _device_all(self, images, text, height, width, prompt):
        """the B forwards: image 0 on the current stream, the others as forked branches"""
        from . import ops
        mv = self.mv
        if prompt == "expression" and mv.test_topk_per_image != 1:      # (:183-194) forward() applies the same rule
            saved = mv.test_topk_per_image
            mv.test_topk_per_image = 1
            try:
                return self._device_all(images, text, height, width, prompt)
            finally:
                mv.test_topk_per_image = saved
        if len(images) == 1:
            return [self._device_part(images[0], text, height, width, prompt)]
        # The ViT runs ONCE over the B images (every linear sees B x 4096 rows: the 256 x 256-tile GEMM kernels need that
        # many to fill the chip; rows are independent, so nothing changes numerically).  Everything after it stays one
        # batch-1 forward per image, the B of them parallel branches of the graph; the finer-grained forks inside a forward
        # run inline (nested fork/join made hipStreamEndCapture crash on this ROCm, and the image-level overlap already
        # fills the idle phases).
        with ops.inline_forks():
            feats = [None] * len(images)
            if self.batch_vit:
                n_tok = (mv.backbone.net.img_size // mv.backbone.net.patch_size) ** 2
                x = mv.backbone.net.forward_tokens(images, mv._mean, mv._std)
                feats = [x[b * n_tok:(b + 1) * n_tok] for b in range(len(images))]
            jobs = [ops.fork(lambda b=b: self._device_part(images[b], text, height, width, prompt, feats[b]), force=True)
                    for b in range(1, len(images))]
            outs = [self._device_part(images[0], text, height, width, prompt, feats[0])]
            return outs + [j.join() for j in jobs]

    def _device_pipelined(self, entry, text, height, width, prompt):
        """one pipelined step: ViT of entry.images (new) || tails of the features in entry.feat (previous step's images)"""
        from . import ops
        mv = self.mv
        net = mv.backbone.net
        B = len(entry.images)
        n_tok = entry.feat.shape[0] // B
        if prompt == "expression" and mv.test_topk_per_image != 1:
            saved = mv.test_topk_per_image
            mv.test_topk_per_image = 1
            try:
                return self._device_pipelined(entry, text, height, width, prompt)
            finally:
                mv.test_topk_per_image = saved
        with ops.inline_forks():
            vjob = ops.fork(lambda: net.forward_tokens(entry.images if B > 1 else entry.images[0], mv._mean, mv._std), force=True)
            feats = [entry.feat[b * n_tok:(b + 1) * n_tok] for b in range(B)]
            jobs = [ops.fork(lambda b=b: self._device_part(entry.images[b], text, height, width, prompt, feats[b]), force=True)
                    for b in range(1, B)]
            outs = [self._device_part(entry.images[0], text, height, width, prompt, feats[0])] + [j.join() for j in jobs]
            entry.feat.copy_(vjob.join())             # behind every tail: the next replay reads the new features
        return outs

    def _run_entry(self, e):
        """the device work of one step of entry `e` on its static image buffers"""
        height, width = e.size
        if self.pipeline:
            return self._device_pipelined(e, e.text, height, width, e.prompt)
        return self._device_all(e.images, e.text, height, width, e.prompt)

    def _build(self, images, text, height, width, prompt):
        mv = self.mv
        dev = images[0].device
        B = len(images)
        entry = SimpleNamespace()
        entry.images = [im.clone() for im in images]
        entry.text = text             # keeps the bank alive: the graph key holds its address
        # warm-up and capture run fusion_tokens, which (phrase / expression prompts, persistent bank) shifts the phrase bank
        # in place: snapshot it and restore it afterwards so that building a graph does not count as three extra images
        bank = mv.features_phrase_bank.clone() if getattr(mv, "text_feature_bank", False) else None
        if self.pipeline:
            net = mv.backbone.net
            n_tok = (net.img_size // net.patch_size) ** 2
            entry.feat = torch.zeros((B * n_tok, net.embed_dim), dtype=mv.compute_dtype, device=dev)
            entry.pending = None          # ticket whose ViT features sit in entry.feat, tails not run yet
        entry.size, entry.prompt = (height, width), prompt
        # (no closure over `entry` is stored on it: a reference cycle would leave the captured graph to the garbage collector,
        # which may then destroy it -- and free its memory pool -- in the middle of a later capture)
        run = lambda: self._run_entry(entry)  # noqa: E731
        for _ in range(2):            # warm every cache (weight packing, geometry, text side) outside the capture
            run()
        torch.cuda.synchronize()
        if self.use_graph:
            entry.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(entry.graph):
                entry.outs = run()
        else:
            entry.graph = None
        if bank is not None:
            mv.features_phrase_bank.copy_(bank)
        k = 1 if prompt == "expression" else mv.test_topk_per_image       # (:183-194) one box per referring expression
        has_masks = self.with_masks and mv.test_mask_on
        entry.slots = []
        for _ in range(self.SLOTS):
            s = SimpleNamespace()
            s.d_rec = torch.empty((B, k, 8), dtype=torch.float32, device=dev)
            s.h_rec = torch.empty((B, k, 8), dtype=torch.float32, pin_memory=True)
            s.d_masks = torch.empty((B, k, height, width), dtype=torch.uint8, device=dev) if has_masks else None
            s.h_masks = torch.empty((B, k, height, width), dtype=torch.uint8, pin_memory=True) if has_masks else None
            s.computed, s.copied = torch.cuda.Event(), torch.cuda.Event()
            s.busy = False
            entry.slots.append(s)
        entry.next_slot = 0
        return entry

    # ------------------------------------------------------------------ pipelined interface
    @torch.no_grad()
    def submit(self, image, text, height=None, width=None, prompt="name"):
        """enqueue one image [3,h,w] (fp32, device) -- or a list of `images_per_step` images of one size -- against the text
        bank [K, D] (device); returns a ticket.  At most SLOTS tickets may be outstanding per (size, vocabulary) entry."""
        from . import ops
        images = list(image) if isinstance(image, (list, tuple)) else [image]
        if len(images) != self.B:
            raise ValueError(f"GraphedForward.submit: expected {self.B} image(s) per step, got {len(images)}")
        h, w = images[0].shape[-2:]
        if any(tuple(im.shape[-2:]) != (h, w) for im in images):
            raise ValueError("GraphedForward.submit: the images of one step must share a size")
        height, width = height or h, width or w
        # the classifier's text side is a per-vocabulary constant baked into the capture: an in-place update of the bank
        # (text._version) must rebuild the graph, exactly like a new bank
        key = (h, w, height, width, text.data_ptr(), text._version, tuple(text.shape), prompt)
        e = self._graphs.get(key)
        if e is None:
            if len(self._graphs) >= self.max_graphs:
                oldest = next(iter(self._graphs))
                self.flush(self._graphs[oldest])       # a ticket waiting for its tails keeps its entry alive through ticket.entry
                self._retire(self._graphs.pop(oldest))
            e = self._graphs[key] = self._build(images, text, height, width, prompt)
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=images[0].device)
        single = not isinstance(image, (list, tuple))
        if not self.pipeline:
            t = _Ticket(e, single)
            self._replay(e, images, text, height, width, prompt, completes=t)
            return t
        # pipelined: this replay runs the tails of the PREVIOUS ticket and the ViT of these images
        t = _Ticket(e, single)
        self._replay(e, images, text, height, width, prompt, completes=e.pending() if e.pending is not None else None)
        e.pending = weakref.ref(t)        # weak: ticket -> entry -> ticket would be a cycle (see _build); a dropped ticket's
        return t                          # detections are simply not delivered

    def _replay(self, e, images, text, height, width, prompt, completes):
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
        if images is not None:
            for buf, im in zip(e.images, images):
                buf.copy_(im, non_blocking=True)
        if e.graph is not None:
            e.graph.replay()
            outs = e.outs
        else:
            outs = self._run_entry(e)
        if completes is None:
            return
        cur.wait_event(s.copied)                      # the slot's previous transfer has left the staging buffers
        has_masks = s.d_masks is not None and outs[0][1] is not None
        for b, (rec, masks128, boxes) in enumerate(outs):
            s.d_rec[b].copy_(rec, non_blocking=True)
            if has_masks:
                ops.paste_bits(masks128, boxes, height, width, out=s.d_masks[b])     # detector_postprocess (:869-871), into the slot
        s.computed.record(cur)
        with torch.cuda.stream(self._copy_stream):
            self._copy_stream.wait_event(s.computed)
            s.h_rec.copy_(s.d_rec, non_blocking=True)
            if has_masks:
                s.h_masks.copy_(s.d_masks, non_blocking=True)
            s.copied.record(self._copy_stream)
        s.busy = True
        s.has_masks = has_masks
        completes.slot, completes.ready = s, True
        completes.rec6 = s.d_rec[:, :, :6] if self.B > 1 else s.d_rec[0, :, :6]

    def flush(self, entry=None):
        """pipelined mode: run the tails of the ticket whose ViT features are waiting (one more replay; its ViT branch recomputes
        the features of the images already in the static buffers, which nobody consumes)"""
        for e in ([entry] if entry is not None else list(self._graphs.values())):
            if self.pipeline and e.pending is not None:
                height, width = e.size
                t, e.pending = e.pending(), None
                self._replay(e, None, e.text, height, width, e.prompt, completes=t)

    def result(self, ticket):
        """wait for the ticket's transfer; returns (instances on the host, device record view [k,6]) -- lists / [B,k,6] when the
        step was submitted as a list.  pred_masks is a zero-copy view of the slot's pinned buffer: valid until SLOTS further
        completed steps.  Pipelined mode: a ticket completes with the NEXT submit; asking earlier flushes the pipeline."""
        e = ticket.entry
        if not ticket.ready:
            self.flush(e)
        s = ticket.slot
        s.copied.synchronize()
        s.busy = False
        height, width = e.size
        insts = []
        for b in range(self.B):
            hr = s.h_rec[b]
            n = int((hr[:, 7] > 0.5).sum())                 # kept detections are a prefix (sorted on the device)
            inst = make_instances((height, width), hr[:n, :4].clone(), hr[:n, 4].clone(), hr[:n, 5].long(),
                                  s.h_masks[b, :n].view(torch.bool) if s.has_masks else None,    # zero-copy view of the pinned buffer
                                  query_index=hr[:n, 6].long())
            insts.append(inst)
        if ticket.single:
            return insts[0], ticket.rec6
        return insts, ticket.rec6

    def __call__(self, image, text, height=None, width=None, prompt="name"):
        """synchronous form: image -> (instances on the host, device record [k,6])"""
        return self.result(self.submit(image, text, height, width, prompt))
