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

`images_per_step = B > 1`: the forwards of B images are captured as PARALLEL BRANCHES of one graph (ops.fork), so the
latency-bound phases of one image (proposal selection, decoder, NMS: mostly idle CUs) run next to the GEMM-heavy phases
of another.  Every image still goes through the batch-1 pipeline; nothing is batched numerically.
"""
from types import SimpleNamespace

import torch


class GraphedForward:
    SLOTS = 2

    def __init__(self, model_vision, use_graph=True, max_graphs=16, with_masks=True, images_per_step=1, batch_vit=True):
        self.mv = model_vision
        self.batch_vit = batch_vit
        self.use_graph = use_graph
        self.max_graphs = max_graphs
        self.with_masks = with_masks
        self.B = int(images_per_step)
        self._graphs = {}
        self._copy_stream = None

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
        for _ in range(2):            # warm every cache (weight packing, geometry, text side) outside the capture
            self._device_all(entry.images, text, height, width, prompt)
        torch.cuda.synchronize()
        if self.use_graph:
            entry.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(entry.graph):
                entry.outs = self._device_all(entry.images, text, height, width, prompt)
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
        entry.size = (height, width)
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
                self._graphs.pop(next(iter(self._graphs)))
            e = self._graphs[key] = self._build(images, text, height, width, prompt)
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=images[0].device)
        s = e.slots[e.next_slot]
        if s.busy:
            raise RuntimeError("GraphedForward.submit: every result slot is outstanding -- call result() on an earlier ticket first")
        e.next_slot = (e.next_slot + 1) % self.SLOTS
        cur = torch.cuda.current_stream()
        if e.graph is not None:
            for buf, im in zip(e.images, images):
                buf.copy_(im, non_blocking=True)
            e.graph.replay()
            outs = e.outs
        else:
            outs = self._device_all(images, text, height, width, prompt)
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
        rec6 = s.d_rec[:, :, :6] if self.B > 1 else s.d_rec[0, :, :6]
        return SimpleNamespace(entry=e, slot=s, rec6=rec6, single=not isinstance(image, (list, tuple)))

    def result(self, ticket):
        """wait for the ticket's transfer; returns (instances on the host, device record view [k,6]) -- lists / [B,k,6] when the
        step was submitted as a list.  pred_masks is a zero-copy view of the slot's pinned buffer: valid until SLOTS further
        submits."""
        e, s = ticket.entry, ticket.slot
        s.copied.synchronize()
        s.busy = False
        height, width = e.size
        insts = []
        for b in range(self.B):
            hr = s.h_rec[b]
            n = int((hr[:, 7] > 0.5).sum())                 # kept detections are a prefix (sorted on the device)
            inst = SimpleNamespace(image_size=(height, width), pred_boxes=hr[:n, :4].clone(), scores=hr[:n, 4].clone(),
                                   pred_classes=hr[:n, 5].long(), query_index=hr[:n, 6].long())
            if s.has_masks:
                inst.pred_masks = s.h_masks[b, :n].view(torch.bool)      # zero-copy view of the pinned buffer
            insts.append(inst)
        if ticket.single:
            return insts[0], ticket.rec6
        return insts, ticket.rec6

    def __call__(self, image, text, height=None, width=None, prompt="name"):
        """synchronous form: image -> (instances on the host, device record [k,6])"""
        return self.result(self.submit(image, text, height, width, prompt))
