"""hipGraph-replayed forward: the per-image launch sequence (~700 kernels) is captured once per (image size,
vocabulary size) and replayed, so the host does not pace the GPU.

The forward of DeformableDETRSegmVL.forward_single is a fixed sequence of launches with fixed shapes and no host
synchronisation (data-dependent selection / NMS / top-k are fixed-shape device code), which makes it capturable with
torch.cuda.CUDAGraph (a hipGraph on ROCm): our C-ABI launchers enqueue on torch's current stream, which is the
capturing stream inside `torch.cuda.graph`.  Results leave the device through pinned host buffers.
"""
from types import SimpleNamespace

import torch


class GraphedForward:
    def __init__(self, model_vision, use_graph=True, max_graphs=16, with_masks=True):
        self.mv = model_vision
        self.use_graph = use_graph
        self.max_graphs = max_graphs
        self.with_masks = with_masks
        self._graphs = {}

    def _device_part(self, image, text, height, width, prompt="name"):
        mv = self.mv
        h, w = image.shape[-2:]
        out = mv.forward_single(image, text, with_masks=self.with_masks, prompt=prompt)
        boxes = out["det_boxes"].clone()
        boxes[:, 0::2] = (boxes[:, 0::2] * (width / w)).clamp(0, width)
        boxes[:, 1::2] = (boxes[:, 1::2] * (height / h)).clamp(0, height)
        keep = (out["det_scores"] >= 0) & ((boxes[:, 2] - boxes[:, 0]) > 0) & ((boxes[:, 3] - boxes[:, 1]) > 0)
        rec = torch.cat([boxes, out["det_scores"][:, None], out["det_classes"][:, None].float(),
                         out["det_query"][:, None].float(), keep[:, None].float()], 1).contiguous()   # [k, 8]
        masks = None
        if "det_masks128" in out:
            from . import ops
            masks = ops.paste_bits(out["det_masks128"], boxes.contiguous(), height, width)
        return rec, masks, rec[:, :6].contiguous()

    def _build(self, image, text, height, width, prompt):
        mv = self.mv
        dev = image.device
        entry = SimpleNamespace()
        entry.image = image.clone()
        entry.text = text
        for _ in range(2):            # warm every cache (weight packing, geometry, text side) outside the capture
            self._device_part(entry.image, text, height, width, prompt)
        torch.cuda.synchronize()
        if self.use_graph:
            entry.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(entry.graph):
                entry.rec, entry.masks, entry.rec6 = self._device_part(entry.image, text, height, width, prompt)
        else:
            entry.graph = None
        k = mv.test_topk_per_image
        entry.h_rec = torch.empty((k, 8), dtype=torch.float32, pin_memory=True)
        entry.h_masks = torch.empty((k, height, width), dtype=torch.uint8, pin_memory=True) if self.with_masks and mv.test_mask_on else None
        return entry

    @torch.no_grad()
    def __call__(self, image, text, height=None, width=None, prompt="name"):
        """image [3,h,w] fp32 on the device, text [K, D] on the device -> (instances on the host, device record [k,6]).
        prompt: "name" (bank feeds the classifier only) or "phrase" / "expression" (bank fused in the encoder)."""
        h, w = image.shape[-2:]
        height, width = height or h, width or w
        key = (h, w, height, width, text.data_ptr(), tuple(text.shape), prompt)
        e = self._graphs.get(key)
        if e is None:
            if len(self._graphs) >= self.max_graphs:
                self._graphs.pop(next(iter(self._graphs)))
            e = self._graphs[key] = self._build(image, text, height, width, prompt)
        if e.graph is not None:
            e.image.copy_(image, non_blocking=True)
            e.graph.replay()
            rec, masks, rec6 = e.rec, e.masks, e.rec6
        else:
            rec, masks, rec6 = self._device_part(image, text, height, width, prompt)
        e.h_rec.copy_(rec, non_blocking=True)
        if masks is not None and e.h_masks is not None:
            e.h_masks.copy_(masks, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        keep = e.h_rec[:, 7] > 0.5
        inst = SimpleNamespace(image_size=(height, width), pred_boxes=e.h_rec[keep, :4].clone(), scores=e.h_rec[keep, 4].clone(),
                               pred_classes=e.h_rec[keep, 5].long(), query_index=e.h_rec[keep, 6].long())
        if masks is not None and e.h_masks is not None:
            # zero-copy view of the pinned staging buffer (valid until the next call with the same key)
            inst.pred_masks = e.h_masks.view(torch.bool) if bool(keep.all()) else e.h_masks[keep].view(torch.bool)
        return inst, rec6
