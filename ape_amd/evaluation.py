"""Evaluator wire format: `instances_to_coco_json`, the function every evaluator of the reference and its
demo call on a model's output (ape/evaluation/d3_evaluation.py:441-493, refcoco_evaluation.py:425-477, the detectron2
original used by lvis_evaluation.py:96 / multi_dataset_evaluator.py:173 / demo/demo_lazy.py:189-198): one dict per
detection with `image_id`, `category_id`, `bbox` (XYWH), `score` and `segmentation` = COCO RLE
`{"size": [h, w], "counts": str}`.

The reference encodes each full-resolution mask on the host with pycocotools after a ~1 MB-per-mask device-to-host copy.
Here the run lengths are produced on the device (`ape_hip_rle_encode`, csrc/imageio.hip) and only they travel; the string
form is the library's host function `ape_hip_rle_to_string`.  Three sources, in this order:
  * `instances.pred_masks_rle` -- already encoded by the runtime (`GraphedForward(mask_format="rle")`);
  * `instances.pred_masks` on the device -- encoded here;
  * `instances.pred_masks` on the host (the reference's output contract) -- uploaded, then encoded (there is no host
    encoder in this package).
"""
import torch

from . import ops


def encode_masks(masks, cap=4096):
    """[n, H, W] bool / uint8 masks (device, or host -> uploaded to the current device) -> list of COCO RLE dicts with
    `counts` as bytes (what pycocotools' encode returns)"""
    n = masks.shape[0]
    if n == 0:
        return []
    if not masks.is_cuda:
        masks = masks.to("cuda", non_blocking=False)
    masks = masks.contiguous()
    H, W = masks.shape[1:]
    out = [None] * n
    todo = list(range(n))
    while todo:
        sub = masks if len(todo) == n else masks[torch.tensor(todo, device=masks.device)]
        counts, nruns = ops.rle_encode(sub, cap=cap)
        nr = nruns.cpu()
        hc = counts[:, : int(min(cap, int(nr.max())))].cpu()
        again = []
        for j, i in enumerate(todo):
            r = int(nr[j])
            if r > cap:
                again.append(i)                                   # truncated: a mask with more runs than the buffer holds
                continue
            out[i] = {"size": [H, W], "counts": ops.rle_to_string(hc[j, :r])}
        todo = again
        cap *= 8
    return out


def rles_from_runs(counts, nruns, size):
    """host run-length arrays ([n, cap] int32, [n]) as produced by the runtime -> list of RLE dicts (counts as bytes)"""
    return [{"size": [int(size[0]), int(size[1])], "counts": ops.rle_to_string(counts[i, : int(nruns[i])])}
            for i in range(counts.shape[0])]


def instances_to_coco_json(instances, img_id):
    """d3_evaluation.py:441-493 with the mask encoding on the device"""
    num_instance = len(instances)
    if num_instance == 0:
        return []
    b = instances.pred_boxes.tensor.detach().cpu().clone()
    b[:, 2] -= b[:, 0]                       # BoxMode.XYXY_ABS -> XYWH_ABS
    b[:, 3] -= b[:, 1]
    boxes = b.tolist()
    scores = instances.scores.tolist()
    classes = instances.pred_classes.tolist()
    rles = None
    if instances.has("pred_masks_rle"):
        rles = [dict(r) for r in instances.pred_masks_rle]
    elif instances.has("pred_masks"):
        rles = encode_masks(instances.pred_masks)
    if rles is not None:
        for r in rles:                       # json cannot hold bytes; utf-8 is what pycocotools' _mask.pyx does too (:471-475)
            if isinstance(r["counts"], bytes):
                r["counts"] = r["counts"].decode("utf-8")
    results = []
    for k in range(num_instance):
        result = {"image_id": img_id, "category_id": classes[k], "bbox": boxes[k], "score": scores[k]}
        if rles is not None:
            result["segmentation"] = rles[k]
        results.append(result)
    return results


def _iou_matrix(a, b):
    """pairwise IoU of XYXY boxes a [n, 4], b [m, 4] (float64)"""
    a, b = a.double(), b.double()
    area_a = (a[:, 2] - a[:, 0]).clamp_min(0) * (a[:, 3] - a[:, 1]).clamp_min(0)
    area_b = (b[:, 2] - b[:, 0]).clamp_min(0) * (b[:, 3] - b[:, 1]).clamp_min(0)
    lt = torch.maximum(a[:, None, :2], b[None, :, :2])
    rb = torch.minimum(a[:, None, 2:], b[None, :, 2:])
    inter = (rb - lt).clamp_min(0).prod(-1)
    return inter / (area_a[:, None] + area_b[None, :] - inter).clamp_min(1e-300)


def box_ap(detections, ground_truth, iou_thresholds=None, max_dets=100):
    """COCO-style box AP of `detections` against `ground_truth` (the protocol of pycocotools' COCOeval that the reference's
    evaluators run -- ape/evaluation/lvis_evaluation.py / detectron2's COCOEvaluator -- restated: area range "all", no crowd
    regions, at most `max_dets` detections per (image, CATEGORY) -- COCOeval applies maxDets inside its per-category evaluateImg, so a
    model top-k of 300 / 500 loses nothing to the default 100 --, IoU thresholds 0.50:0.05:0.95, greedy matching in score order to the
    unmatched ground truth of the same class with the highest IoU, precision made monotone and sampled at 101 recall points,
    averaged over thresholds and over the classes that have ground truth).  pycocotools is not installable here: unpinned.

    detections: list (one entry per image) of (boxes [n, 4] XYXY, scores [n], classes [n]);
    ground_truth: list of (boxes [m, 4] XYXY, classes [m]).  -> dict(AP, AP50, AP75, classes, gt, dets)."""
    if iou_thresholds is None:
        iou_thresholds = [0.5 + 0.05 * i for i in range(10)]
    T = len(iou_thresholds)
    thr = torch.tensor(iou_thresholds, dtype=torch.float64)
    per_class = {}                      # class -> list of (scores [n], matched [T, n]) per image, and the GT count
    n_gt = {}
    for (db, ds, dc), (gb, gc) in zip(detections, ground_truth):
        db, ds, dc = db.detach().cpu().double(), ds.detach().cpu().double(), dc.detach().cpu().long()
        gb, gc = gb.detach().cpu().double(), gc.detach().cpu().long()
        order = torch.argsort(ds, descending=True, stable=True)
        db, ds, dc = db[order], ds[order], dc[order]
        for c in set(dc.tolist()) | set(gc.tolist()):
            d_idx, g_idx = (dc == c).nonzero().flatten()[:max_dets], (gc == c).nonzero().flatten()
            n_gt[c] = n_gt.get(c, 0) + int(g_idx.numel())
            if d_idx.numel() == 0:
                continue
            matched = torch.zeros((T, d_idx.numel()), dtype=torch.bool)
            if g_idx.numel() > 0:
                iou = _iou_matrix(db[d_idx], gb[g_idx])
                for t in range(T):
                    taken = torch.zeros(g_idx.numel(), dtype=torch.bool)
                    for i in range(d_idx.numel()):                       # detections are in descending score order
                        cand = iou[i].clone()
                        cand[taken] = -1.0
                        j = int(cand.argmax())
                        if cand[j] >= min(float(thr[t]), 1 - 1e-10):
                            taken[j] = True
                            matched[t, i] = True
            per_class.setdefault(c, []).append((ds[d_idx], matched))
    rec_thrs = torch.linspace(0.0, 1.0, 101, dtype=torch.float64)
    aps = []
    for c, total in n_gt.items():
        if total == 0:
            continue
        entries = per_class.get(c, [])
        if not entries:
            aps.append(torch.zeros(T, dtype=torch.float64))
            continue
        scores = torch.cat([e[0] for e in entries])
        matched = torch.cat([e[1] for e in entries], dim=1)
        order = torch.argsort(scores, descending=True, stable=True)
        tp = matched[:, order].double().cumsum(1)
        fp = (~matched[:, order]).double().cumsum(1)
        recall = tp / total
        precision = tp / (tp + fp).clamp_min(1e-300)
        ap_t = torch.zeros(T, dtype=torch.float64)
        for t in range(T):
            pr = torch.flip(torch.cummax(torch.flip(precision[t], [0]), 0)[0], [0])        # monotone non-increasing envelope
            idx = torch.searchsorted(recall[t].contiguous(), rec_thrs, right=False)
            q = torch.where(idx < pr.numel(), pr[idx.clamp_max(pr.numel() - 1)], torch.zeros_like(rec_thrs))
            ap_t[t] = q.mean()
        aps.append(ap_t)
    if not aps:
        return {"AP": float("nan"), "AP50": float("nan"), "AP75": float("nan"), "classes": 0, "gt": 0, "dets": 0}
    A = torch.stack(aps)                                                                    # [classes, T]
    pick = lambda v: float(A[:, min(range(T), key=lambda i: abs(iou_thresholds[i] - v))].mean())   # noqa: E731
    return {"AP": float(A.mean()), "AP50": pick(0.5), "AP75": pick(0.75), "classes": int(A.shape[0]),
            "gt": int(sum(n_gt.values())), "dets": int(sum(e[0].numel() for es in per_class.values() for e in es))}
