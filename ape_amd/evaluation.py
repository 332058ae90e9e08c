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
