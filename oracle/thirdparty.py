"""CPU restatement of the THIRD-PARTY arithmetic the APE forward pass depends on.

TEST INFRASTRUCTURE (oracle).  Nothing under ape_amd/ may import this module.

The reference (shenyunhang/APE) calls into libraries that are not vendored under /root/reference and are
not installable in this image: detrex @776058e, detectron2 @017abbf (requirements.txt:10-11), torchvision
(requirements.txt:2, unpinned), timm (DropPath, identity in eval).  Their algorithms are restated here from
their published sources; parity for THESE functions is therefore "unpinned" (no reference test or source on
disk to check against) and is anchored on the reference's own call sites, cited per function.
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# detrex.utils.inverse_sigmoid (eps = 1e-3) -- call sites deformable_transformer_vl.py:237,
# deformable_detr_segm_vl.py:490
# ----------------------------------------------------------------------------------------------
def inverse_sigmoid(x, eps=1e-3):
    x = x.clamp(min=0, max=1)
    x1 = x.clamp(min=eps)
    x2 = (1 - x).clamp(min=eps)
    return torch.log(x1 / x2)


# detrex.layers.box_ops -- call sites deformable_transformer_vl.py:570, deformable_detr_segm_vl.py:789
def box_cxcywh_to_xyxy(x):
    x_c, y_c, w, h = x.unbind(-1)
    return torch.stack([x_c - 0.5 * w, y_c - 0.5 * h, x_c + 0.5 * w, y_c + 0.5 * h], dim=-1)


def box_xyxy_to_cxcywh(x):
    x0, y0, x1, y1 = x.unbind(-1)
    return torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, (x1 - x0), (y1 - y0)], dim=-1)


# ----------------------------------------------------------------------------------------------
# detrex.layers.PositionEmbeddingSine(num_pos_feats=128, temperature=1e4, normalize=True, offset=-0.5)
# config ape_deta_r50.py:35-40; call site deformable_detr_segm_vl.py:386-388
# ----------------------------------------------------------------------------------------------
def position_embedding_sine(mask, num_pos_feats=128, temperature=10000, normalize=True, offset=-0.5, eps=1e-6,
                            scale=2 * math.pi):
    """mask [B,H,W] bool (True = padded) -> [B, 2*num_pos_feats, H, W] float32"""
    not_mask = ~mask
    y_embed = not_mask.cumsum(1, dtype=torch.float32)
    x_embed = not_mask.cumsum(2, dtype=torch.float32)
    if normalize:
        y_embed = (y_embed + offset) / (y_embed[:, -1:, :] + eps) * scale
        x_embed = (x_embed + offset) / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32, device=mask.device)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_pos_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    B, H, W = mask.shape
    pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).view(B, H, W, -1)
    pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).view(B, H, W, -1)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


# ----------------------------------------------------------------------------------------------
# detectron2.layers.batch_norm.LayerNorm ("LN" of get_norm): channel LayerNorm on NCHW, eps 1e-6
# call sites vit_eva_clip.py:808,829-842
# ----------------------------------------------------------------------------------------------
def layer_norm_2d(x, weight, bias, eps=1e-6):
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return weight[:, None, None] * x + bias[:, None, None]


# detectron2 LastLevelMaxPool: F.max_pool2d(x, kernel_size=1, stride=2, padding=0) -- vit_eva_clip.py:907-912
def last_level_max_pool(x):
    return F.max_pool2d(x, kernel_size=1, stride=2, padding=0)


# ----------------------------------------------------------------------------------------------
# detectron2.structures.ImageList.from_tensors with padding_constraints {"square_size": S}
# (call site deformable_detr_segm_vl.py:850-854; vit_eva_clip.py:864-869: the key "size_divisiblity" is
# misspelt in the reference so only the square pad applies; Backbone.size_divisibility is the base-class 0)
# ----------------------------------------------------------------------------------------------
def pad_to_square(image, square_size, pad_value=0.0):
    """image [C,h,w] -> ([C,S,S], (h,w)); top-left placement"""
    h, w = image.shape[-2:]
    S = max(square_size, h, w) if square_size > 0 else None
    if S is None:
        return image, (h, w)
    return F.pad(image, (0, S - w, 0, S - h), value=pad_value), (h, w)


# ----------------------------------------------------------------------------------------------
# torchvision.ops.nms / batched_nms (per-category greedy NMS, result sorted by descending score)
# call sites: deformable_transformer_vl.py:592-597 (via torchvision), fast_rcnn.py:192 (via detectron2)
# ----------------------------------------------------------------------------------------------
def nms(boxes, scores, iou_threshold):
    """boxes [n,4] xyxy, scores [n] -> kept indices ordered by descending score (stable)"""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros((0,), dtype=torch.long)
    b = boxes.float()
    order = torch.sort(scores.float(), descending=True, stable=True)[1]
    bs = b[order]
    areas = (bs[:, 2] - bs[:, 0]) * (bs[:, 3] - bs[:, 1])
    suppressed = torch.zeros(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        if i + 1 >= n:
            break
        xx1 = torch.maximum(bs[i, 0], bs[i + 1:, 0])
        yy1 = torch.maximum(bs[i, 1], bs[i + 1:, 1])
        xx2 = torch.minimum(bs[i, 2], bs[i + 1:, 2])
        yy2 = torch.minimum(bs[i, 3], bs[i + 1:, 3])
        inter = (xx2 - xx1).clamp(min=0) * (yy2 - yy1).clamp(min=0)
        ovr = inter / (areas[i] + areas[i + 1:] - inter)
        suppressed[i + 1:] |= ovr > iou_threshold
    return order[torch.tensor(keep, dtype=torch.long)]


def batched_nms(boxes, scores, idxs, iou_threshold):
    """Per-category NMS ("vanilla" strategy of torchvision.ops.boxes.batched_nms); kept indices sorted by
    descending score.  Ties keep their index order (stable sort)."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    boxes = boxes.float()
    keep_mask = torch.zeros_like(scores, dtype=torch.bool)
    for class_id in torch.unique(idxs):
        curr = torch.where(idxs == class_id)[0]
        keep_mask[curr[nms(boxes[curr], scores[curr], iou_threshold)]] = True
    keep = torch.where(keep_mask)[0]
    return keep[torch.sort(scores[keep].float(), descending=True, stable=True)[1]]


# ----------------------------------------------------------------------------------------------
# torchvision.ops.roi_align(aligned=True, sampling_ratio=0) as used by
# detectron2 BitMasks.crop_and_resize(boxes, 128)  -- call site deformable_detr_segm_vl.py:606-608
# ----------------------------------------------------------------------------------------------
def _bilinear(img, y, x):
    """img [H,W]; y,x 1-D float tensors of equal length -> sampled values (roi_align's bilinear_interpolate)"""
    H, W = img.shape
    out_of_range = (y < -1.0) | (y > H) | (x < -1.0) | (x > W)
    y = y.clamp(min=0)
    x = x.clamp(min=0)
    y_low = y.floor().long()
    x_low = x.floor().long()
    yc = y_low >= H - 1
    xc = x_low >= W - 1
    y_high = torch.where(yc, torch.full_like(y_low, H - 1), y_low + 1)
    x_high = torch.where(xc, torch.full_like(x_low, W - 1), x_low + 1)
    y_low = torch.where(yc, torch.full_like(y_low, H - 1), y_low)
    x_low = torch.where(xc, torch.full_like(x_low, W - 1), x_low)
    y = torch.where(yc, y_low.to(y.dtype), y)
    x = torch.where(xc, x_low.to(x.dtype), x)
    ly, lx = y - y_low, x - x_low
    hy, hx = 1.0 - ly, 1.0 - lx
    v = (hy * hx * img[y_low, x_low] + hy * lx * img[y_low, x_high] + ly * hx * img[y_high, x_low] +
         ly * lx * img[y_high, x_high])
    return torch.where(out_of_range, torch.zeros_like(v), v)


def roi_align_aligned(img, box, out_size):
    """img [H,W] float, box xyxy (4,), -> [out_size,out_size]; spatial_scale 1, aligned=True, adaptive sampling"""
    x1, y1, x2, y2 = [float(v) for v in box]
    sw, sh = x1 - 0.5, y1 - 0.5
    rw, rh = (x2 - 0.5) - sw, (y2 - 0.5) - sh
    bw, bh = rw / out_size, rh / out_size
    gh = int(math.ceil(rh / out_size))
    gw = int(math.ceil(rw / out_size))
    count = max(gh * gw, 1)
    if gh <= 0 or gw <= 0:
        return torch.zeros((out_size, out_size), dtype=torch.float32)
    ph = torch.arange(out_size, dtype=torch.float32)
    iy = torch.arange(gh, dtype=torch.float32)
    ix = torch.arange(gw, dtype=torch.float32)
    ys = sh + ph[:, None] * bh + (iy[None, :] + 0.5) * bh / gh  # [P, gh]
    xs = sw + ph[:, None] * bw + (ix[None, :] + 0.5) * bw / gw  # [P, gw]
    Y = ys[:, None, :, None].expand(out_size, out_size, gh, gw).reshape(-1)
    X = xs[None, :, None, :].expand(out_size, out_size, gh, gw).reshape(-1)
    v = _bilinear(img.float(), Y, X).reshape(out_size, out_size, gh * gw)
    return v.sum(-1) / count


def bitmasks_crop_and_resize(bitmasks, boxes, mask_size):
    """detectron2 BitMasks.crop_and_resize: bitmasks [n,H,W] bool, boxes [n,4] -> [n,mask_size,mask_size] bool"""
    out = [roi_align_aligned(bitmasks[i].float(), boxes[i], mask_size) >= 0.5 for i in range(len(boxes))]
    return torch.stack(out) if out else torch.zeros((0, mask_size, mask_size), dtype=torch.bool)


# ----------------------------------------------------------------------------------------------
# detectron2.modeling.postprocessing.detector_postprocess + layers.mask_ops.paste_masks_in_image
# call site deformable_detr_segm_vl.py:869-871
# ----------------------------------------------------------------------------------------------
def paste_mask(mask, box, img_h, img_w, threshold=0.5):
    """mask [m,m] float in [0,1], box xyxy -> [img_h,img_w] bool  (full-image grid; the CPU path's skip_empty
    window only restricts WHERE the identical samples are evaluated)"""
    x0, y0, x1, y1 = [float(v) for v in box]
    img_y = (torch.arange(0, img_h, dtype=torch.float32) + 0.5 - y0) / (y1 - y0) * 2 - 1
    img_x = (torch.arange(0, img_w, dtype=torch.float32) + 0.5 - x0) / (x1 - x0) * 2 - 1
    gx = img_x[None, :].expand(img_h, img_w)
    gy = img_y[:, None].expand(img_h, img_w)
    grid = torch.stack([gx, gy], dim=2)[None]
    out = F.grid_sample(mask[None, None].float(), grid, align_corners=False)[0, 0]
    return out >= threshold


def sem_seg_postprocess(result, image_size, output_height, output_width):
    """detectron2.modeling.postprocessing.sem_seg_postprocess: crop the padding away, bilinear resize
    (align_corners=False) to the output resolution.  result [C, H, W]."""
    result = result[:, : image_size[0], : image_size[1]].expand(1, -1, -1, -1)
    return F.interpolate(result, size=(output_height, output_width), mode="bilinear", align_corners=False)[0]


def detector_postprocess(boxes, scores, classes, masks128, image_size, output_height, output_width, mask_threshold=0.5):
    """boxes [n,4] in the padded-input frame (image_size = (h,w) before padding); returns the rescaled / clipped /
    non-empty-filtered detections and the pasted masks [n', H, W] bool."""
    scale_x, scale_y = output_width / image_size[1], output_height / image_size[0]
    b = boxes.clone().float()
    b[:, 0::2] *= scale_x
    b[:, 1::2] *= scale_y
    b[:, 0::2] = b[:, 0::2].clamp(min=0, max=output_width)
    b[:, 1::2] = b[:, 1::2].clamp(min=0, max=output_height)
    keep = ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)
    b, scores, classes = b[keep], scores[keep], classes[keep]
    out_masks = None
    if masks128 is not None:
        m = masks128[keep]
        out_masks = torch.zeros((len(b), output_height, output_width), dtype=torch.bool)
        for i in range(len(b)):
            out_masks[i] = paste_mask(m[i].float(), b[i], output_height, output_width, mask_threshold)
    return b, scores, classes, out_masks, keep
