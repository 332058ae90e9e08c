"""oracle/imageio.py -- CPU restatement of the byte/integer work on either side of the forward pass (TEST INFRASTRUCTURE:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it).

1. Input pipeline.  `ape/engine/defaults.py:213-222` applies the config's test augmentation
   (`ResizeShortestEdge(short_edge_length=1024, max_size=1024)`, configs/common/data/*:100-113) through detectron2's
   `ResizeTransform.apply_image`, which for uint8 images is `PIL.Image.fromarray(img).resize((new_w, new_h), BILINEAR)`.
   The arithmetic therefore lives in a third-party dependency that is NOT under /root/reference: **Pillow**
   (requirements: detectron2 -> Pillow>=7.1; this image has Pillow 12.2.0), `src/libImaging/Resample.c`:
   `precompute_coeffs`, `normalize_coeffs_8bpc`, `ImagingResampleHorizontal_8bpc`, `ImagingResampleVertical_8bpc`.
   Restated below from the published algorithm: a separable triangle filter whose support is scaled by the
   down-sampling factor, coefficients normalised in double precision, converted to 22-bit fixed point with
   round-half-away, a horizontal pass to uint8 and then a vertical pass to uint8, each accumulating from 1 << 21 and
   shifting right by 22 with a clip to [0, 255].  PINNED: tests/test_imageio.py compares it bit for bit with the
   installed Pillow over random sizes (up- and down-scaling), and tests/golden/imageio_golden.npz holds Pillow outputs.

2. Evaluator wire format.  `instances_to_coco_json` (detectron2, called from
   ape/evaluation/*_evaluation.py and demo/demo_lazy.py:189-198) turns every mask into COCO run-length encoding with
   pycocotools (`mask_util.encode(np.asfortranarray(mask))`, cocoapi `common/maskApi.c`: `rleEncode`, `rleToString`),
   also absent from /root/reference and not installed here (pycocotools 2.0.x).  Restated from the published algorithm:
   column-major runs starting with a run of zeros; the string form stores each count (from the third on as a difference
   to the count two back) in 5-bit groups, low group first, bit 0x20 = continuation, sign-extended, offset 48.
   **Parity unpinned** for the RLE (no pycocotools to run); anchored on the round trip decode(encode(m)) == m, on the
   format's invariants (counts sum to h*w, alternate 0/1 starting with 0) and on hand-checked vectors in the tests.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2          # Resample.c: coefficients of the 8 bit per channel path are 22-bit fixed point


def shortest_edge_size(h, w, short_edge_length, max_size):
    """detectron2 ResizeShortestEdge.get_output_shape (the augmentation configs/common/data/*:100-113 instantiate)"""
    scale = short_edge_length * 1.0 / min(h, w)
    if h < w:
        newh, neww = short_edge_length, scale * w
    else:
        newh, neww = scale * h, short_edge_length
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * scale, neww * scale
    return int(newh + 0.5), int(neww + 0.5)


def precompute_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle, support 1) filter over the whole
    axis (box = 0 .. in_size) -> (bounds [out, 2] = (first source index, count), kk [out, ksize] int32 fixed point)"""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.empty(xmax, np.float64)
        ww = 0.0
        for x in range(xmax):
            t = (x + xmin - center + 0.5) * ss
            if t < 0.0:
                t = -t
            w[x] = 1.0 - t if t < 1.0 else 0.0
            ww += w[x]
        for x in range(xmax):
            if ww != 0.0:
                w[x] /= ww
            v = w[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if w[x] < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk):
    """one separable pass along axis 0 of img [n, ...] uint8 -> [out, ...] uint8 (ImagingResample{Horizontal,Vertical}_8bpc)"""
    out = np.empty((bounds.shape[0],) + img.shape[1:], np.uint8)
    src = img.astype(np.int64)
    for i in range(bounds.shape[0]):
        lo, n = int(bounds[i, 0]), int(bounds[i, 1])
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(n):
            acc += src[lo + x] * int(kk[i, x])
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def resize_bilinear_u8(img, newh, neww):
    """HWC uint8 -> [newh, neww, C] uint8: Image.resize((neww, newh), BILINEAR).  Pillow resamples horizontally first
    (skipped when the width does not change), then vertically (skipped when the height does not change)."""
    h, w = img.shape[:2]
    out = img
    if neww != w:
        b, k = precompute_coeffs(w, neww)
        out = _pass(out.transpose(1, 0, 2), b, k).transpose(1, 0, 2)
    if newh != h:
        b, k = precompute_coeffs(h, newh)
        out = _pass(out, b, k)
    return np.ascontiguousarray(out)


def predictor_input(image_bgr, short_edge_length=1024, max_size=1024, input_format="RGB"):
    """ape/engine/defaults.py:213-222: BGR uint8 HWC -> the model's `image` input, float32 CHW"""
    img = image_bgr[:, :, ::-1] if input_format == "RGB" else image_bgr
    newh, neww = shortest_edge_size(img.shape[0], img.shape[1], short_edge_length, max_size)
    out = resize_bilinear_u8(np.ascontiguousarray(img), newh, neww)
    return out.astype(np.float32).transpose(2, 0, 1)


# ------------------------------------------------------------------------------------------------ COCO run-length encoding
def rle_encode(mask):
    """maskApi.c rleEncode: mask [h, w] of 0/1 -> list of run lengths in column-major order, first run counts zeros"""
    flat = np.asarray(mask, np.uint8).T.reshape(-1)         # column-major
    counts = []
    prev, run = 0, 0
    for v in flat:
        if v != prev:
            counts.append(run)
            run, prev = 0, v
        run += 1
    counts.append(run)
    return counts


def rle_to_string(counts):
    """maskApi.c rleToString: counts -> the ASCII string stored under "counts" in COCO json"""
    out = bytearray()
    for i, c in enumerate(counts):
        x = int(c)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            ch = x & 0x1F
            x >>= 5
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(ch + 48)
    return bytes(out)


def rle_from_string(s):
    """maskApi.c rleFrString"""
    counts = []
    p = 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = s[p] - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def rle_decode(counts, h, w):
    flat = np.zeros(h * w, np.uint8)
    pos, v = 0, 0
    for c in counts:
        flat[pos:pos + c] = v
        pos += c
        v ^= 1
    return flat.reshape(w, h).T
