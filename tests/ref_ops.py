"""Plain-PyTorch fp32 definitions of every op in ape_amd.ops (same signatures).

TEST INFRASTRUCTURE ONLY.  Two uses:
  * `-m gpu` parity tests compare each HIP kernel with the function of the same name here;
  * `-m "not gpu"` tests monkeypatch `ape_amd.ops` with this module so the host-side composition of the
    model can be checked against the oracle on CPU (tests/conftest.py: `fake_ops`).
Math is done in float32 on whatever device the inputs live on; outputs are rounded to the requested dtype
exactly once, like the kernels.
"""

import math

import torch
import torch.nn.functional as F

ACT_NONE, ACT_RELU, ACT_GELU, ACT_SWIGLU, ACT_SILU = 0, 1, 2, 3, 4
MASK_NONE, MASK_ZERO_INPUT, MASK_ZERO_OUTPUT = 0, 1, 2
DT_F32, DT_BF16 = 0, 1


def _act(x, act):
    if act == ACT_RELU:
        return F.relu(x)
    if act == ACT_GELU:
        return F.gelu(x)
    if act == ACT_SILU:
        return F.silu(x)
    return x


def _rotate_half(x):
    x1, x2 = x[..., 0::2], x[..., 1::2]
    return torch.stack((-x2, x1), dim=-1).flatten(-2)


def gemm(a, w, bias=None, *, out=None, out_dtype=None, residual=None, act=ACT_NONE, alpha=1.0, clamp=0.0,
         rowmask=None, mask_mode=MASK_NONE, trans_out=False, rope=None, m_pad=None, splitk=None, tile64=None, rownorm=None, norm=None,
         rowstats=None):
    M, K = a.shape
    N = w.shape[0]
    x = (a.float() @ w.float().t()) * alpha
    if rowstats is not None:                     # the folded LayerNorm's row terms from the operand itself (ApeGemmArgs.rowstat_cols)
        cols, eps, cv = rowstats
        st = row_stats(a[:, :cols], eps)
        rownorm = (st[0], st[1], cv)
    if rownorm is not None:
        rs, sh, cv = rownorm
        x = x * rs.float()[:, None] + sh.float()[:, None] * cv.float()[None, :]
    rm = rowmask.bool() if rowmask is not None else None
    if rm is not None and mask_mode == MASK_ZERO_INPUT:
        x = x.masked_fill(rm[:, None], 0.0)
    if bias is not None:
        x = x + bias.float()
    if rope is not None:
        cos, sin, rows, hd, cols = rope[:5]          # a sixth entry (packed pairs) is the same table in another layout
        idx = torch.arange(M, device=a.device) % rows
        c, s = cos.float()[idx], sin.float()[idx]  # [M, hd]
        t = x[:, :cols].reshape(M, cols // hd, hd)
        t = t * c[:, None, :] + _rotate_half(t) * s[:, None, :]
        x = torch.cat([t.reshape(M, cols), x[:, cols:]], dim=1)
    if act == ACT_SWIGLU:
        x = F.silu(x[:, 0::2]) * x[:, 1::2]
    else:
        x = _act(x, act)
    if clamp > 0:
        x = x.clamp(-clamp, clamp)
    if residual is not None:
        x = x + residual.float()
    if rm is not None and mask_mode == MASK_ZERO_OUTPUT:
        x = x.masked_fill(rm[:, None], 0.0)
    if norm is not None:            # LayerNorm of the fp32 sums (the kernel's epilogue), one rounding of the result
        x = F.layer_norm(x, (x.shape[1],), norm[0].float(), norm[1].float(), norm[2])
    odt = out.dtype if out is not None else (out_dtype or a.dtype)
    if trans_out:
        ld = m_pad or M
        res = torch.zeros((N, ld), dtype=odt, device=a.device)
        res[:, :M] = x.t().to(odt)
        if out is not None:
            out[:, :M] = res[:, :M]
            return out
        return res
    if out is not None:
        out[:, : x.shape[1]] = x.to(odt)
        return out
    return x.to(odt)


def gemm_norm_fusable(a, w, residual=None, out_dtype=None):
    return (a.dtype in (torch.bfloat16, torch.float16) and a.shape[1] == 256 and w.shape[0] == 256 and a.shape[0] >= 2048
            and (out_dtype or a.dtype) == a.dtype and (residual is None or residual.dtype == a.dtype))


def head_gemv(x, w, bias=None, alpha=1.0, *, bf16_copy=False):
    out = torch.einsum("hd,hnd->hn", x.float(), w.float()) * alpha
    out = out + bias.float().reshape(out.shape) if bias is not None else out
    return (out, out.to(torch.bfloat16 if bf16_copy is True else bf16_copy)) if bf16_copy else out


def row_stats(x, eps):
    xf = x.float()
    mean = xf.mean(dim=1)
    rstd = torch.rsqrt(xf.var(dim=1, unbiased=False) + eps)
    return rstd, -mean * rstd


def gemv(x, w, bias=None, alpha=1.0, *, scale=None, add=None):
    y = (x.float() @ w.float().t()) * alpha
    if bias is not None:
        y = y + bias.float()
    if scale is not None:
        y = y * scale.float()
    return y if add is None else (y, add.float() + y)


def layernorm(x, w, b, eps, *, out=None, out_dtype=None, act=ACT_NONE, cpad=None, add=None, out2=None):
    M, C = x.shape
    cpad = cpad or C
    y = _act(F.layer_norm(x.float(), (C,), w.float(), b.float(), eps), act)
    odt = out.dtype if out is not None else (out_dtype or x.dtype)
    yp = torch.zeros((M, cpad), dtype=torch.float32, device=x.device)
    yp[:, :C] = y
    if out is None:
        out = yp.to(odt)
    else:
        out[:, :cpad] = yp.to(odt)
    if add is None:
        return out
    y2 = torch.zeros((M, cpad), dtype=torch.float32, device=x.device)
    y2[:, :C] = y + add.float()[:, :C]
    if out2 is None:
        out2 = y2.to(odt)
    else:
        out2[:, :cpad] = y2.to(odt)
    return out, out2


def postnorm_residual(stream, t, norm, copy_dtype=None):
    if t is not None:
        stream += F.layer_norm(t.float(), (t.shape[1],), norm[0].float(), norm[1].float(), norm[2])
    return stream.to(copy_dtype) if copy_dtype is not None else None


def groupnorm(x, w, b, groups, eps, *, act=ACT_NONE, add=None, out=None, out_dtype=None):
    HW, C = x.shape
    y = F.group_norm(x.float().t().reshape(1, C, HW), groups, w.float(), b.float(), eps).reshape(C, HW).t()
    if add is not None:
        y = y + add.float()
    y = _act(y, act)
    odt = out.dtype if out is not None else (out_dtype or x.dtype)
    if out is not None:
        out.copy_(y.to(odt))
        return out
    return y.to(odt).contiguous()


def _to_list(v):
    return v.detach().cpu().tolist() if torch.is_tensor(v) else list(v)


def ms_deform_attn_core(value, spatial_shapes, sampling_locations, attention_weights):
    """value [B,S,M,D], sampling_locations [B,Q,M,L,P,2], attention_weights [B,Q,M,L,P] -> [B,Q,M*D]
    (grid_sample formulation, align_corners=False, zero padding)."""
    B, S, M, D = value.shape
    _, Q, _, L, P, _ = sampling_locations.shape
    shapes = [(int(h), int(w)) for h, w in _to_list(spatial_shapes)]
    vals = value.float().split([h * w for h, w in shapes], dim=1)
    grids = 2 * sampling_locations.float() - 1
    sampled = []
    for lvl, (h, w) in enumerate(shapes):
        v = vals[lvl].flatten(2).transpose(1, 2).reshape(B * M, D, h, w)
        g = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)
        sampled.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    aw = attention_weights.float().transpose(1, 2).reshape(B * M, 1, Q, L * P)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * aw).sum(-1).view(B, M * D, Q)
    return out.transpose(1, 2).contiguous()


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step=64):
    return ms_deform_attn_core(value, spatial_shapes, sampling_loc, attn_weight).to(value.dtype)


def msda_locations(offw, ref, spatial_shapes):
    """offw [Q, 8*L*4*3] (offsets then logits), ref [Q, L, 2|4] -> (loc [Q,8,L,4,2], weights [Q,8,L,4])."""
    shapes = [(int(h), int(w)) for h, w in _to_list(spatial_shapes)]
    L = len(shapes)
    Q = offw.shape[0]
    off = offw[:, : 8 * L * 4 * 2].float().reshape(Q, 8, L, 4, 2)
    logit = offw[:, 8 * L * 4 * 2:].float().reshape(Q, 8, L * 4)
    aw = logit.softmax(-1).reshape(Q, 8, L, 4)
    ref = ref.float()
    if ref.shape[-1] == 2:
        norm = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32, device=offw.device)
        loc = ref[:, None, :, None, :] + off / norm[None, None, :, None, :]
    else:
        loc = ref[:, None, :, None, :2] + off / 4 * ref[:, None, :, None, 2:] * 0.5
    return loc, aw


def msda_fused(value, spatial_shapes, level_start_index, offw, ref, *, batch=1, out_dtype=None, out=None):
    S = value.shape[0] // batch
    Q = offw.shape[0] // batch
    loc, aw = msda_locations(offw.float(), ref.reshape(batch * Q, -1, ref.shape[-1]), spatial_shapes)
    v = value[:, :256].float().reshape(batch, S, 8, 32)
    o = ms_deform_attn_core(v, spatial_shapes, loc.reshape(batch, Q, *loc.shape[1:]), aw.reshape(batch, Q, *aw.shape[1:]))
    o = o.reshape(batch * Q, 256)
    odt = out.dtype if out is not None else (out_dtype or value.dtype)
    if out is not None:
        out.copy_(o.to(odt))
        return out
    return o.to(odt)


def embed_tokens(tokens, table, pos, length, stride):
    B, W = tokens.shape[0], table.shape[1]
    out = torch.zeros((B, stride, W), dtype=torch.float32, device=tokens.device)
    out[:, :length] = table[tokens[:, :length].long()].float() + pos[:length].float()[None]
    return out.reshape(B * stride, W)


def attention(q, k, vt, *, batch, n, heads, head_dim, scale, out=None, stride=None, causal=False, v_head_dim=None):
    E = heads * head_dim
    hdv = head_dim if v_head_dim is None else v_head_dim
    stride = n if stride is None else stride
    rows = batch * stride

    def win(t):                                     # [rows, E] -> [batch, heads, n, head_dim] (window b = rows b*stride .. +n)
        tf = t[:, :E].float()
        if tf.shape[0] < rows:
            tf = torch.cat([tf, tf.new_zeros((rows - tf.shape[0], E))], 0)
        return tf[:rows].reshape(batch, stride, heads, head_dim)[:, :n].permute(0, 2, 1, 3)

    qf, kf = win(q), win(k)
    vf = vt[:, :rows].float().reshape(heads, hdv, batch, stride)[..., :n].permute(2, 0, 3, 1)
    att = (qf @ kf.transpose(-1, -2)) * scale
    if causal:
        att = att + torch.full((n, n), float("-inf"), device=att.device).triu_(1)
    if q.dtype in (torch.bfloat16, torch.float16) and not causal:
        # the kernel's rounding points (csrc/attention.hip): a flash loop over key tiles of 64 -- the tile's probabilities 2^(s c - m) are
        # rounded to the operands' 16-bit type relative to the running maximum m, the row sum runs over the ROUNDED probabilities (an MFMA
        # of the packed tile with ones), earlier sums are rescaled in fp32
        c = scale * 1.4426950408889634
        m = torch.full(qf.shape[:-1] + (1,), float("-inf"), device=qf.device)
        l = torch.zeros_like(m)
        acc = torch.zeros(qf.shape[:-1] + (hdv,), device=qf.device)
        for t0 in range(0, n, 64):
            st = qf @ kf[:, :, t0:t0 + 64].transpose(-1, -2)
            m_new = torch.maximum(m, st.amax(dim=-1, keepdim=True) * c)
            alpha = torch.exp2(m - m_new)
            pt = torch.exp2(st * c - m_new).to(q.dtype).float()
            l = l * alpha + pt.sum(dim=-1, keepdim=True)
            acc = acc * alpha + pt @ vf[:, :, t0:t0 + 64]
            m = m_new
        o = (acc / l).permute(0, 2, 1, 3)                                # [batch, n, heads, hdv]
    else:
        o = (att.softmax(-1) @ vf).permute(0, 2, 1, 3)                   # [batch, n, heads, hdv]
    full = o.new_zeros((batch, stride, heads, hdv))
    full[:, :n] = o
    full = full.reshape(rows, heads * hdv)
    if out is not None:
        out[:rows].copy_(full.to(out.dtype))
        return out
    return full.to(q.dtype)


def relpos_extend(q, k, t, ty, tx, *, heads, head_stride, head_dim, hk, wk, ext_dim, scale, t_rows_per_token=None):
    rows = q.shape[0]
    per = ty.numel()
    y = ty.long().repeat(rows // per)
    x = tx.long().repeat(rows // per)
    qh = q[:, :heads * head_stride].float().reshape(rows, heads, head_stride)[..., :head_dim]
    kh = k[:, :heads * head_stride].float().reshape(rows, heads, head_stride)[..., :head_dim]
    tf = t.float().reshape(rows, t_rows_per_token or heads, -1)[:, :heads]
    qe = q.new_zeros((rows, heads, ext_dim), dtype=torch.float32)
    ke = torch.zeros_like(qe)
    qe[..., :head_dim] = qh * scale
    ke[..., :head_dim] = kh
    ih = y[:, None] - torch.arange(hk, device=q.device)[None] + hk - 1                      # [rows, hk]
    iw = x[:, None] - torch.arange(wk, device=q.device)[None] + wk - 1 + (2 * hk - 1)
    qe[..., head_dim:head_dim + hk] = torch.gather(tf, 2, ih[:, None].expand(-1, heads, -1))
    qe[..., head_dim + hk:head_dim + hk + wk] = torch.gather(tf, 2, iw[:, None].expand(-1, heads, -1))
    ke[..., head_dim:head_dim + hk] = torch.nn.functional.one_hot(y, hk)[:, None].float()
    ke[..., head_dim + hk:head_dim + hk + wk] = torch.nn.functional.one_hot(x, wk)[:, None].float()
    return qe.reshape(rows, -1).to(q.dtype), ke.reshape(rows, -1).to(q.dtype)


# ------------------------------------------------------------------------------------------------
# spatial gathers
# ------------------------------------------------------------------------------------------------
def patchify(img, tok2raster, ht, wt, mean, std, *, out_dtype, out=None):
    h, w = img.shape[1:]
    m = torch.tensor(mean, dtype=torch.float32, device=img.device).view(3, 1, 1)
    s = torch.tensor(std, dtype=torch.float32, device=img.device).view(3, 1, 1)
    x = F.pad((img.float() - m) / s, (0, wt * 16 - w, 0, ht * 16 - h))
    p = x.view(3, ht, 16, wt, 16).permute(1, 3, 0, 2, 4).reshape(ht * wt, 768)
    if tok2raster is not None:
        p = p[tok2raster.long()]
    if out is not None:
        out.copy_(p.to(out.dtype))
        return out
    return p.to(out_dtype).contiguous()


def im2col3x3(x, perm, h, w, *, out=None):
    C = x.shape[1]
    src = x[perm.long()] if perm is not None else x[: h * w]
    img = src.float().reshape(h, w, C).permute(2, 0, 1)[None]
    cols = F.unfold(img, kernel_size=3, padding=1)  # [1, C*9, h*w] with index c*9 + tap
    cols = cols.view(C, 9, h * w).permute(2, 1, 0).reshape(h * w, 9 * C).to(x.dtype)
    if out is not None:
        out.copy_(cols)
        return out
    return cols.contiguous()


def conv3x3_implicit_ok(x, w, h, wd):
    return False


def conv3x3(x, perm, h, wd, w, bias=None, *, out_dtype=None):
    return gemm(im2col3x3(x, perm, h, wd), w, bias, out_dtype=out_dtype)


def maxpool2x2(x, perm, h, w):
    C = x.shape[1]
    src = x[perm.long()] if perm is not None else x[: h * w]
    img = src.float().reshape(h, w, C).permute(2, 0, 1)[None]
    y = F.max_pool2d(img, 2, 2)[0].permute(1, 2, 0).reshape(-1, C)
    return y.to(x.dtype).contiguous()


def gather_rows(x, idx, *, out=None):
    y = x[idx.long()]
    if out is not None:
        out.copy_(y)
        return out
    return y.contiguous()


# ------------------------------------------------------------------------------------------------
# NMS (greedy, IoU > thr suppresses), VL pooling, mask post-processing
# ------------------------------------------------------------------------------------------------
def _greedy(boxes, order_idx, valid, thr):
    """visit boxes[order_idx[k]] in order k; returns keep flags per visiting position"""
    from oracle import thirdparty as tp  # test infrastructure only

    b = boxes.float().cpu()[order_idx.cpu().long()]
    n = b.shape[0]
    v = torch.ones(n, dtype=torch.bool) if valid is None else valid.cpu().bool()
    keep = torch.zeros(n, dtype=torch.uint8)
    if v.any():
        pos = v.nonzero().flatten()
        # tp.nms sorts by score: give strictly decreasing scores in visiting order
        kept = tp.nms(b[pos], torch.arange(len(pos), 0, -1).float(), thr)
        keep[pos[kept]] = 1
    return keep


def nms_segments(boxes, groups, seg_offsets, max_segment, iou_thr, valid=None):
    n = boxes.shape[0]
    keep = torch.zeros(n, dtype=torch.uint8)
    so = seg_offsets.cpu().tolist()
    g = groups.cpu()
    for s0, s1 in zip(so[:-1], so[1:]):
        if s1 <= s0:
            continue
        idx = torch.arange(s0, s1)
        # groups may still differ inside a segment: suppression only within equal group ids
        for gid in torch.unique(g[idx]):
            sub = idx[g[idx] == gid]
            keep[sub] = _greedy(boxes, sub, None if valid is None else valid[sub], iou_thr)
    return keep.to(boxes.device)


def nms_classes(boxes, order, iou_thr, valid=None):
    """greedy NMS per class, visiting boxes[order[c, k]] in order k.  The same arithmetic as oracle.thirdparty.nms (float32:
    inter / (area_i + area_j - inter) > thr) run for ALL classes at once -- one pass over the visiting positions with [K, n] tensors
    instead of K x n tiny tensor operations (1203 classes x 900 boxes took a minute of the GPU suite);
    `nms_classes_one_by_one` is the per-class form it is checked against (tests/test_host_model.py)."""
    K, n = order.shape
    dev = boxes.device
    b = boxes.float().cpu()[order.cpu().long()]                       # [K, n, 4] in visiting order
    v = torch.ones((K, n), dtype=torch.bool) if valid is None else valid.cpu().bool().clone()
    areas = (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
    suppressed = torch.zeros((K, n), dtype=torch.bool)
    keep = torch.zeros((K, n), dtype=torch.uint8)
    for i in range(n):
        alive = v[:, i] & ~suppressed[:, i]                           # classes whose i-th box is kept
        keep[:, i] = alive.to(torch.uint8)
        if i + 1 >= n or not bool(alive.any()):
            continue
        bi, rest = b[:, i:i + 1], b[:, i + 1:]
        xx1, yy1 = torch.maximum(bi[..., 0], rest[..., 0]), torch.maximum(bi[..., 1], rest[..., 1])
        xx2, yy2 = torch.minimum(bi[..., 2], rest[..., 2]), torch.minimum(bi[..., 3], rest[..., 3])
        inter = (xx2 - xx1).clamp(min=0) * (yy2 - yy1).clamp(min=0)
        ovr = inter / (areas[:, i:i + 1] + areas[:, i + 1:] - inter)
        # an invalid (skipped) box suppresses nothing; suppression of invalid positions is harmless (they are never kept)
        suppressed[:, i + 1:] |= alive[:, None] & (ovr > iou_thr)
    return keep.to(dev)


def nms_classes_one_by_one(boxes, order, iou_thr, valid=None):
    K, n = order.shape
    keep = torch.zeros((K, n), dtype=torch.uint8)
    for c in range(K):
        keep[c] = _greedy(boxes, order[c], None if valid is None else valid[c], iou_thr)
    return keep.to(boxes.device)


def vl_pool(scores, x, sub=None):
    w = scores.float()
    w = (w - w.max()).clamp(-50000, 50000)
    wl = (w - w.max(dim=0, keepdim=True)[0]).clamp(-50000, 50000).softmax(dim=0)  # over tokens
    out = wl.t() @ x.float()
    return out if sub is None else out - sub.float()[None, :]


def segment_softmax(scores, nseg, gmax, out_dtype):
    T, C = scores.shape
    w = (scores.float() - gmax.float()).clamp(-50000, 50000).reshape(T, nseg, C // nseg)
    return w.softmax(dim=-1).reshape(T, C).to(out_dtype)


def _pad_cols(t, pad):
    extra = (-t.shape[1]) % pad
    return F.pad(t, (0, extra)).contiguous() if extra else t.contiguous()


def col_softmax_t(scores, gmax, out_dtype, pad=1):
    w = (scores.float() - gmax.float()).clamp(-50000, 50000)
    wl = (w - w.max(dim=0, keepdim=True)[0]).clamp(-50000, 50000).softmax(dim=0)
    return _pad_cols(wl.t().to(out_dtype), pad)


def transpose(x, out_dtype=None, pad=1):
    return _pad_cols(x.t().to(out_dtype or x.dtype), pad)


def mask_upsample_bits(logits, h0, w0, size):
    n = logits.shape[0]
    up = F.interpolate(logits.float().reshape(1, n, h0, w0), size=(size, size), mode="bilinear", align_corners=False)[0]
    return (up > 0).to(torch.uint8)


def roi_align_bits(bits, boxes, p):
    from oracle import thirdparty as tp

    return tp.bitmasks_crop_and_resize(bits.cpu().bool(), boxes.cpu(), p).to(torch.uint8).to(bits.device)


def paste_bits(masks, boxes, ho, wo, out=None):
    from oracle import thirdparty as tp

    res = [tp.paste_mask(masks[i].cpu().float(), boxes[i].cpu(), ho, wo).to(torch.uint8) for i in range(masks.shape[0])]
    res = torch.stack(res).to(masks.device)
    if out is not None:
        out.copy_(res)
        return out
    return res


def mask_upsample_sigmoid(logits_t, h0, w0, size, crop_h, crop_w, out_dtype):
    n = logits_t.shape[1]
    up = F.interpolate(logits_t.float().t().reshape(1, n, h0, w0), size=(size, size), mode="bilinear", align_corners=False)[0]
    return up[:, :crop_h, :crop_w].sigmoid().reshape(n, -1).t().contiguous().to(out_dtype)


def arange_i64(n, device):
    return torch.arange(int(n), dtype=torch.int64, device=device)


def stuff_collapse(logits, nt):
    return torch.cat([logits[:, :nt].min(dim=1, keepdim=True)[0], logits[:, nt:]], dim=1).contiguous()


def sem_class_weights(logits, qidx, valid, temp, kp, out_dtype):
    cls = torch.softmax(logits.float()[qidx].sigmoid() / temp, dim=-1)
    if valid is not None:
        cls = cls * (valid.float() >= 0)[:, None]
    A = torch.zeros((cls.shape[1], kp), dtype=out_dtype, device=logits.device)
    A[:, : cls.shape[0]] = cls.t().to(out_dtype)
    return A


def pan_class_scores(logits, qidx, valid, thresh, transform, temp):
    x = logits.float() if qidx is None else logits.float()[qidx]
    sig = x.sigmoid()
    scores, labels = sig.max(-1)
    keep = scores > thresh
    if valid is not None:
        keep = keep & (valid.float() >= 0)
    if transform:
        scores, labels = torch.softmax(sig / temp, dim=-1).max(-1)
    return scores.contiguous(), labels, keep, labels.to(torch.int32)


def argmax_labels(x, class0=None, out=None):
    if class0 is not None:
        x = x.clone()
        x[0] = class0
    r = x.argmax(0).to(torch.int16)
    if out is not None:
        out.copy_(r)
        return out
    return r


def bilinear_resize(x, height, width):
    return F.interpolate(x.float()[None], size=(height, width), mode="bilinear", align_corners=False)[0]


def box_refine(delta, ref, vr4, eps=1e-3):
    new_ref = ref
    if delta is not None:
        x = ref.clamp(min=0, max=1)
        new_ref = (delta + torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))).sigmoid()
    return new_ref, (new_ref[:, None, :] * vr4[None]).contiguous()


def det_records(det_boxes, det_scores, det_classes, det_query, frame):
    boxes = torch.minimum((det_boxes * frame[:4]).clamp_min(0.0), frame[4:])
    keep = (det_scores >= 0) & ((boxes[:, 2] - boxes[:, 0]) > 0) & ((boxes[:, 3] - boxes[:, 1]) > 0)
    score = torch.where(keep, det_scores, torch.full_like(det_scores, -1.0))
    rec = torch.cat([boxes, score[:, None], det_classes[:, None].float(), det_query[:, None].float(), keep[:, None].float()], 1)
    order = torch.sort((~keep).to(torch.int8), stable=True)[1]
    return rec[order].contiguous(), boxes[order].contiguous(), order.to(torch.int32)


def query_init(coords_unact, topk, dim_t, out_dtype, scale=2 * math.pi):
    """deformable_transformer_vl.py:412-420, 629-634 for [T,4] unactivated coords and the selected tokens"""
    coords = coords_unact[topk.long()]
    pos = (coords.sigmoid() * scale)[:, :, None] / dim_t
    pe = torch.stack((pos[:, :, 0::2].sin(), pos[:, :, 1::2].cos()), dim=3).flatten(1)
    return coords.sigmoid(), pe.to(out_dtype).contiguous(), topk.to(torch.int32)


def query_finish(pos, pix, norm_pos, norm_pix, out_dtype):
    """deformable_transformer_vl.py:635-645: pos_trans_norm, split, + pix_trans_norm"""
    E = pix.shape[1]
    pt = F.layer_norm(pos.float(), (2 * E,), norm_pos[0].float(), norm_pos[1].float(), norm_pos[2])
    px = F.layer_norm(pix.float(), (E,), norm_pix[0].float(), norm_pix[1].float(), norm_pix[2])
    query_pos = pt[:, :E].to(out_dtype).contiguous()
    query = (pt[:, E:] + px).to(out_dtype).contiguous()
    return query_pos, query, (query.float() + query_pos.float()).to(out_dtype)


# ---- input pipeline / evaluator wire format: the definitions are the libraries the reference calls (Pillow) and the
#      oracle's restatement of pycocotools' RLE (oracle/imageio.py)
def resize_coeffs(in_size, out_size):
    from oracle import imageio as O
    b, k = O.precompute_coeffs(in_size, out_size)
    return torch.from_numpy(b), torch.from_numpy(k)


def resize_bilinear_u8(src, newh, neww, *, out=None, float_chw=False, flip=False):
    import numpy as np
    from PIL import Image
    a = src.cpu().numpy()
    r = np.asarray(Image.fromarray(a).resize((neww, newh), Image.BILINEAR)) if (newh, neww) != a.shape[:2] else a
    if flip:
        r = r[:, :, ::-1]
    t = torch.from_numpy(np.ascontiguousarray(r)).to(src.device)
    if float_chw:
        t = t.permute(2, 0, 1).float()
    if out is not None:
        out.copy_(t)
        return out
    return t.contiguous()


def rle_encode(masks, cap=4096, counts=None, nruns=None):
    from oracle import imageio as O
    n = masks.shape[0]
    out_c, out_n = counts, nruns
    counts = torch.zeros((n, cap), dtype=torch.int32)
    nruns = torch.zeros((n,), dtype=torch.int32)
    for i in range(n):
        c = O.rle_encode(masks[i].cpu().numpy() != 0)
        nruns[i] = len(c)
        m = min(len(c), cap)
        counts[i, :m] = torch.tensor(c[:m], dtype=torch.int64).to(torch.int32)
    if out_c is not None:
        out_c.copy_(counts)
        out_n.copy_(nruns)
        return out_c, out_n
    return counts.to(masks.device), nruns.to(masks.device)


def rle_to_string(counts):
    from oracle import imageio as O
    return O.rle_to_string([int(c) & 0xFFFFFFFF for c in (counts.tolist() if hasattr(counts, "tolist") else counts)])


# ------------------------------------------------------------------------------------------------
# data-dependent selections: tensor-level definitions (the sort / cumsum formulation the reference's code amounts to, with the
# defined tie rule: equal scores -> lowest index)
# ------------------------------------------------------------------------------------------------
def _stable_topk(values, k):
    return torch.sort(values, descending=True, stable=True)[1][:k]


def enc_finalize(cls2, d, anchors):
    pick = cls2[:, 1] > cls2[:, 0]
    enc_class = torch.where(pick, cls2[:, 1], cls2[:, 0])
    enc_coord = torch.where(pick[:, None], d[:, 4:], d[:, :4]) + anchors
    cs = enc_coord.sigmoid()
    xyxy = torch.stack([cs[:, 0] - 0.5 * cs[:, 2], cs[:, 1] - 0.5 * cs[:, 3], cs[:, 0] + 0.5 * cs[:, 2],
                        cs[:, 1] + 0.5 * cs[:, 3]], -1).clamp(0, 1)
    return enc_class, enc_coord, xyxy.contiguous()


def select_proposals(logit, xyxy, level_shapes, pre_nms_topk, num_queries, iou_thr):
    dev = logit.device
    T, L = logit.numel(), len(level_shapes)
    ns = [h * w for h, w in level_shapes]
    starts = [sum(ns[:i]) for i in range(L)]
    level_ids = torch.cat([torch.full((m,), i, dtype=torch.long) for i, m in enumerate(ns)]).to(dev)
    k = min(pre_nms_topk, T)
    nq = num_queries
    prob = logit.sigmoid()
    cands = []
    for n_l, start in zip(ns, starts):
        inl = start + _stable_topk(prob[start:start + n_l], min(k, n_l))
        if n_l < k:
            extra = torch.arange(k - n_l, device=dev)
            extra = torch.where(extra >= start, extra + n_l, extra)
            inl = torch.cat([inl, extra])
        cands.append(inl)
    cand = torch.cat(cands)
    n = cand.numel()
    sc, lv, bx = logit[cand], level_ids[cand], xyxy[cand]
    o1 = torch.sort(sc, descending=True, stable=True)[1]
    o2 = torch.sort(lv[o1], stable=True)[1]
    order = o1[o2]
    ar = torch.arange(L, device=dev)
    counts = (lv[None, :] == ar[:, None]).sum(1)
    seg = torch.cat([counts.new_zeros(1), counts.cumsum(0)]).to(torch.int32)
    max_seg = min(n, k + sum(max(0, k - m) for m in ns))
    keep_s = nms_segments(bx[order].float().contiguous(), lv[order].to(torch.int32).contiguous(), seg, max_seg, iou_thr)
    keep1 = torch.zeros(n, dtype=torch.bool, device=dev)
    keep1[o2] = keep_s.bool()
    cand1, lv1 = cand[o1], lv[o1]
    alt = _stable_topk(logit, min(nq, T))
    pad = n - alt.numel()
    cand_alt = torch.cat([alt, alt.new_zeros(pad)]) if pad > 0 else alt[:n]
    valid_alt = torch.arange(n, device=dev) < alt.numel()
    use_alt = keep1.sum() < nq
    candx = torch.where(use_alt, cand_alt, cand1)
    valid = torch.where(use_alt, valid_alt, keep1)
    lvx = torch.where(use_alt, level_ids[cand_alt], lv1)
    is_lvl = (lvx[None, :] == ar[:, None]) & valid[None, :]
    sel = (is_lvl & (is_lvl.cumsum(1) <= nq // L)).any(0)
    need = nq - sel.sum()
    notsel = valid & ~sel
    sel = sel | (notsel & (notsel.cumsum(0) <= need))
    slot = torch.where(sel, sel.cumsum(0) - 1, torch.full_like(candx, nq))
    out = torch.zeros(nq + 1, dtype=torch.long, device=dev)
    out.scatter_(0, slot, candx)
    return out[:nq]


def detections(logits, boxes, scale, score_thresh, iou_thr, topk):
    Q, K = logits.shape
    scores = logits.sigmoid()
    cx, cy, bw, bh = boxes.unbind(-1)
    xyxy = torch.stack([cx - 0.5 * bw, cy - 0.5 * bh, cx + 0.5 * bw, cy + 0.5 * bh], -1) * scale
    finite = torch.isfinite(xyxy).all(1) & torch.isfinite(scores).all(1)
    xyxy = torch.minimum(xyxy.clamp_min(0.0), scale)
    xyxy = torch.where(finite[:, None], xyxy, torch.zeros_like(xyxy)).contiguous()
    st = scores.t().contiguous()
    sorted_scores, order = torch.sort(st, dim=1, descending=True, stable=True)
    valid = (sorted_scores > score_thresh) & finite[order]
    keep = nms_classes(xyxy, order.to(torch.int32).contiguous(), iou_thr, valid.to(torch.uint8).contiguous())
    masked = torch.where(keep.bool(), sorted_scores, torch.full_like(sorted_scores, -1.0)).reshape(-1)
    k = min(topk, masked.numel())
    top_scores, flat = torch.sort(masked, descending=True, stable=True)
    top_scores, flat = top_scores[:k], flat[:k]
    cls = torch.div(flat, Q, rounding_mode="floor")
    qidx = order.reshape(-1)[flat]
    return dict(det_boxes=xyxy[qidx], det_scores=top_scores, det_classes=cls, det_query=qidx)


def ffn_fused(x, w1, b1, w2, b2, residual=None, out=None, w2_permuted=False, norm=None):
    """the two-GEMM form at the kernel's rounding points: H rounded to x's 16-bit type, fp32 accumulation, one rounding of the output"""
    if w2_permuted:
        from ape_amd.packing import ffn_w2_perm
        inv = torch.empty(w2.shape[1], dtype=torch.long)
        inv[ffn_w2_perm(w2.shape[1])] = torch.arange(w2.shape[1])
        w2 = w2[:, inv.to(w2.device)]
    h = torch.relu(x.float() @ w1.float().t() + b1.float()).to(x.dtype)
    y = h.float() @ w2.float().t() + b2.float()
    if residual is not None:
        y = y + residual.float()
    if norm is not None:                      # LayerNorm of the fp32 sums (the kernel's epilogue), one rounding of the result
        y = F.layer_norm(y, (y.shape[1],), norm[0].float(), norm[1].float(), norm[2])
    y = y.to(x.dtype)
    if out is not None:
        out.copy_(y)
        return out
    return y


def panoptic_merge(masks, scores, keep, classes, isthing, height, width, *, prob, overlap_threshold, stuff_offset=-1):
    """the reference's loop (deformable_detr_segm_vl.py:921-998) on fixed-size inputs: kept queries in index order"""
    k = masks.shape[0]
    dev = masks.device
    pr = bilinear_resize(masks, height, width).sigmoid()
    keep = keep.bool()
    panoptic_seg = torch.zeros((height, width), dtype=torch.int32, device=dev)
    info = torch.zeros((k, 3), dtype=torch.int32, device=dev)
    n = 0
    idx = torch.nonzero(keep).flatten().tolist()
    if idx:
        cur_masks = pr[idx]
        cur_mask_ids = (scores[idx].view(-1, 1, 1) * cur_masks).argmax(0)
        cur, stuff_memory = 0, {}
        for j, q in enumerate(idx):
            c = int(classes[q])
            thing = bool(isthing[c])
            own = cur_mask_ids == j
            conf = cur_masks[j] >= prob
            ma, oa = int(own.sum()), int(conf.sum())
            both = own & conf
            if ma > 0 and oa > 0 and int(both.sum()) > 0:
                if ma / oa < overlap_threshold:
                    continue
                if not thing:
                    if c in stuff_memory:
                        panoptic_seg[both] = stuff_memory[c]
                        continue
                    stuff_memory[c] = cur + 1
                cur += 1
                panoptic_seg[both] = cur
                info[n] = torch.tensor([cur, int(thing), c - stuff_offset + 1 if (not thing and stuff_offset >= 0) else c], dtype=torch.int32)
                n += 1
    return panoptic_seg, info, torch.tensor([n], dtype=torch.int32, device=dev)


def zeros(shape, dtype, device):
    return torch.zeros(shape, dtype=dtype, device=device)

