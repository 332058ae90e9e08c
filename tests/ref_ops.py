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
         rowmask=None, mask_mode=MASK_NONE, trans_out=False, rope=None, m_pad=None):
    M, K = a.shape
    N = w.shape[0]
    x = (a.float() @ w.float().t()) * alpha
    rm = rowmask.bool() if rowmask is not None else None
    if rm is not None and mask_mode == MASK_ZERO_INPUT:
        x = x.masked_fill(rm[:, None], 0.0)
    if bias is not None:
        x = x + bias.float()
    if rope is not None:
        cos, sin, rows, hd, cols = rope
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


def gemv(x, w, bias=None, alpha=1.0):
    y = (x.float() @ w.float().t()) * alpha
    if bias is not None:
        y = y + bias.float()
    return y


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
    loc, aw = msda_locations(offw, ref.reshape(batch * Q, -1, ref.shape[-1]), spatial_shapes)
    v = value[:, :256].float().reshape(batch, S, 8, 32)
    o = ms_deform_attn_core(v, spatial_shapes, loc.reshape(batch, Q, *loc.shape[1:]), aw.reshape(batch, Q, *aw.shape[1:]))
    o = o.reshape(batch * Q, 256)
    odt = out.dtype if out is not None else (out_dtype or value.dtype)
    if out is not None:
        out.copy_(o.to(odt))
        return out
    return o.to(odt)


def attention(q, k, vt, *, batch, n, heads, head_dim, scale, out=None):
    E = heads * head_dim
    qf = q[:, :E].float().reshape(batch, n, heads, head_dim).permute(0, 2, 1, 3)
    kf = k[:, :E].float().reshape(batch, n, heads, head_dim).permute(0, 2, 1, 3)
    vf = vt[:, : batch * n].float().reshape(heads, head_dim, batch, n).permute(2, 0, 3, 1)
    att = (qf @ kf.transpose(-1, -2)) * scale
    o = att.softmax(-1) @ vf
    o = o.permute(0, 2, 1, 3).reshape(batch * n, E)
    if out is not None:
        out.copy_(o.to(out.dtype))
        return out
    return o.to(q.dtype)
