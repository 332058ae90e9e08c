"""Tensor-level wrappers around the C-ABI (include/ape_hip.h): torch tensors in, device pointers out.

PyTorch is used here only as the allocator / stream provider.  Every function launches on
`torch.cuda.current_stream()`, never synchronises, and raises if a tensor is not on a HIP device or
the library is missing -- there is no CPU or PyTorch fallback in the product path.
"""
import ctypes
import math
import os

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_NONE, ACT_RELU, ACT_SILU, ACT_SWIGLU, DT_BF16, DT_F32, MASK_NONE,  # noqa: F401
                   MASK_ZERO_INPUT, MASK_ZERO_OUTPUT)

_DT = {torch.float32: DT_F32, torch.bfloat16: DT_BF16, torch.float16: _lib.DT_F16}
# the two 16-bit flavours of the pipeline (csrc/common.h h16<>): bfloat16 = BASELINE's dtype, float16 = the reference's own
# evaluation dtype (tools/train_net.py:642); every 16-bit operand of one call has the same one
HALF16 = (torch.bfloat16, torch.float16)


def _dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"ape_amd: unsupported dtype {t.dtype} (float32 / bfloat16 / float16 only)")


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("ape_amd.ops: tensor is not on a HIP device; the HIP path has no CPU fallback")


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _rowmajor(t, name):
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"ape_amd.ops: {name} must be 2-D with unit inner stride, got shape {tuple(t.shape)} stride {t.stride()}")
    return t


def _f32vec(t, name):
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError(f"ape_amd.ops: {name} must be a contiguous float32 tensor")
    return t


def _ld(t):
    # leading dimension of a 2-D row-major view (a single row may report any stride)
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def _auto_tiling(M, N, K, dtype, trans_out, act):
    """(tile64, splitk) chosen from graph-replayed measurements on MI355X (tools/gpu_probe_small.py):
    * K <= 512: 64x64 tiles always -- 32 KiB of LDS per workgroup lets 4-5 workgroups overlap on a CU, which is what
      a 4..16-iteration K loop needs (87296x2048x256: 369 vs 480 us; 900x256x256: 7 vs 20 us);
    * few output tiles: 64x64 tiles plus split-K up to ~256 workgroups (900x256x2048: 11 vs 38 us);
    * one 128x128 tile per CU: 64x64 tiles for K <= 1024 (4096x1024x1024: 26 vs 30 us; the transposed-output variant
      prefers 128x128: 19 vs 21 us); long K stays on the 128x128 kernel unsplit (4096x1024x2752: 48 vs 50 us split)."""
    if dtype not in HALF16 or K % 32 != 0:
        return 0, 1
    if K % 64 == 0 and K >= 512:
        # eight-wave 256 x 256 / 256 x 128 tiles with the counted-wait pipeline (csrc/gemm_p8.hip) once they fill the chip
        # (measured, profiles/r02_gemm_p8_table.log: 16384x5504x1024 286 -> 193 us, 65536x256x2304 122 -> 77 us,
        # 4096x2048x1024 35 -> 29 us with 256 x 128 tiles); a transposed output is the exchanged problem
        rows, cols = (N, M) if trans_out else (M, N)
        tm = (rows + 255) // 256
        if tm * ((cols + 255) // 256) >= 200:
            return 3, 1
        if tm * ((cols + 127) // 128) >= 200:
            return 4, 1
    t128 = ((M + 127) // 128) * ((N + 127) // 128)
    t64 = ((M + 63) // 64) * ((N + 63) // 64)
    splittable = not trans_out and act != ACT_SWIGLU and N % 4 == 0
    if K <= 512:
        return 1, 1
    if t128 < 96:
        return 1, (max(1, min(K // 512, 256 // t64)) if splittable else 1)
    if t128 <= 256 and K <= 1024 and not trans_out:
        return 1, 1
    return 0, 1


def gemm(a, w, bias=None, *, out=None, out_dtype=None, residual=None, act=ACT_NONE, alpha=1.0, clamp=0.0,
         rowmask=None, mask_mode=MASK_NONE, trans_out=False, rope=None, m_pad=None, splitk=None, tile64=None, rownorm=None, norm=None,
         rowstats=None):
    """C = epi(alpha * a @ w.T); see ApeGemmArgs in include/ape_hip.h for the epilogue order.

    rowstats = (cols, eps, colvec [N]): the folded LayerNorm of `a` over its first `cols` columns (the rest is zero padding) with the row
    statistics computed BY the GEMM launch (ApeGemmArgs.rowstat_cols: the 256 x 128 tile kernel); where that kernel does not apply
    (`gemm_rowstats_fusable`) the statistics come from a `row_stats` launch and travel as `rownorm` -- same result up to fp32 summation order.

    norm = (weight [N], bias [N], eps): LayerNorm of the finished row (after bias / residual) in the same launch -- the K = N = 256,
    M >= 2048, 16-bit kernel only (`gemm_norm_fusable`); anything else is an argument error of the library.

    rownorm = (rowscale [M], rowshift [M], colvec [N]) fp32: acc * rowscale[m] + rowshift[m] * colvec[n] right after alpha
    (a LayerNorm of `a` folded into this GEMM: statistics from `row_stats`, gamma folded into w, colvec = row sums of w).

    a [M,K], w [N,K] (same dtype).  rope = (cos, sin, rows, head_dim, cols[, packed pairs]).  trans_out returns C^T as
    [N, m_pad or M].  act=ACT_SWIGLU expects interleaved (gate, up) rows in w and returns N/2 columns.
    """
    _dev(a, w, bias, out, residual, rowmask)
    _rowmajor(a, "a"), _rowmajor(w, "w")
    if a.dtype != w.dtype:
        raise TypeError("ape_amd.ops.gemm: a and w must share a dtype")
    M, K = a.shape
    N, Kw = w.shape
    if K != Kw:
        raise ValueError(f"ape_amd.ops.gemm: K mismatch {K} vs {Kw}")
    n_out = N // 2 if act == ACT_SWIGLU else N
    if out is None:
        odt = out_dtype or a.dtype
        if trans_out:
            ld = m_pad or M
            out = torch.zeros((N, ld), dtype=odt, device=a.device) if ld != M else torch.empty((N, M), dtype=odt, device=a.device)
        else:
            out = torch.empty((M, n_out), dtype=odt, device=a.device)
    _rowmajor(out, "out")
    args = _lib.GemmArgs()
    args.A, args.W, args.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    args.bias = _f32vec(bias, "bias").data_ptr() if bias is not None else None
    args.M, args.N, args.K = M, N, K
    args.lda, args.ldw, args.ldc = _ld(a), _ld(w), _ld(out)
    # IEEE-half output: one kernel produces it (K = 256, >= 2048 rows, no residual) -- the launcher checks
    args.in_dt, args.out_dt = _dt(a), _dt(out)
    if residual is not None:
        _rowmajor(residual, "residual")
        args.residual, args.ldr, args.res_dt = residual.data_ptr(), _ld(residual), _dt(residual)
    if rowmask is not None:
        if rowmask.dtype not in (torch.uint8, torch.bool) or not rowmask.is_contiguous() or rowmask.numel() != M:
            raise ValueError("ape_amd.ops.gemm: rowmask must be a contiguous uint8/bool [M] tensor")
        args.rowmask = rowmask.data_ptr()
    args.mask_mode = mask_mode if rowmask is not None else MASK_NONE
    args.act, args.trans_out = act, 1 if trans_out else 0
    if rope is not None:
        cos, sin, rows, hd, cols = rope[:5]
        _dev(cos, sin)
        args.rope_cos, args.rope_sin = _f32vec(cos, "rope cos").data_ptr(), _f32vec(sin, "rope sin").data_ptr()
        args.rope_rows, args.rope_hd, args.rope_cols = rows, hd, cols
        if len(rope) > 5 and rope[5] is not None:        # packed (cos, sin) pairs [rows, hd / 2, 2]: see ApeGemmArgs.rope_cs
            cs = rope[5]
            _dev(cs)
            if cs.dtype != torch.float32 or not cs.is_contiguous() or cs.numel() != cos.numel():
                raise ValueError("ape_amd.ops.gemm: packed rope table must be contiguous float32 [rows, head_dim / 2, 2]")
            args.rope_cs = cs.data_ptr()
    args.alpha, args.clamp = float(alpha), float(clamp)
    t64, sk = _auto_tiling(M, N, K, a.dtype, trans_out, act)
    if rowstats is not None:
        if rownorm is not None:
            raise ValueError("ape_amd.ops.gemm: rowstats and rownorm are two ways of passing the same row terms")
        cols, eps, cv = rowstats
        _dev(cv)
        if cv.numel() != N or not 0 < cols <= K:
            raise ValueError("ape_amd.ops.gemm: rowstats = (cols <= K, eps, colvec [N])")
        fusable = (t64 == 4 and tile64 in (None, 4) and splitk in (None, 1) and N % 128 == 0 and act == ACT_NONE and alpha == 1.0 and not clamp > 0.0
                   and rowmask is None and rope is None and not trans_out and norm is None and os.environ.get("APE_NO_ROWSTAT") != "1"
                   and (residual is None or (_ld(residual) % 8 == 0 and residual.data_ptr() % 16 == 0)) and out.data_ptr() % 16 == 0
                   and (_ld(out) * out.element_size()) % 16 == 0)
        if fusable:
            args.colvec, args.rowstat_cols, args.rowstat_eps = _f32vec(cv, "colvec").data_ptr(), int(cols), float(eps)
        else:
            st = row_stats(a[:, :cols], eps)
            rownorm = (st[0], st[1], cv)
    if rownorm is not None:
        rs, sh, cv = rownorm
        _dev(rs, sh, cv)
        if rs.numel() != M or sh.numel() != M or cv.numel() != N:
            raise ValueError("ape_amd.ops.gemm: rownorm = (rowscale [M], rowshift [M], colvec [N])")
        args.rowscale, args.rowshift, args.colvec = (_f32vec(rs, "rowscale").data_ptr(), _f32vec(sh, "rowshift").data_ptr(),
                                                     _f32vec(cv, "colvec").data_ptr())
    if norm is not None:
        _dev(norm[0], norm[1])
        args.ln_w, args.ln_b, args.ln_eps = _f32vec(norm[0], "norm weight").data_ptr(), _f32vec(norm[1], "norm bias").data_ptr(), float(norm[2])
    if splitk is not None:
        sk = int(splitk)
    if tile64 is not None:
        t64 = int(tile64)
    args.tile64 = t64
    if sk > 1:
        ws = torch.empty((sk * M * N,), dtype=torch.float32, device=a.device)
        args.splitk, args.workspace = sk, ws.data_ptr()
    _lib.check(_lib.load().ape_hip_gemm(ctypes.byref(args), _stream()), "ape_hip_gemm")
    return out


def zeros(shape, dtype, device):
    """a zero-filled device tensor WITHOUT a tensor-library launch: torch.empty (no kernel) + the library's stream-ordered zero-fill
    (ape_hip_zero: hipMemsetAsync, a memset node under capture) -- the padded V^T operand buffers of the forward"""
    t = torch.empty(shape, dtype=dtype, device=device)
    _dev(t)
    _lib.check(_lib.load().ape_hip_zero(t.data_ptr(), t.numel() * t.element_size(), _stream()), "ape_hip_zero")
    return t


def gemm_norm_fusable(a, w, residual=None, out_dtype=None):
    """can `gemm(a, w, ..., residual=residual, norm=...)` run the LayerNorm in its epilogue?  (csrc/gemm.hip gemm_kres_ln_kernel)"""
    return (a.dtype in HALF16 and a.shape[1] == 256 and w.shape[0] == 256 and a.shape[0] >= 2048 and (out_dtype or a.dtype) == a.dtype
            and (residual is None or residual.dtype == a.dtype) and os.environ.get("APE_NO_LN_EPILOGUE") != "1")


def row_stats(x, eps):
    """LayerNorm statistics of the rows of x [M, C] (row-major view): (rstd [M], -mean*rstd [M]) fp32 -- the row terms of a
    LayerNorm folded into the consuming GEMM (`gemm(..., rownorm=...)`)."""
    _dev(x)
    _rowmajor(x, "x")
    M, C = x.shape
    out = torch.empty((2, M), dtype=torch.float32, device=x.device)
    rc = _lib.load().ape_hip_row_stats(_p(x), _ld(x), _dt(x), M, C, float(eps), _p(out[0]), _p(out[1]), _stream())
    _lib.check(rc, "ape_hip_row_stats")
    return out[0], out[1]


def gemv(x, w, bias=None, alpha=1.0, *, scale=None, add=None):
    """out[m,n] = scale[n] * (alpha * x[m,:] . w[n,:] + bias[n]); x fp32 [M,K] (M small), w f32/bf16 [N,K] -> fp32 [M,N].
    With `add` (fp32 [M,N]) returns (out, add + out): the element-wise tails of the single-token language side ride in the
    epilogue instead of separate element-wise launches."""
    _dev(x, w, bias, scale, add)
    _rowmajor(x, "x"), _rowmajor(w, "w")
    if x.dtype != torch.float32:
        raise TypeError("ape_amd.ops.gemv: x must be float32")
    M, K = x.shape
    N = w.shape[0]
    out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    if scale is None and add is None:
        rc = _lib.load().ape_hip_gemv(_p(x), _ld(x), _p(w), _ld(w), _dt(w), _p(_f32vec(bias, "bias")), _p(out), N, M, N, K,
                                     float(alpha), _stream())
        _lib.check(rc, "ape_hip_gemv")
        return out
    out2 = None
    if add is not None:
        _rowmajor(add, "add")
        if add.dtype != torch.float32 or tuple(add.shape) != (M, N):
            raise ValueError(f"ape_amd.ops.gemv: add must be float32 {(M, N)}")
        out2 = torch.empty((M, N), dtype=torch.float32, device=x.device)
    if scale is not None and scale.numel() != N:
        raise ValueError("ape_amd.ops.gemv: scale must have one entry per output column")
    rc = _lib.load().ape_hip_gemv_affine(_p(x), _ld(x), _p(w), _ld(w), _dt(w), _p(_f32vec(bias, "bias")), _p(out), N, M, N, K,
                                        float(alpha), _p(_f32vec(scale, "scale")), _p(add), _ld(add) if add is not None else 0,
                                        _p(out2), N, _stream())
    _lib.check(rc, "ape_hip_gemv_affine")
    return out if add is None else (out, out2)


def geometry(S, h, w, level_shapes, dim_t, level_embeds, offset, eps, scale, *, lvl_pos, mask_u8, mask, invalid_u8, enc_ref,
             proposals, valid_ratios, vr4, box_scale):
    """fill the per-image-size constant buffers of the deformable encoder for an (h, w) image inside the S x S pad
    (csrc/geometry.hip; the tensor-level definition is modeling/ape_deta/geometry.build_geometry + lvl_pos)."""
    _dev(dim_t, level_embeds, lvl_pos, mask_u8, mask, invalid_u8, enc_ref, proposals, valid_ratios, vr4, box_scale)
    L = len(level_shapes)
    T = sum(a * b for a, b in level_shapes)
    if lvl_pos.shape[0] != T or mask_u8.numel() != T or enc_ref.shape != (T, L, 2) or proposals.shape != (T, 4):
        raise ValueError("ape_amd.ops.geometry: buffer shapes do not match the level shapes")
    for t in (lvl_pos, mask_u8, mask, invalid_u8, enc_ref, proposals, valid_ratios, vr4, box_scale, dim_t, level_embeds):
        if not t.is_contiguous():
            raise ValueError("ape_amd.ops.geometry: contiguous buffers only")
    hw = (ctypes.c_int * (2 * L))(*[int(v) for ab in level_shapes for v in ab])
    rc = _lib.load().ape_hip_geometry(int(S), int(h), int(w), L, hw, _p(dim_t), dim_t.numel(), _p(level_embeds), float(offset),
                                     float(eps), float(scale), _p(lvl_pos), _dt(lvl_pos), _p(mask_u8), _p(mask), _p(invalid_u8),
                                     _p(enc_ref), _p(proposals), _p(valid_ratios), _p(vr4), _p(box_scale), _stream())
    _lib.check(rc, "ape_hip_geometry")


def head_gemv(x, w, bias=None, alpha=1.0, *, bf16_copy=False):
    """out[h, n] = alpha * x[h, :] . w[h, n, :] + bias[h, n]; x [H, D], w [H, N, D], bias [H, N] (all fp32) -> [H, N] fp32
    (bf16_copy: True or a 16-bit dtype -> (out, out rounded to that type; True = bfloat16), the copy written by the same launch)."""
    _dev(x, w, bias)
    if x.dtype != torch.float32 or w.dtype != torch.float32 or not w.is_contiguous():
        raise TypeError("ape_amd.ops.head_gemv: fp32 x and contiguous fp32 w")
    _rowmajor(x, "x")
    H, N, D = w.shape
    if x.shape != (H, D):
        raise ValueError(f"ape_amd.ops.head_gemv: x {tuple(x.shape)} vs w {tuple(w.shape)}")
    if bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous() or bias.numel() != H * N):
        raise ValueError("ape_amd.ops.head_gemv: bias must be a contiguous fp32 [H, N] tensor")
    out = torch.empty((H, N), dtype=torch.float32, device=x.device)
    cdt = torch.bfloat16 if bf16_copy is True else bf16_copy
    if bf16_copy and cdt not in HALF16:
        raise TypeError("ape_amd.ops.head_gemv: the copy is bfloat16 or float16")
    cp = torch.empty((H, N), dtype=cdt, device=x.device) if bf16_copy else None
    rc = _lib.load().ape_hip_head_gemv(_p(x), _ld(x), _p(w), _p(bias), _p(out), N, H, N, D, float(alpha), _p(cp), N,
                                       _dt(cp) if cp is not None else 0, _stream())
    _lib.check(rc, "ape_hip_head_gemv")
    return (out, cp) if bf16_copy else out


def layernorm(x, w, b, eps, *, out=None, out_dtype=None, act=ACT_NONE, cpad=None, add=None, out2=None):
    """Row LayerNorm; returns y, or (y, y + add) when `add` is given.  Columns C..cpad-1 of y are zeroed."""
    _dev(x, w, b, out, add, out2)
    _rowmajor(x, "x")
    M, C = x.shape
    cpad = cpad or C
    odt = out_dtype or x.dtype
    if out is None:
        out = torch.empty((M, cpad), dtype=odt, device=x.device)
    args = _lib.LayerNormArgs()
    args.x, args.w, args.b, args.y = x.data_ptr(), _f32vec(w, "w").data_ptr(), _f32vec(b, "b").data_ptr(), out.data_ptr()
    args.M, args.C, args.Cpad = M, C, cpad
    args.ldx, args.ldy = _ld(x), _ld(out)
    args.x_dt, args.y_dt, args.act, args.eps = _dt(x), _dt(out), act, float(eps)
    if add is not None:
        _rowmajor(add, "add")
        if out2 is None:
            out2 = torch.empty((M, cpad), dtype=out.dtype, device=x.device)
        args.add, args.y2, args.ldadd, args.ldy2, args.add_dt = add.data_ptr(), out2.data_ptr(), _ld(add), _ld(out2), _dt(add)
    _lib.check(_lib.load().ape_hip_layernorm(ctypes.byref(args), _stream()), "ape_hip_layernorm")
    return (out, out2) if add is not None else out


def postnorm_residual(stream, t, norm, copy_dtype=None):
    """Post-norm residual step (vit_eva_clip.py:505-523, postnorm=True): stream += LayerNorm(t) IN PLACE (stream fp32 [M, C]);
    returns the new stream in `copy_dtype` (None: no copy).  t None: only the copy.  norm = (weight, bias, eps)."""
    _dev(stream, t)
    _rowmajor(stream, "stream")
    if stream.dtype != torch.float32:
        raise TypeError("ape_amd.ops.postnorm_residual: the residual stream is float32")
    M, C = stream.shape
    if copy_dtype not in (None, torch.float32) + HALF16 or (t is not None and t.dtype not in (torch.float32,) + HALF16):
        raise TypeError("ape_amd.ops.postnorm_residual: t and the copy are float32, bfloat16 or float16")
    copy = torch.empty((M, C), dtype=copy_dtype, device=stream.device) if copy_dtype is not None else None
    w = b = None
    eps = 0.0
    if t is not None:
        _rowmajor(t, "t")
        if tuple(t.shape) != (M, C):
            raise ValueError("ape_amd.ops.postnorm_residual: t must match the stream")
        w, b, eps = _f32vec(norm[0], "w"), _f32vec(norm[1], "b"), float(norm[2])
    rc = _lib.load().ape_hip_postnorm_residual(_p(t), _ld(t) if t is not None else 0, _dt(t) if t is not None else 0, _p(w), _p(b), eps,
                                               _p(stream), _ld(stream), _p(copy), C, _dt(copy) if copy is not None else 0, M, C, _stream())
    _lib.check(rc, "ape_hip_postnorm_residual")
    return copy


def groupnorm(x, w, b, groups, eps, *, act=ACT_NONE, add=None, out=None, out_dtype=None):
    """GroupNorm over a token-major map x[HW, C]: y = act(GN(x) * w + b + add)."""
    _dev(x, w, b, add, out)
    _rowmajor(x, "x")
    HW, C = x.shape
    if out is None:
        out = torch.empty((HW, C), dtype=out_dtype or x.dtype, device=x.device)
    lib = _lib.load()
    ws = torch.empty((lib.ape_hip_groupnorm_workspace_floats(HW, groups),), dtype=torch.float32, device=x.device)
    args = _lib.GroupNormArgs()
    args.x, args.w, args.b, args.y = x.data_ptr(), _f32vec(w, "w").data_ptr(), _f32vec(b, "b").data_ptr(), out.data_ptr()
    args.workspace = ws.data_ptr()
    args.HW, args.C, args.G = HW, C, groups
    args.ldx, args.ldy = _ld(x), _ld(out)
    args.x_dt, args.y_dt, args.act, args.eps = _dt(x), _dt(out), act, float(eps)
    if add is not None:
        _rowmajor(add, "add")
        args.add, args.ldadd, args.add_dt = add.data_ptr(), _ld(add), _dt(add)
    _lib.check(lib.ape_hip_groupnorm(ctypes.byref(args), _stream()), "ape_hip_groupnorm")
    return out


def _levels(spatial_shapes, level_start_index):
    if torch.is_tensor(spatial_shapes):
        spatial_shapes = spatial_shapes.detach().cpu().tolist()
    if torch.is_tensor(level_start_index):
        level_start_index = level_start_index.detach().cpu().tolist()
    L = len(spatial_shapes)
    flat = [int(v) for hw in spatial_shapes for v in hw]
    shp = (ctypes.c_int64 * (2 * L))(*flat)
    st = (ctypes.c_int64 * L)(*[int(v) for v in level_start_index])
    return shp, st, L


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step=64):
    """Drop-in for torch.ops.ape.ms_deform_attn_forward (ape/layers/csrc/vision.cpp:76-79).

    value [B,S,8,32], sampling_loc [B,Q,8,L,4,2], attn_weight [B,Q,8,L,4] (same dtype) -> [B,Q,256].
    """
    _dev(value, sampling_loc, attn_weight)
    if not (value.is_contiguous() and sampling_loc.is_contiguous() and attn_weight.is_contiguous()):
        raise ValueError("ms_deform_attn_forward: inputs must be contiguous")  # ms_deform_attn_cuda.cu:29-33
    if not (value.dtype == sampling_loc.dtype == attn_weight.dtype):
        raise TypeError("ms_deform_attn_forward: value / sampling_loc / attn_weight must share a dtype")
    B, S, M, D = value.shape
    Q = sampling_loc.shape[1]
    if M != 8 or D != 32 or sampling_loc.shape[4] != 4:
        raise ValueError("ms_deform_attn_forward: built for 8 heads x 32 channels x 4 points")
    shp, st, L = _levels(spatial_shapes, level_start_index)
    if sampling_loc.shape[3] != L:
        raise ValueError("ms_deform_attn_forward: num_levels mismatch")
    out = torch.empty((B, Q, M * D), dtype=value.dtype, device=value.device)
    dt = _dt(value)          # APE_DT_F16: half storage, fp32 arithmetic
    rc = _lib.load().ape_hip_ms_deform_attn_forward(_p(value), M * D, shp, st, _p(sampling_loc), _p(attn_weight), _p(out),
                                                   M * D, B, S, Q, L, dt, _stream())
    _lib.check(rc, "ape_hip_ms_deform_attn_forward")
    return out


def msda_fused(value, spatial_shapes, level_start_index, offw, ref, *, batch=1, out_dtype=None, out=None):
    """Fused softmax + sampling-location + bilinear gather (multi_scale_deform_attn.py:278-348).

    value [batch*S, >=256] (row-major view; f32, bf16 or IEEE half), offw [batch*Q, 8*L*4*3] fp32 or (bf16 / half values only)
    fp16 (offsets then logits), ref [batch*Q, L, 2|4] fp32 -> [batch*Q, 256].
    """
    _dev(value, offw, ref, out)
    _rowmajor(value, "value"), _rowmajor(offw, "offw")
    if offw.dtype not in (torch.float32, torch.float16) or ref.dtype != torch.float32 or not ref.is_contiguous():
        raise TypeError("ape_amd.ops.msda_fused: offw must be float32 / float16, ref contiguous float32")
    shp, st, L = _levels(spatial_shapes, level_start_index)
    S = value.shape[0] // batch
    Q = offw.shape[0] // batch
    if offw.shape[1] != 8 * L * 4 * 3 or ref.shape[-2] != L:
        raise ValueError("ape_amd.ops.msda_fused: offw/ref shape mismatch")
    if out is None:
        # half values: the f16 flavour of the pipeline writes f16; the bf16 flavour (half VALUES next to bf16 activations) passes
        # out_dtype=torch.bfloat16 explicitly (layers/multi_scale_deform_attn.py)
        out = torch.empty((batch * Q, 256), dtype=out_dtype or value.dtype, device=value.device)
    if offw.dtype == torch.float16 and value.dtype == torch.float32:
        raise TypeError("ape_amd.ops.msda_fused: half offsets go with bf16 / f16 values")
    fn = _lib.load().ape_hip_msda_fused_h if offw.dtype == torch.float16 else _lib.load().ape_hip_msda_fused
    v_dt = _dt(value)
    rc = fn(_p(value), _ld(value), v_dt, shp, st, _p(offw), _ld(offw), _p(ref), ref.shape[-1], _p(out), _ld(out), _dt(out),
            batch, S, Q, L, _stream())
    _lib.check(rc, "ape_hip_msda_fused")
    return out


def attention(q, k, vt, *, batch, n, heads, head_dim, scale, out=None, stride=None, causal=False, v_head_dim=None):
    """softmax(scale * q k^T) v per (window, head); causal: keys after the query are masked (CLIP text tower).  q,k: [rows, >=heads*head_dim] views; vt: V transposed
    [heads*v_head_dim, >= (batch-1)*stride + round_up(n, 64)] (finite padding; checked); returns [rows, heads*v_head_dim].  Window
    b owns rows b*stride .. b*stride+n-1 (stride defaults to n; rows = (batch-1)*stride + n ... batch*stride).
    v_head_dim (default head_dim): the V / output width per head when it differs from the q.k width -- the relative-position
    channels of ape_amd.ops.relpos_extend ride in q / k only (head_dim 128 | 256 | 288 | 320 over v_head_dim 128)."""
    _dev(q, k, vt, out)
    _rowmajor(q, "q"), _rowmajor(k, "k"), _rowmajor(vt, "vt")
    if not (q.dtype == k.dtype == vt.dtype):
        raise TypeError("ape_amd.ops.attention: q/k/vt must share a dtype")
    stride = n if stride is None else int(stride)
    # the kernels read V^T in 64-column tiles starting at each window's first column (keys beyond n get weight 0, but the
    # columns must exist and be finite): the last window's last tile ends at (batch - 1) * stride + round_up(n, 64)
    need = (batch - 1) * stride + (n + 63) // 64 * 64
    if vt.shape[1] < need or q.shape[0] < (batch - 1) * stride + n:
        raise ValueError(f"ape_amd.ops.attention: vt has {vt.shape[1]} columns, the tiled reads need {need} "
                         f"((batch - 1) * stride + round_up(n, 64)); q has {q.shape[0]} rows")
    hdv = head_dim if v_head_dim is None else int(v_head_dim)
    if out is None:
        out = (torch.empty if stride == n else torch.zeros)((batch * stride, heads * hdv), dtype=q.dtype, device=q.device)
    if hdv != head_dim:
        if causal:
            raise ValueError("ape_amd.ops.attention: v_head_dim != head_dim has no causal variant")
        rc = _lib.load().ape_hip_attention_ext(_p(q), _ld(q), _p(k), _ld(k), _p(vt), _ld(vt), _p(out), _ld(out), batch, n, stride, heads,
                                               head_dim, hdv, float(scale), _dt(q), _stream())
        _lib.check(rc, "ape_hip_attention_ext")
        return out
    fn = _lib.load().ape_hip_attention_causal if causal else _lib.load().ape_hip_attention_strided
    rc = fn(_p(q), _ld(q), _p(k), _ld(k), _p(vt), _ld(vt), _p(out), _ld(out), batch, n, stride, heads, head_dim, float(scale),
            _dt(q), _stream())
    _lib.check(rc, "ape_hip_attention")
    return out


def relpos_extend(q, k, t, ty, tx, *, heads, head_stride, head_dim, hk, wk, ext_dim, scale, t_rows_per_token=None):
    """operands of attention with decomposed relative positions (ape/modeling/backbone/vit_eva.py:121-146, utils_eva.py:132-161):
    q, k [rows, >= heads * head_stride] views (head h at columns h * head_stride .. + head_dim); t [rows * t_rows_per_token (default heads),
    >= 2 hk - 1 + 2 wk - 1] = q . [Rh ; Rw]^T, the row of (token, head) being token * t_rows_per_token + head; ty / tx int32 [period]: position of token (row % period) in its attention group.  Returns
    q_ext = [scale q | q.Rh[ty - kh] | q.Rw[tx - kw] | 0], k_ext = [k | one-hot ty | one-hot tx | 0]: [rows, heads * ext_dim] each."""
    _dev(q, k, t, ty, tx)
    _rowmajor(q, "q"), _rowmajor(k, "k"), _rowmajor(t, "t")
    if not (q.dtype == k.dtype == t.dtype) or _ld(q) != _ld(k):
        raise TypeError("ape_amd.ops.relpos_extend: q / k / t must share a dtype (and q, k a row stride)")
    _i32(ty, "ty"), _i32(tx, "tx")
    rows = q.shape[0]
    tper = heads if t_rows_per_token is None else int(t_rows_per_token)
    if t.shape[0] != rows * tper or ty.numel() != tx.numel() or rows % ty.numel():
        raise ValueError("ape_amd.ops.relpos_extend: t must have rows * t_rows_per_token rows; rows a multiple of the coordinate period")
    qe = torch.empty((rows, heads * ext_dim), dtype=q.dtype, device=q.device)
    ke = torch.empty_like(qe)
    rc = _lib.load().ape_hip_relpos_extend(_p(q), _p(k), _ld(q), _p(t), _ld(t), tper, _p(ty), _p(tx), ty.numel(), _p(qe), _p(ke), _ld(qe), rows,
                                           heads, head_stride, head_dim, hk, wk, ext_dim, float(scale), _dt(q), _stream())
    _lib.check(rc, "ape_hip_relpos_extend")
    return qe, ke


def embed_tokens(tokens, table, pos, length, stride):
    """CLIP text tower input: out[b * stride + t] = table[tokens[b, t]] + pos[t] (t < length), zero rows up to `stride`;
    tokens int32 [B, ctx]; table [vocab, W], pos [ctx, W] (f32 / bf16) -> fp32 [B * stride, W]"""
    _dev(tokens, table, pos)
    if tokens.dtype != torch.int32 or tokens.dim() != 2 or tokens.stride(1) != 1:
        raise ValueError("ape_amd.ops.embed_tokens: tokens must be int32 [B, ctx] with unit inner stride")
    _rowmajor(table, "table"), _rowmajor(pos, "pos")
    if table.dtype != pos.dtype or table.shape[1] != pos.shape[1] or length > tokens.shape[1] or length > pos.shape[0] or stride < length:
        raise ValueError("ape_amd.ops.embed_tokens: inconsistent table / pos / length / stride")
    B, W = tokens.shape[0], table.shape[1]
    out = torch.empty((B * stride, W), dtype=torch.float32, device=tokens.device)
    rc = _lib.load().ape_hip_embed_tokens(_p(tokens), tokens.stride(0), _p(table), _ld(table), _p(pos), _ld(pos), _dt(table), _p(out),
                                         W, B, int(length), int(stride), W, table.shape[0], _stream())
    _lib.check(rc, "ape_hip_embed_tokens")
    return out


def _i32(t, name):
    if t is None:
        return None
    if t.dtype != torch.int32 or not t.is_contiguous():
        raise ValueError(f"ape_amd.ops: {name} must be a contiguous int32 tensor")
    return t


def patchify(img, tok2raster, ht, wt, mean, std, *, out_dtype, out=None):
    """(img - mean)/std, zero pad to the token grid, 16x16 patch rows [ht*wt, 768] in token order."""
    _dev(img, tok2raster, out)
    if img.dtype != torch.float32 or img.dim() != 3 or img.shape[0] != 3 or not img.is_contiguous():
        raise ValueError("ape_amd.ops.patchify: img must be contiguous float32 [3,h,w]")
    if out is None:
        out = torch.empty((ht * wt, 768), dtype=out_dtype, device=img.device)
    elif tuple(out.shape) != (ht * wt, 768) or not out.is_contiguous():
        raise ValueError("ape_amd.ops.patchify: out must be a contiguous [ht*wt, 768] tensor")
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = (ctypes.c_float * 3)(*[float(v) for v in std])
    rc = _lib.load().ape_hip_patchify(_p(img), img.shape[1], img.shape[2], _p(_i32(tok2raster, "tok2raster")), ht, wt, m, s,
                                     _p(out), 768, _dt(out), _stream())
    _lib.check(rc, "ape_hip_patchify")
    return out


def im2col3x3(x, perm, h, w, *, out=None):
    """[h*w, C] (rows addressed through perm: raster index -> row) -> [h*w, 9*C] operand of a 3x3/pad-1 conv."""
    _dev(x, perm, out)
    _rowmajor(x, "x")
    C = x.shape[1]
    if out is None:
        out = torch.empty((h * w, 9 * C), dtype=x.dtype, device=x.device)
    rc = _lib.load().ape_hip_im2col3x3(_p(x), _ld(x), _p(_i32(perm, "perm")), h, w, C, _p(out), _ld(out), _dt(x), _stream())
    _lib.check(rc, "ape_hip_im2col3x3")
    return out


_ZERO_ROWS = {}


def conv3x3_implicit_ok(x, w, h, wd):
    """does `conv3x3` run as the implicit-GEMM flavour of the eight-wave tile kernel?  (16-bit, 256 channels; >= 200 tiles of 256 x 256
    -- the 256 x 256-pixel maps -- or >= 100 tiles of 256 x 128: the 128 x 128-pixel p3 map)"""
    rows = (h * wd + 255) // 256
    return (x.dtype in HALF16 and x.dtype == w.dtype and x.shape[1] == 256 and w.shape[1] == 9 * 256
            and (rows * ((w.shape[0] + 255) // 256) >= 200
                 or (rows * ((w.shape[0] + 127) // 128) >= 100 and os.environ.get("APE_CONV_P3_IM2COL") != "1"))
            and os.environ.get("APE_CONV_IM2COL") != "1")


def conv3x3(x, perm, h, wd, w, bias=None, *, out_dtype=None):
    """3 x 3 / stride 1 / zero padding 1 convolution of a token-major map: x [rows, C] (raster pixel r lives in row perm[r]; perm None:
    row r), w [N, 9 C] in (ky, kx, ci) order -> [h * wd, N].  Large 256-channel maps in a 16-bit type run as an IMPLICIT GEMM (the tile
    kernel stages its A operand from the shifted input rows: ApeGemmArgs.conv_*; the [h wd, 9 C] im2col matrix never exists);
    everything else is im2col3x3 + gemm.  Bit-identical either way."""
    _dev(x, perm, w, bias)
    if perm is not None and perm.numel() != h * wd:
        raise ValueError(f"ape_amd.ops.conv3x3: perm has {perm.numel()} entries for a {h} x {wd} map")
    if perm is None and x.shape[0] < h * wd:
        raise ValueError(f"ape_amd.ops.conv3x3: x has {x.shape[0]} rows for a {h} x {wd} map")
    # the tile kernel addresses source rows with 32-bit byte offsets (perm[r] * lda * 2): a perm that reaches into a token buffer
    # beyond 4 GiB goes through im2col, whose gather uses 64-bit addresses
    if not conv3x3_implicit_ok(x, w, h, wd) or x.shape[0] * _ld(x) * 2 >= 2 ** 32:
        return gemm(im2col3x3(x, perm, h, wd), w, bias, out_dtype=out_dtype)
    _rowmajor(x, "x"), _rowmajor(w, "w")
    M, N = h * wd, w.shape[0]
    out = torch.empty((M, N), dtype=out_dtype or x.dtype, device=x.device)
    key = (x.device, )
    if key not in _ZERO_ROWS:
        _ZERO_ROWS[key] = torch.zeros((256,), dtype=torch.uint8, device=x.device)
    args = _lib.GemmArgs()
    args.A, args.W, args.C = x.data_ptr(), w.data_ptr(), out.data_ptr()
    args.bias = _f32vec(bias, "bias").data_ptr() if bias is not None else None
    args.M, args.N, args.K = M, N, 9 * 256
    args.lda, args.ldw, args.ldc = _ld(x), _ld(w), _ld(out)
    args.in_dt, args.out_dt = _dt(x), _dt(out)
    args.act, args.mask_mode, args.alpha, args.tile64 = ACT_NONE, MASK_NONE, 1.0, 3
    args.conv_perm = _i32(perm, "perm").data_ptr() if perm is not None else None
    args.conv_zero, args.conv_h, args.conv_w = _ZERO_ROWS[key].data_ptr(), int(h), int(wd)
    rc = _lib.load().ape_hip_gemm(ctypes.byref(args), _stream())
    _lib.check(rc, "ape_hip_gemm(conv3x3)")
    return out


def maxpool2x2(x, perm, h, w):
    _dev(x, perm)
    _rowmajor(x, "x")
    C = x.shape[1]
    out = torch.empty(((h // 2) * (w // 2), C), dtype=x.dtype, device=x.device)
    rc = _lib.load().ape_hip_maxpool2x2(_p(x), _ld(x), _p(_i32(perm, "perm")), h, w, C, _p(out), C, _dt(x), _stream())
    _lib.check(rc, "ape_hip_maxpool2x2")
    return out


def gather_rows(x, idx, *, out=None):
    """out[r] = x[idx[r]]; idx int32 or int64 (contiguous)"""
    _dev(x, idx, out)
    _rowmajor(x, "x")
    n, C = idx.numel(), x.shape[1]
    if out is None:
        out = torch.empty((n, C), dtype=x.dtype, device=x.device)
    if idx.dtype == torch.int64:
        if not idx.is_contiguous():
            raise ValueError("ape_amd.ops.gather_rows: idx must be contiguous")
        rc = _lib.load().ape_hip_gather_rows_i64(_p(x), _ld(x), _p(idx), n, C, _p(out), _ld(out), _dt(x), _stream())
        _lib.check(rc, "ape_hip_gather_rows_i64")
        return out
    rc = _lib.load().ape_hip_gather_rows(_p(x), _ld(x), _p(_i32(idx, "idx")), n, C, _p(out), _ld(out), _dt(x), _stream())
    _lib.check(rc, "ape_hip_gather_rows")
    return out


def _boxes(b):
    if b.dtype != torch.float32 or not b.is_contiguous() or b.dim() != 2 or b.shape[1] != 4:
        raise ValueError("ape_amd.ops: boxes must be contiguous float32 [n,4]")
    return b


def _u8(t, name):
    if t is None:
        return None
    if t.dtype not in (torch.uint8, torch.bool) or not t.is_contiguous():
        raise ValueError(f"ape_amd.ops: {name} must be contiguous uint8/bool")
    return t


def nms_segments(boxes, groups, seg_offsets, max_segment, iou_thr, valid=None):
    """Greedy NMS inside each contiguous segment (candidates sorted by descending score within a segment).
    boxes [n,4] xyxy, groups int32 [n], seg_offsets int32 [G+1] (device) -> keep uint8 [n]."""
    _dev(boxes, groups, seg_offsets, valid)
    n = boxes.shape[0]
    lib = _lib.load()
    mask = torch.empty((n, lib.ape_hip_nms_mask_words(n)), dtype=torch.int64, device=boxes.device)
    keep = torch.zeros((n,), dtype=torch.uint8, device=boxes.device)
    _lib.check(lib.ape_hip_nms_mask(_p(_boxes(boxes)), _p(_i32(groups, "groups")), n, float(iou_thr), _p(mask), _stream()), "ape_hip_nms_mask")
    rc = lib.ape_hip_nms_scan_segments(_p(mask), n, _p(_i32(seg_offsets, "seg_offsets")), seg_offsets.numel() - 1, int(max_segment),
                                       _p(_u8(valid, "valid")), _p(keep), _stream())
    _lib.check(rc, "ape_hip_nms_scan_segments")
    return keep


def nms_classes(boxes, order, iou_thr, valid=None):
    """Class-wise greedy NMS over class-agnostic boxes: class c visits boxes order[c, :] (descending score).
    boxes [n,4], order int32 [K,n], valid uint8 [K,n] -> keep uint8 [K,n] (in visiting order)."""
    _dev(boxes, order, valid)
    n = boxes.shape[0]
    K = order.shape[0]
    lib = _lib.load()
    mask = torch.empty((n, lib.ape_hip_nms_mask_words(n)), dtype=torch.int64, device=boxes.device)
    keep = torch.zeros((K, n), dtype=torch.uint8, device=boxes.device)
    _lib.check(lib.ape_hip_nms_mask(_p(_boxes(boxes)), None, n, float(iou_thr), _p(mask), _stream()), "ape_hip_nms_mask")
    rc = lib.ape_hip_nms_scan_classes(_p(mask), n, _p(_i32(order, "order")), K, _p(_u8(valid, "valid")), _p(keep), _stream())
    _lib.check(rc, "ape_hip_nms_scan_classes")
    return keep


def vl_pool(scores, x, sub=None):
    """out[h,:] = sum_t softmax_t(scores[t,h]) x[t,:] - sub  (single-text-token language side, fuse_helper.py:89-116,140; sub [C]
    fp32: pooling x - sub)."""
    _dev(scores, x, sub)
    _rowmajor(scores, "scores"), _rowmajor(x, "x")
    if scores.dtype != torch.float32 or scores.shape[1] != 8:
        raise ValueError("ape_amd.ops.vl_pool: scores must be float32 [T,8]")
    T, C = x.shape
    if sub is not None and sub.numel() != C:
        raise ValueError("ape_amd.ops.vl_pool: sub must have one entry per channel")
    lib = _lib.load()
    ws = torch.empty((lib.ape_hip_vl_pool_workspace_floats(T, C),), dtype=torch.float32, device=x.device)
    out = torch.empty((8, C), dtype=torch.float32, device=x.device)
    rc = lib.ape_hip_vl_pool(_p(scores), _ld(scores), _p(x), _ld(x), _dt(x), T, C, _p(ws), _p(_f32vec(sub, "sub")), _p(out), _stream())
    _lib.check(rc, "ape_hip_vl_pool")
    return out


def segment_softmax(scores, nseg, gmax, out_dtype):
    """Vision side of the dense bi-attention: softmax over each of the nseg column segments of every row of
    clamp(scores - gmax) (fuse_helper.py:89-99,131).  scores fp32 [T, nseg*L], gmax fp32 device scalar."""
    _dev(scores, gmax)
    _rowmajor(scores, "scores")
    if scores.dtype != torch.float32 or gmax.dtype != torch.float32 or scores.shape[1] % nseg:
        raise ValueError("ape_amd.ops.segment_softmax: scores must be float32 [T, nseg*L], gmax a float32 scalar")
    T, C = scores.shape
    out = torch.empty((T, C), dtype=out_dtype, device=scores.device)
    rc = _lib.load().ape_hip_segment_softmax(_p(scores), _ld(scores), T, nseg, C // nseg, _p(gmax), _p(out), _ld(out), _dt(out),
                                             _stream())
    _lib.check(rc, "ape_hip_segment_softmax")
    return out


def _padded(rows, cols, pad, dtype, device):
    cp = (cols + pad - 1) // pad * pad
    out = torch.empty((rows, cp), dtype=dtype, device=device)
    if cp != cols:
        out[:, cols:].zero_()
    return out


def col_softmax_t(scores, gmax, out_dtype, pad=1):
    """Language side of the dense bi-attention: softmax over the T rows of clamp(scores - gmax), returned TRANSPOSED
    [C, Tp] (fuse_helper.py:101-116) -- the A operand of the token reduction; Tp = T rounded up to `pad`, zero filled."""
    _dev(scores, gmax)
    _rowmajor(scores, "scores")
    if scores.dtype != torch.float32 or gmax.dtype != torch.float32:
        raise ValueError("ape_amd.ops.col_softmax_t: scores must be float32, gmax a float32 scalar")
    T, C = scores.shape
    lib = _lib.load()
    ws = torch.empty((lib.ape_hip_colstats_workspace_floats(T, C),), dtype=torch.float32, device=scores.device)
    stats = torch.empty((2, C), dtype=torch.float32, device=scores.device)
    rc = lib.ape_hip_colstats(_p(scores), _ld(scores), T, C, _p(gmax), _p(ws), _p(stats[0]), _p(stats[1]), _stream())
    _lib.check(rc, "ape_hip_colstats")
    out = _padded(C, T, pad, out_dtype, scores.device)
    rc = lib.ape_hip_transpose(_p(scores), _ld(scores), _dt(scores), T, C, _p(gmax), _p(stats[0]), _p(stats[1]), _p(out), _ld(out),
                               _dt(out), _stream())
    _lib.check(rc, "ape_hip_transpose")
    return out


def transpose(x, out_dtype=None, pad=1):
    """out[c, t] = x[t, c] (materialised; feeds GEMM operands that must be K-contiguous); columns zero-padded to `pad`."""
    _dev(x)
    _rowmajor(x, "x")
    T, C = x.shape
    out = _padded(C, T, pad, out_dtype or x.dtype, x.device)
    rc = _lib.load().ape_hip_transpose(_p(x), _ld(x), _dt(x), T, C, None, None, None, _p(out), _ld(out), _dt(out), _stream())
    _lib.check(rc, "ape_hip_transpose")
    return out


def mask_upsample_bits(logits, h0, w0, size):
    """bilinear (align_corners=False) upsample of n mask-logit rows [n, h0*w0] to size x size, thresholded at 0."""
    _dev(logits)
    _rowmajor(logits, "logits")
    n = logits.shape[0]
    out = torch.empty((n, size, size), dtype=torch.uint8, device=logits.device)
    rc = _lib.load().ape_hip_mask_upsample_bits(_p(logits), _ld(logits), _dt(logits), h0, w0, size, n, _p(out), _stream())
    _lib.check(rc, "ape_hip_mask_upsample_bits")
    return out


def roi_align_bits(bits, boxes, p):
    """BitMasks.crop_and_resize: bits uint8 [n,H,W], boxes [n,4] -> uint8 [n,p,p]."""
    _dev(bits, boxes)
    n, H, W = bits.shape
    out = torch.empty((n, p, p), dtype=torch.uint8, device=bits.device)
    rc = _lib.load().ape_hip_roi_align_bits(_p(_u8(bits, "bits")), H, W, _p(_boxes(boxes)), n, p, _p(out), _stream())
    _lib.check(rc, "ape_hip_roi_align_bits")
    return out


def paste_bits(masks, boxes, ho, wo, out=None):
    """paste_masks_in_image: masks uint8 [n,P,P], boxes [n,4] (output frame) -> uint8 [n,ho,wo]."""
    _dev(masks, boxes, out)
    n, P, _ = masks.shape
    if out is None:
        out = torch.empty((n, ho, wo), dtype=torch.uint8, device=masks.device)
    elif out.dtype != torch.uint8 or tuple(out.shape) != (n, ho, wo) or not out.is_contiguous():
        raise ValueError("ape_amd.ops.paste_bits: out must be a contiguous uint8 [n, ho, wo] tensor")
    rc = _lib.load().ape_hip_paste_bits(_p(_u8(masks, "masks")), P, _p(_boxes(boxes)), n, ho, wo, _p(out), _stream())
    _lib.check(rc, "ape_hip_paste_bits")
    return out


def mask_upsample_sigmoid(logits_t, h0, w0, size, crop_h, crop_w, out_dtype):
    """sigmoid of the bilinear (align_corners=False) upsample h0 x w0 -> size x size of PIXEL-MAJOR mask logits
    [h0*w0, n], for the top-left crop_h x crop_w region only -> [crop_h*crop_w, n] (pixel-major probabilities)."""
    _dev(logits_t)
    _rowmajor(logits_t, "logits_t")
    if logits_t.shape[0] != h0 * w0:
        raise ValueError("ape_amd.ops.mask_upsample_sigmoid: logits_t must be [h0*w0, n]")
    n = logits_t.shape[1]
    out = torch.empty((crop_h * crop_w, n), dtype=out_dtype, device=logits_t.device)
    rc = _lib.load().ape_hip_mask_upsample_sigmoid(_p(logits_t), _ld(logits_t), _dt(logits_t), h0, w0, size, crop_h, crop_w, n,
                                                   _p(out), _ld(out), _dt(out), _stream())
    _lib.check(rc, "ape_hip_mask_upsample_sigmoid")
    return out


_ARANGE = {}


def arange_i64(n, device):
    """the constant [0, 1, ..., n - 1] int64 on `device`, created once per (n, device) -- the first call happens in the un-captured warm-up
    runs of a step, so a captured step only READS it (no tensor-library launch inside the graph)"""
    key = (int(n), str(device))
    if key not in _ARANGE:
        _ARANGE[key] = torch.arange(int(n), dtype=torch.int64, device=device)
    return _ARANGE[key]


def stuff_collapse(logits, nt):
    """get_stuff_score (deformable_detr_segm_vl.py:1251-1271) with a leading "things" stuff class: [Q, K] fp32 -> [Q, K - nt + 1] (column 0 =
    the minimum over the nt thing columns, then the stuff columns)"""
    _dev(logits)
    _rowmajor(logits, "logits")
    Q, K = logits.shape
    out = torch.empty((Q, K - nt + 1), dtype=torch.float32, device=logits.device)
    _lib.check(_lib.load().ape_hip_stuff_collapse(_p(logits), _ld(logits), Q, K, int(nt), _p(out), _ld(out), _stream()), "ape_hip_stuff_collapse")
    return out


def sem_class_weights(logits, qidx, valid, temp, kp, out_dtype):
    """softmax_c(sigmoid(logits[qidx]) / temp) * valid (:891-894), TRANSPOSED and zero-padded to kp query columns: the [K, kp] A operand of
    the semantic branch's class x query product.  logits [Q, K] fp32, qidx int64 [k], valid: fp32 [k] detection scores (a row counts iff its
    score >= 0: the fixed-shape detection lists mark empty slots with -1) or None"""
    _dev(logits, qidx, valid)
    _rowmajor(logits, "logits")
    if qidx.dtype != torch.int64 or not qidx.is_contiguous():
        raise TypeError("ape_amd.ops.sem_class_weights: qidx must be contiguous int64")
    k, K = qidx.numel(), logits.shape[1]
    A = torch.empty((K, kp), dtype=out_dtype, device=logits.device)
    v = _f32vec(valid, "valid")
    rc = _lib.load().ape_hip_sem_class_weights(_p(logits), _ld(logits), _p(qidx), _p(v), k, int(kp), K, float(temp), _p(A), _ld(A), _dt(A), _stream())
    _lib.check(rc, "ape_hip_sem_class_weights")
    return A


def pan_class_scores(logits, qidx, valid, thresh, transform, temp):
    """_postprocess_panoptic's per-query scores (:944-949): -> (score fp32 [k], label int64 [k], keep bool [k], label int32 [k]); logits [Q, K] fp32, qidx int64 [k]
    (None: the rows themselves), valid: fp32 [k] detection scores (valid iff >= 0) or None"""
    _dev(logits, qidx, valid)
    _rowmajor(logits, "logits")
    k = qidx.numel() if qidx is not None else logits.shape[0]
    dev = logits.device
    score = torch.empty((k,), dtype=torch.float32, device=dev)
    label = torch.empty((k,), dtype=torch.int64, device=dev)          # the kernel writes int64 (torch.max's index type)
    keep = torch.empty((k,), dtype=torch.bool, device=dev)
    label32 = torch.empty((k,), dtype=torch.int32, device=dev)
    v = _f32vec(valid, "valid")
    rc = _lib.load().ape_hip_pan_class_scores(_p(logits), _ld(logits), _p(qidx), _p(v), k, logits.shape[1], float(thresh), 1 if transform else 0,
                                              float(temp), _p(score), _p(label), _p(label32), _p(keep), _stream())
    _lib.check(rc, "ape_hip_pan_class_scores")
    return score, label, keep, label32


def argmax_labels(x, class0=None, out=None):
    """int16 argmax over the class axis of fp32 scores [C, H, W] (contiguous) -> [H, W]; class0: constant that replaces class 0's scores
    (deformable_detr_segm_vl.py:654-663)"""
    _dev(x, out)
    if x.dtype != torch.float32 or x.dim() != 3 or not x.is_contiguous():
        raise ValueError("ape_amd.ops.argmax_labels: x must be contiguous float32 [C, H, W]")
    C, H, W = x.shape
    if out is None:
        out = torch.empty((H, W), dtype=torch.int16, device=x.device)
    rc = _lib.load().ape_hip_argmax_labels(_p(x), H * W, C, H * W, float("nan") if class0 is None else float(class0), _p(out), _stream())
    _lib.check(rc, "ape_hip_argmax_labels")
    return out


def bilinear_resize(x, height, width):
    """F.interpolate(x[None], (height, width), mode="bilinear", align_corners=False)[0] for fp32 x [C, h, w]
    (rows contiguous; channel / row strides free)."""
    _dev(x)
    if x.dtype != torch.float32 or x.dim() != 3 or x.stride(2) != 1:
        raise ValueError("ape_amd.ops.bilinear_resize: x must be float32 [C, h, w] with contiguous rows")
    C, h, w = x.shape
    out = torch.empty((C, height, width), dtype=torch.float32, device=x.device)
    rc = _lib.load().ape_hip_bilinear_resize(_p(x), x.stride(0), x.stride(1), h, w, C, _p(out), height, width, _stream())
    _lib.check(rc, "ape_hip_bilinear_resize")
    return out


def panoptic_merge(masks, scores, keep, classes, isthing, height, width, *, prob, overlap_threshold, stuff_offset=-1):
    """_postprocess_panoptic (deformable_detr_segm_vl.py:921-998) without a host round trip: masks [k, h, w] fp32 mask logits at the
    input resolution (rows contiguous), scores [k] fp32, keep [k] bool, classes [k] (any int type), isthing [num_classes] bool ->
    (panoptic_seg int32 [height, width], info int32 [k, 3] = (id, isthing, category_id) per segment, count int32 [1]); stuff_offset >= 0
    remaps a stuff segment's category to class - stuff_offset + 1 (:985-986)."""
    _dev(masks, scores, keep, classes, isthing)
    if masks.dtype != torch.float32 or masks.dim() != 3 or masks.stride(2) != 1 or scores.dtype != torch.float32:
        raise ValueError("ape_amd.ops.panoptic_merge: masks must be float32 [k, h, w] with contiguous rows, scores float32")
    k, h, w = masks.shape
    dev = masks.device
    scores = scores.contiguous()
    # bool -> uint8 views are free; the class ids arrive as int64 from pan_class_scores: their LOW words are the int32 ids (little endian,
    # ids >= 0), read through a stride-2 view by a one-launch gather only when they are not int32 already
    keep8 = keep.view(torch.uint8) if keep.dtype == torch.bool and keep.is_contiguous() else keep.to(torch.uint8).contiguous()
    cls32 = classes if classes.dtype == torch.int32 and classes.is_contiguous() else classes.to(torch.int32).contiguous()
    thing8 = isthing.view(torch.uint8) if isthing.dtype == torch.bool and isthing.is_contiguous() else isthing.to(torch.uint8).contiguous()
    owner = torch.empty((height, width), dtype=torch.int16, device=dev)
    conf = torch.empty((height, width), dtype=torch.uint8, device=dev)
    areas = torch.empty((k, 3), dtype=torch.int32, device=dev)
    seg_id = torch.empty((k,), dtype=torch.int32, device=dev)
    info = torch.empty((k, 3), dtype=torch.int32, device=dev)
    count = torch.empty((1,), dtype=torch.int32, device=dev)
    out = torch.empty((height, width), dtype=torch.int32, device=dev)
    lib = _lib.load()
    _lib.check(lib.ape_hip_panoptic_pixels(_p(masks), masks.stride(0), masks.stride(1), h, w, k, _p(scores), _p(keep8), float(prob), height, width,
                                           _p(owner), _p(conf), _p(areas), _stream()), "ape_hip_panoptic_pixels")
    _lib.check(lib.ape_hip_panoptic_decide(_p(areas), _p(cls32), _p(keep8), k, _p(thing8), thing8.numel(), float(overlap_threshold), int(stuff_offset),
                                           _p(seg_id), _p(info), _p(count), _stream()), "ape_hip_panoptic_decide")
    _lib.check(lib.ape_hip_panoptic_write(_p(owner), _p(conf), _p(seg_id), height, width, _p(out), _stream()), "ape_hip_panoptic_write")
    return out, info, count


def box_refine(delta, ref, vr4, eps=1e-3):
    """Decoder box refinement: new_ref = sigmoid(delta + inverse_sigmoid(ref, eps)) (delta None -> new_ref = ref) and the
    per-level MSDA reference ref_in[q, l, :] = new_ref[q, :] * vr4[l, :].  fp32 [Q,4] / [L,4] -> ([Q,4], [Q,L,4])."""
    _dev(delta, ref, vr4)
    if ref.dtype != torch.float32 or vr4.dtype != torch.float32 or not ref.is_contiguous() or not vr4.is_contiguous():
        raise TypeError("ape_amd.ops.box_refine: ref / vr4 must be contiguous float32")
    if delta is not None:
        _rowmajor(delta, "delta")
        if delta.dtype != torch.float32 or delta.shape != ref.shape:
            raise TypeError("ape_amd.ops.box_refine: delta must be float32 [Q,4]")
    Q, L = ref.shape[0], vr4.shape[0]
    new_ref = torch.empty_like(ref) if delta is not None else ref
    ref_in = torch.empty((Q, L, 4), dtype=torch.float32, device=ref.device)
    rc = _lib.load().ape_hip_box_refine(_p(delta), _ld(delta) if delta is not None else 0, _p(ref), _p(vr4), L, Q, float(eps),
                                        _p(new_ref) if delta is not None else None, _p(ref_in), _stream())
    _lib.check(rc, "ape_hip_box_refine")
    return new_ref, ref_in


def det_records(det_boxes, det_scores, det_classes, det_query, frame):
    """Detection records in the output frame, kept rows first (csrc/boxes.hip det_records_kernel): det_boxes [k,4] / det_scores [k]
    fp32, det_classes / det_query [k] int64, frame [8] fp32 (sx, sy, sx, sy, width, height, width, height) ->
    (rec [k,8] = box, score | -1, class, query, keep; boxes [k,4]; order [k] int32 = source row of every output row)."""
    _dev(det_boxes, det_scores, det_classes, det_query, frame)
    k = det_scores.numel()
    for t, dt_, n in ((det_boxes, torch.float32, 4 * k), (det_scores, torch.float32, k), (det_classes, torch.int64, k),
                      (det_query, torch.int64, k), (frame, torch.float32, 8)):
        if t.dtype != dt_ or not t.is_contiguous() or t.numel() != n:
            raise ValueError("ape_amd.ops.det_records: contiguous fp32 boxes [k,4] / scores [k] / frame [8], int64 classes / query [k]")
    dev = det_boxes.device
    rec = torch.empty((k, 8), dtype=torch.float32, device=dev)
    boxes = torch.empty((k, 4), dtype=torch.float32, device=dev)
    order = torch.empty((k,), dtype=torch.int32, device=dev)
    rc = _lib.load().ape_hip_det_records(_p(det_boxes), _p(det_scores), _p(det_classes), _p(det_query), _p(frame), k, _p(rec), _p(boxes),
                                        _p(order), _stream())
    _lib.check(rc, "ape_hip_det_records")
    return rec, boxes, order


def query_init(coords_unact, topk, dim_t, out_dtype, scale=2 * math.pi):
    """Two-stage query initialisation, part 1 (deformable_transformer_vl.py:412-420, 629-634): coords_unact [T,4] fp32, topk [Q]
    int64 -> (reference [Q,4] fp32 = sigmoid(coords[topk]), pe [Q, 4P] out_dtype = sine embedding of the proposals, topk as int32)."""
    _dev(coords_unact, topk, dim_t)
    if coords_unact.dtype != torch.float32 or not coords_unact.is_contiguous() or coords_unact.dim() != 2 or coords_unact.shape[1] != 4:
        raise ValueError("ape_amd.ops.query_init: coords must be contiguous float32 [T,4]")
    if topk.dtype != torch.int64 or not topk.is_contiguous():
        raise ValueError("ape_amd.ops.query_init: topk must be contiguous int64")
    T, Q, P = coords_unact.shape[0], topk.numel(), dim_t.numel()
    dev = coords_unact.device
    reference = torch.empty((Q, 4), dtype=torch.float32, device=dev)
    pe = torch.empty((Q, 4 * P), dtype=out_dtype, device=dev)
    topk32 = torch.empty((Q,), dtype=torch.int32, device=dev)
    rc = _lib.load().ape_hip_query_init(_p(coords_unact), _p(topk), T, _p(_f32vec(dim_t, "dim_t")), P, float(scale), Q, _p(reference),
                                       _p(pe), 4 * P, _dt(pe), _p(topk32), _stream())
    _lib.check(rc, "ape_hip_query_init")
    return reference, pe, topk32


def query_finish(pos, pix, norm_pos, norm_pix, out_dtype):
    """Two-stage query initialisation, part 2 (:635-645): pos [Q, 2E] / pix [Q, E] fp32 GEMM outputs, norm_* = (weight, bias, eps)
    -> (query_pos, query, query + query_pos) [Q, E] in out_dtype."""
    _dev(pos, pix)
    _rowmajor(pos, "pos"), _rowmajor(pix, "pix")
    Q, E = pix.shape
    if pos.dtype != torch.float32 or pix.dtype != torch.float32 or pos.shape != (Q, 2 * E):
        raise ValueError("ape_amd.ops.query_finish: pos [Q, 2E] / pix [Q, E] float32")
    outs = [torch.empty((Q, E), dtype=out_dtype, device=pos.device) for _ in range(3)]
    rc = _lib.load().ape_hip_query_finish(_p(pos), _ld(pos), _p(pix), _ld(pix), Q, E, _p(_f32vec(norm_pos[0], "w")), _p(_f32vec(norm_pos[1], "b")),
                                         float(norm_pos[2]), _p(_f32vec(norm_pix[0], "w")), _p(_f32vec(norm_pix[1], "b")), float(norm_pix[2]),
                                         _p(outs[0]), _p(outs[1]), _p(outs[2]), E, _dt(outs[0]), _stream())
    _lib.check(rc, "ape_hip_query_finish")
    return tuple(outs)


# ------------------------------------------------------------------------------------------------------------------
# Fork / join of independent launch sequences.  Inside a hipGraph capture the side streams become parallel branches
# of the graph, which the GPU executes concurrently (measured: 40 small GEMMs 219 -> 170 us, 4096x1024x1024 GEMM pairs
# -25 %): the decoder / language-side kernels are latency bound and leave most CUs idle, and two M = 4096 GEMMs
# together fill the chip better than one after the other.  Discipline: a branch starts behind everything the caller has
# enqueued so far and the caller joins before it consumes the branch's outputs, so the caching allocator's per-stream
# reuse stays safe.  APE_NO_FORK=1 (or no HIP device) runs the branches inline.
# ------------------------------------------------------------------------------------------------------------------
_SIDE_STREAMS = {}
_FORK_DEPTH = [0]


class _Joined:
    def __init__(self, result, done=None, keep=None):
        # `keep` (the branch closure, i.e. its input tensors) stays referenced until join(): the caller's stream must not
        # recycle those blocks while the branch may still be reading them
        self.result, self.done, self.keep = result, done, keep

    def join(self):
        if self.done is not None:
            torch.cuda.current_stream().wait_event(self.done)
            self.done = None
        self.keep = None
        return self.result


_FORK_INLINE = [False]
_FORK_NEST = [0]


class inline_forks:
    """context: nested ops.fork calls run inline (used when whole images are already parallel branches of one graph)"""

    def __enter__(self):
        self.prev = _FORK_INLINE[0]
        _FORK_INLINE[0] = True

    def __exit__(self, *exc):
        _FORK_INLINE[0] = self.prev


def fork(fn, force=False):
    """run fn() on a side stream, concurrent with whatever the caller enqueues next; returns a handle with .join()"""
    if not torch.cuda.is_available() or os.environ.get("APE_NO_FORK") == "1" or (_FORK_INLINE[0] and not force):
        return _Joined(fn())
    if _FORK_NEST[0] > 0 and os.environ.get("APE_FORK_NESTED") != "1":
        # branches do not fork again: one level of side streams per parent.  (Graphs whose capture nested side streams two
        # levels deep crashed hipGraphLaunch once a process had built three of them -- ROCm 7.2 -- and the second level
        # bought nothing measurable.)
        return _Joined(fn())
    dev = torch.cuda.current_device()
    cur = torch.cuda.current_stream()
    # one pool of side streams PER PARENT stream: a branch's outputs are consumed by its parent, and the next branch on the
    # same side stream starts behind the parent's position -- two pipelines forked from different parents (two images in
    # flight inside one graph) must therefore never share side streams
    pool = _SIDE_STREAMS.setdefault((dev, cur.cuda_stream), [torch.cuda.Stream(device=dev) for _ in range(8)])
    side = pool[_FORK_DEPTH[0] % len(pool)]
    _FORK_DEPTH[0] += 1
    start = torch.cuda.Event()
    start.record(cur)
    side.wait_event(start)
    with torch.cuda.stream(side):
        _FORK_NEST[0] += 1
        try:
            result = fn()
        finally:
            _FORK_NEST[0] -= 1
        done = torch.cuda.Event()
        done.record(side)
    return _Joined(result, done, keep=fn)


# ------------------------------------------------------------------------------------------------------------------
# Input pipeline / evaluator wire format (csrc/imageio.hip).  The coefficient tables and the RLE string are HOST functions
# of the library (plain C, usable without a device); the pixel work are kernels.
# ------------------------------------------------------------------------------------------------------------------
def resize_coeffs(in_size, out_size):
    """Pillow's bilinear coefficients of one axis: (bounds [out, 2] int32, kk [out, ksize] int32) as host tensors"""
    lib = _lib.load()
    ks = lib.ape_hip_resize_coeffs(int(in_size), int(out_size), None, None, 0)
    if ks <= 0:
        _lib.check(ks, "ape_hip_resize_coeffs")
    bounds = torch.empty((out_size, 2), dtype=torch.int32)
    kk = torch.empty((out_size, ks), dtype=torch.int32)
    rc = lib.ape_hip_resize_coeffs(int(in_size), int(out_size), ctypes.c_void_p(bounds.data_ptr()), ctypes.c_void_p(kk.data_ptr()), ks)
    if rc != ks:
        _lib.check(rc if rc < 0 else -1, "ape_hip_resize_coeffs")
    return bounds, kk


class _ResizePlan:
    """device coefficient tables + tiling of one (H, W) -> (newh, neww) resize; built once per size pair"""

    def __init__(self, H, W, newh, neww, device):
        self.key = (H, W, newh, neww)
        self.bh = self.kh = self.bv = self.kv = None
        self.ks_h = self.ks_v = 0
        self.tile_h, self.lds_rows = 32, 32
        if neww != W:
            b, k = resize_coeffs(W, neww)
            self.bh, self.kh, self.ks_h = b.to(device), k.to(device), k.shape[1]
        if newh != H:
            b, k = resize_coeffs(H, newh)
            th = ctypes.c_int(0)
            rows = _lib.load().ape_hip_resize_tile_rows(ctypes.c_void_p(b.data_ptr()), newh, ctypes.byref(th))
            if rows <= 0:
                _lib.check(rows if rows < 0 else -1, "ape_hip_resize_tile_rows")
            self.tile_h, self.lds_rows = th.value, rows
            self.bv, self.kv, self.ks_v = b.to(device), k.to(device), k.shape[1]


_RESIZE_PLANS = {}


def resize_bilinear_u8(src, newh, neww, *, out=None, float_chw=False, flip=False):
    """PIL `Image.resize((neww, newh), BILINEAR)` of an HWC uint8 device image, bit exact.  float_chw=False -> uint8
    [newh, neww, 3]; True -> float32 [3, newh, neww] (the model's `image` input; `out` may be a view with free row / plane
    strides, e.g. the top-left corner of a larger canvas).  flip reverses the channel order (BGR <-> RGB)."""
    _dev(src, out)
    if src.dtype != torch.uint8 or src.dim() != 3 or src.shape[2] != 3 or src.stride(2) != 1 or src.stride(1) != 3:
        raise ValueError("ape_amd.ops.resize_bilinear_u8: src must be uint8 [H, W, 3] with packed pixels")
    H, W = src.shape[:2]
    key = (H, W, newh, neww, src.device.index)
    plan = _RESIZE_PLANS.get(key)
    if plan is None:
        if len(_RESIZE_PLANS) > 256:
            _RESIZE_PLANS.clear()
        plan = _RESIZE_PLANS[key] = _ResizePlan(H, W, newh, neww, src.device)
    if float_chw:
        if out is None:
            out = torch.empty((3, newh, neww), dtype=torch.float32, device=src.device)
        if out.dtype != torch.float32 or tuple(out.shape) != (3, newh, neww) or out.stride(2) != 1:
            raise ValueError("ape_amd.ops.resize_bilinear_u8: out must be float32 [3, newh, neww] with contiguous rows")
        kind, ld, plane = 1, out.stride(1), out.stride(0)
    else:
        if out is None:
            out = torch.empty((newh, neww, 3), dtype=torch.uint8, device=src.device)
        if out.dtype != torch.uint8 or tuple(out.shape) != (newh, neww, 3) or out.stride(2) != 1 or out.stride(1) != 3:
            raise ValueError("ape_amd.ops.resize_bilinear_u8: out must be uint8 [newh, neww, 3] with packed pixels")
        kind, ld, plane = 0, out.stride(0), 0
    rc = _lib.load().ape_hip_resize_bilinear_u8(_p(src), H, W, src.stride(0), _p(plan.bh), _p(plan.kh), plan.ks_h, _p(plan.bv),
                                                _p(plan.kv), plan.ks_v, newh, neww, plan.tile_h, plan.lds_rows, _p(out), kind,
                                                ld, plane, int(bool(flip)), _stream())
    _lib.check(rc, "ape_hip_resize_bilinear_u8")
    return out


def rle_encode(masks, cap=4096, counts=None, nruns=None):
    """COCO run lengths of n row-major masks [n, H, W] (uint8 / bool, non-zero = foreground): (counts [n, cap] int32 view of
    uint32, nruns [n] int32) on the device; nruns[i] > cap marks a truncated encoding.  counts / nruns: optional
    preallocated outputs (contiguous int32 [n, cap] / [n])."""
    _dev(masks)
    if masks.dtype == torch.bool:
        masks = masks.view(torch.uint8)
    if masks.dtype != torch.uint8 or masks.dim() != 3 or not masks.is_contiguous():
        raise ValueError("ape_amd.ops.rle_encode: masks must be contiguous uint8 / bool [n, H, W]")
    n, H, W = masks.shape
    if counts is None:
        counts = torch.empty((n, cap), dtype=torch.int32, device=masks.device)
    if nruns is None:
        nruns = torch.zeros((n,), dtype=torch.int32, device=masks.device)
    _dev(counts, nruns)
    if (counts.dtype != torch.int32 or nruns.dtype != torch.int32 or tuple(counts.shape) != (n, cap) or tuple(nruns.shape) != (n,)
            or not counts.is_contiguous() or not nruns.is_contiguous()):
        raise ValueError("ape_amd.ops.rle_encode: counts / nruns must be contiguous int32 [n, cap] / [n]")
    if n == 0:
        return counts, nruns
    lib = _lib.load()
    words = lib.ape_hip_rle_workspace_words(n, H, W, cap)
    if words <= 0:
        raise ValueError("ape_amd.ops.rle_encode: workspace exceeds 2^31 words; encode fewer masks per call")
    ws = torch.empty((words,), dtype=torch.int32, device=masks.device)
    rc = lib.ape_hip_rle_encode(_p(masks), n, H, W, _p(ws), _p(counts), cap, _p(nruns), _stream())
    _lib.check(rc, "ape_hip_rle_encode")
    return counts, nruns


def rle_to_string(counts):
    """host: run lengths (1-D int32/uint32 host tensor or sequence) -> the bytes stored under "counts" in COCO json"""
    c = torch.as_tensor(counts, dtype=torch.int64).to(torch.int32).contiguous() if not isinstance(counts, torch.Tensor) \
        else counts.contiguous()
    if c.is_cuda or c.dtype not in (torch.int32,):
        raise ValueError("ape_amd.ops.rle_to_string: counts must be a host int32 tensor")
    n = c.numel()
    buf = ctypes.create_string_buffer(max(8, 7 * n))
    rc = _lib.load().ape_hip_rle_to_string(ctypes.c_void_p(c.data_ptr()), n, buf, len(buf))
    if rc < 0:
        raise RuntimeError("ape_hip_rle_to_string: buffer too small")
    return buf.raw[:rc]


# ------------------------------------------------------------------------------------------------------------------
# Data-dependent selections (csrc/topk.hip): fixed shapes, no host synchronisation
# ------------------------------------------------------------------------------------------------------------------
def _level_arrays(level_shapes):
    L = len(level_shapes)
    ns = [int(h) * int(w) for h, w in level_shapes]
    starts = [sum(ns[:i]) for i in range(L)]
    return (ctypes.c_int * L)(*starts), (ctypes.c_int * L)(*ns), L, starts, ns


def _f32c(t, name, shape=None):
    if t.dtype != torch.float32 or not t.is_contiguous() or (shape is not None and tuple(t.shape) != tuple(shape)):
        raise ValueError(f"ape_amd.ops: {name} must be contiguous float32" + (f" {tuple(shape)}" if shape is not None else ""))
    return t


def enc_finalize(cls2, d, anchors):
    """two-stage heads (deformable_transformer_vl.py:503-533): cls2 [T,2] (main | ambiguous logit), d [T,8] (main | ambiguous
    box deltas), anchors [T,4] -> (enc_class [T], enc_coord_unact [T,4], clamped xyxy of sigmoid(enc_coord) [T,4])"""
    _dev(cls2, d, anchors)
    T = cls2.shape[0]
    _f32c(cls2, "cls2", (T, 2)), _f32c(d, "d", (T, 8)), _f32c(anchors, "anchors", (T, 4))
    enc_class = torch.empty((T,), dtype=torch.float32, device=cls2.device)
    enc_coord = torch.empty((T, 4), dtype=torch.float32, device=cls2.device)
    xyxy = torch.empty((T, 4), dtype=torch.float32, device=cls2.device)
    rc = _lib.load().ape_hip_enc_finalize(_p(cls2), _p(d), _p(anchors), T, _p(enc_class), _p(enc_coord), _p(xyxy), _stream())
    _lib.check(rc, "ape_hip_enc_finalize")
    return enc_class, enc_coord, xyxy


def select_proposals(logit, xyxy, level_shapes, pre_nms_topk, num_queries, iou_thr):
    """two-stage proposal selection (deformable_transformer_vl.py:565-627): per-level top-k of sigmoid(logit) (ties: lowest
    index), per-level NMS, per-level quota + fill-up, "naive top-k" fallback -> topk_proposals [num_queries] int64.
    logit [T] fp32, xyxy [T,4] fp32 in [0,1]; seven launches, no host synchronisation."""
    _dev(logit, xyxy)
    T = logit.numel()
    _f32c(logit, "logit", (T,)), _f32c(xyxy, "xyxy", (T, 4))
    starts_c, ns_c, L, starts, ns = _level_arrays(level_shapes)
    if sum(ns) != T:
        raise ValueError("ape_amd.ops.select_proposals: level shapes do not add up to the token count")
    k, k_alt = min(int(pre_nms_topk), T), min(int(num_queries), T)
    n = L * k
    dev = logit.device
    lib = _lib.load()
    cand = torch.empty((n,), dtype=torch.int32, device=dev)
    alt = torch.empty((k_alt,), dtype=torch.int32, device=dev)
    ws = torch.empty((lib.ape_hip_topk_workspace_words(T),), dtype=torch.int64, device=dev)
    _lib.check(lib.ape_hip_proposal_topk(_p(logit), T, starts_c, ns_c, L, k, k_alt, _p(ws), _p(cand), _p(alt), _stream()), "ape_hip_proposal_topk")
    boxes_b = torch.empty((n, 4), dtype=torch.float32, device=dev)
    ints = torch.empty((4 * n + L + 1,), dtype=torch.int32, device=dev)
    groups_b, cand_a, lv_a, pos_b, seg = ints[:n], ints[n:2 * n], ints[2 * n:3 * n], ints[3 * n:4 * n], ints[4 * n:]
    rc = lib.ape_hip_proposal_order(_p(cand), n, _p(logit), _p(xyxy), starts_c, ns_c, L, _p(boxes_b), _p(groups_b), _p(seg), _p(cand_a),
                                    _p(lv_a), _p(pos_b), _stream())
    _lib.check(rc, "ape_hip_proposal_order")
    # static bound on a level's segment: its own top-k plus the zero-score fillers the short levels borrow (all of which may
    # come from one level); lets the scan kernel stage the segment's bit matrix in LDS when it fits
    max_seg = min(n, k + sum(max(0, k - m) for m in ns))
    mask = torch.empty((n, lib.ape_hip_nms_mask_words(n)), dtype=torch.int64, device=dev)
    keep = torch.zeros((n,), dtype=torch.uint8, device=dev)
    _lib.check(lib.ape_hip_nms_mask(_p(boxes_b), _p(groups_b), n, float(iou_thr), _p(mask), _stream()), "ape_hip_nms_mask")
    _lib.check(lib.ape_hip_nms_scan_segments(_p(mask), n, _p(seg), L, int(max_seg), None, _p(keep), _stream()), "ape_hip_nms_scan_segments")
    out = torch.empty((int(num_queries),), dtype=torch.int64, device=dev)
    rc = lib.ape_hip_proposal_quota(_p(cand_a), _p(lv_a), _p(pos_b), _p(keep), n, _p(alt), k_alt, starts_c, ns_c, L, int(num_queries),
                                    _p(out), _stream())
    _lib.check(rc, "ape_hip_proposal_quota")
    return out


def detections(logits, boxes, scale, score_thresh, iou_thr, topk):
    """fast_rcnn_inference_single_image (ape_deta/fast_rcnn.py:97-201) on class-agnostic boxes: logits [Q,K] fp32 (row-major
    view), boxes [Q,4] cxcywh in [0,1], scale [4] device (w,h,w,h) -> dict(det_boxes [k,4], det_scores [k] (-1 = empty slot),
    det_classes [k] int64, det_query [k] int64).  Four launches."""
    _dev(logits, boxes, scale)
    _rowmajor(logits, "logits")
    Q, K = logits.shape
    if logits.dtype != torch.float32:
        raise TypeError("ape_amd.ops.detections: logits must be float32")
    _f32c(boxes, "boxes", (Q, 4)), _f32c(scale, "scale", (4,))
    dev = logits.device
    lib = _lib.load()
    xyxy = torch.empty((Q, 4), dtype=torch.float32, device=dev)
    finite = torch.empty((Q,), dtype=torch.uint8, device=dev)
    sorted_scores = torch.empty((K, Q), dtype=torch.float32, device=dev)
    order = torch.empty((K, Q), dtype=torch.int32, device=dev)
    valid = torch.empty((K, Q), dtype=torch.uint8, device=dev)
    rc = lib.ape_hip_det_sort(_p(logits), _ld(logits), Q, K, _p(boxes), _p(scale), float(score_thresh), _p(xyxy), _p(finite),
                              _p(sorted_scores), _p(order), _p(valid), _stream())
    _lib.check(rc, "ape_hip_det_sort")
    keep = nms_classes(xyxy, order, iou_thr, valid)
    k = min(int(topk), K * Q)
    det = dict(det_boxes=torch.empty((k, 4), dtype=torch.float32, device=dev), det_scores=torch.empty((k,), dtype=torch.float32, device=dev),
               det_classes=torch.empty((k,), dtype=torch.int64, device=dev), det_query=torch.empty((k,), dtype=torch.int64, device=dev))
    ws = torch.empty((lib.ape_hip_topk_workspace_words(K * Q),), dtype=torch.int64, device=dev)
    rc = lib.ape_hip_det_topk(_p(sorted_scores), _p(keep), _p(order), _p(xyxy), K, Q, k, _p(ws), _p(det["det_boxes"]), _p(det["det_scores"]),
                              _p(det["det_classes"]), _p(det["det_query"]), _stream())
    _lib.check(rc, "ape_hip_det_topk")
    return det


def ffn_fused(x, w1, b1, w2, b2, residual=None, out=None, w2_permuted=False, norm=None):
    """residual + relu(x w1^T + b1) w2^T + b2 in one kernel (csrc/ffn_fused.hip); x [M, 256], w1 [HID, 256],
    w2 [256, HID] (w2_permuted: hidden columns in packing.permute_ffn_w2's order) all bf16 or all fp16, biases fp32 -> [M, 256] in x's type.  The hidden
    activations never reach HBM.  norm = (weight [256], bias [256], eps): the result is LayerNorm(residual + ffn(x)) (fp32 statistics of
    the fp32 sums) -- the transformer layer's post-FFN norm in the same launch."""
    _dev(x, w1, w2, b1, b2, residual, out)
    for t, name in ((x, "x"), (w1, "w1"), (w2, "w2")):
        _rowmajor(t, name)
        if t.dtype not in HALF16 or t.dtype != x.dtype:
            raise TypeError(f"ape_amd.ops.ffn_fused: {name} must be bfloat16 or float16, one type per call")
    M, K = x.shape
    HID, N = w1.shape[0], w2.shape[0]
    if w1.shape[1] != K or w2.shape[1] != HID:
        raise ValueError("ape_amd.ops.ffn_fused: weight shapes do not chain")
    if out is None:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    if out.dtype != x.dtype:
        raise TypeError("ape_amd.ops.ffn_fused: out must have x's dtype")
    if residual is not None:
        _rowmajor(residual, "residual")
        if residual.dtype != x.dtype or tuple(residual.shape) != (M, N):
            raise TypeError("ape_amd.ops.ffn_fused: residual must be [M, N] in x's dtype")
    if norm is not None:
        _dev(norm[0], norm[1])
        if norm[0].numel() != N or norm[1].numel() != N:
            raise ValueError("ape_amd.ops.ffn_fused: norm = (weight [N], bias [N], eps)")
    rc = _lib.load().ape_hip_ffn_fused(_p(x), _ld(x), _p(w1), _ld(w1), _p(_f32vec(b1, "b1")), _p(w2), _ld(w2), _p(_f32vec(b2, "b2")),
                                      _p(residual), _ld(residual) if residual is not None else 0, _p(out), _ld(out), M, K, HID, N,
                                      1 if w2_permuted else 0, _dt(x), _p(_f32vec(norm[0], "norm weight")) if norm is not None else None,
                                      _p(_f32vec(norm[1], "norm bias")) if norm is not None else None,
                                      float(norm[2]) if norm is not None else 0.0, _stream())
    _lib.check(rc, "ape_hip_ffn_fused")
    return out
