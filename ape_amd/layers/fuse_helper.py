"""Bi-directional vision<->language attention block -- mirror of ape/layers/fuse_helper.py
(BiMultiHeadAttention :8-166, BiAttentionBlock :178-232): same constructor kwargs and parameter names.

Two device paths share the parameters:

* dense (`forward_tokens_dense`, L > 1: phrase / expression prompts fuse the whole text bank, :356-358) -- the score
  tensor is ONE [T, 8L] matrix S = LN_v(v) U^T with U[(h,l)] = scale * W_v,h^T k_h[l] (the 256->2048 query projection
  over T tokens is folded into the L text tokens), both softmaxes run on S (csrc/softmax.hip), and the two value
  paths are single GEMMs with the per-head output projections folded into their small operand:
      v' = LN_v(v) + softmax_l(S) Z,           Z[(h,l)] = gamma_v * W_ov,h vl_h[l]              (K = 8L)
      l' = LN_l(l) + gamma_l out_l( concat_h( (softmax_t(S)^T LN_v(v))_h W_vv,h^T + b_vv,h ) )   (K = T, split-K)
* single token (`forward_tokens_single`, L = 1: name-prompt mode, deformable_detr_segm_vl.py:349-352 passes one zero
  token).  With L = 1 the softmax over the language axis is identically 1, so
    delta_v = out_v_proj(values_l_proj(LN_l(l)))                              (a per-layer vector)
and the language update is an attention pool over all vision tokens,
    S[t,h]  = scale * (LN_v(v)[t] . (W_v,h^T k_h) + b_v,h . k_h)             ([T,8], one skinny GEMM)
    delta_l = out_l_proj( concat_h( (sum_t softmax_t(S)[t,h] LN_v(v)[t]) W_vv,h^T + b_vv,h ) )
which removes the three 256->2048 projections over 87k tokens (1.66 of the model's 6.95 TFLOP) while keeping
the reference's global-max subtraction, +-5e4 clamps, unmasked padding and residual-on-the-normalised-tensor
(fuse_helper.py:89-111, 224-231).  The reassociation changes summation order only.
"""
import torch
import torch.nn as nn

from .. import ops
from ..packing import attach_cache, f32, pack_matrix


class BiMultiHeadAttention(nn.Module):
    def __init__(self, v_dim, l_dim, embed_dim, num_heads, dropout=0.1, stable_softmax_2d=False,
                 clamp_min_for_underflow=True, clamp_max_for_overflow=True, use_attention_mask_v=False):
        super().__init__()
        self.embed_dim, self.num_heads, self.head_dim = embed_dim, num_heads, embed_dim // num_heads
        self.v_dim, self.l_dim = v_dim, l_dim
        assert self.head_dim * num_heads == embed_dim
        self.scale = self.head_dim ** (-0.5)
        self.dropout = dropout
        self.v_proj = nn.Linear(v_dim, embed_dim)
        self.l_proj = nn.Linear(l_dim, embed_dim)
        self.values_v_proj = nn.Linear(v_dim, embed_dim)
        self.values_l_proj = nn.Linear(l_dim, embed_dim)
        self.out_v_proj = nn.Linear(embed_dim, v_dim)
        self.out_l_proj = nn.Linear(embed_dim, l_dim)
        self.stable_softmax_2d = stable_softmax_2d
        self.clamp_min_for_underflow = clamp_min_for_underflow
        self.clamp_max_for_overflow = clamp_max_for_overflow
        self.use_attention_mask_v = use_attention_mask_v


class BiAttentionBlock(nn.Module):
    def __init__(self, v_dim, l_dim, embed_dim, num_heads, dropout=0.1, drop_path=0.0, init_values=1e-4,
                 stable_softmax_2d=False, clamp_min_for_underflow=True, clamp_max_for_overflow=True,
                 use_attention_mask_v=False):
        super().__init__()
        if num_heads != 8:
            raise ValueError("ape_amd VL fusion kernels are built for 8 heads (APE configs)")
        if not (stable_softmax_2d and clamp_min_for_underflow and clamp_max_for_overflow) or use_attention_mask_v:
            raise ValueError("ape_amd VL fusion implements the APE configuration: stable_softmax_2d + both clamps, "
                             "vision padding unmasked")
        self.layer_norm_v = nn.LayerNorm(v_dim)
        self.layer_norm_l = nn.LayerNorm(l_dim)
        self.attn = BiMultiHeadAttention(v_dim, l_dim, embed_dim, num_heads, dropout, stable_softmax_2d,
                                         clamp_min_for_underflow, clamp_max_for_overflow, use_attention_mask_v)
        self.gamma_v = nn.Parameter(init_values * torch.ones((v_dim)), requires_grad=True)
        self.gamma_l = nn.Parameter(init_values * torch.ones((l_dim)), requires_grad=True)
        attach_cache(self)

    def packed(self, dt):
        def build(dt):
            a = self.attn
            nh, hd, vd = a.num_heads, a.head_dim, a.v_dim
            return dict(
                lnv=(f32(self.layer_norm_v.weight), f32(self.layer_norm_v.bias), self.layer_norm_v.eps),
                lnl=(f32(self.layer_norm_l.weight), f32(self.layer_norm_l.bias), self.layer_norm_l.eps),
                wl=f32(a.l_proj.weight), bl=f32(a.l_proj.bias), wvl=f32(a.values_l_proj.weight), bvl=f32(a.values_l_proj.bias),
                wov=f32(a.out_v_proj.weight), bov=f32(a.out_v_proj.bias), wol=f32(a.out_l_proj.weight), bol=f32(a.out_l_proj.bias),
                wv=f32(a.v_proj.weight).view(nh, hd, vd), bv=f32(a.v_proj.bias).view(nh, hd),
                wvT=f32(a.v_proj.weight).view(nh, hd, vd).transpose(1, 2).contiguous(),            # [h, v_dim, hd]
                bv1=f32(a.v_proj.bias).view(nh, 1, hd).contiguous(),
                wvv=f32(a.values_v_proj.weight).view(nh, hd, vd).contiguous(), bvv=f32(a.values_v_proj.bias).view(nh, hd).contiguous(),
                gv=f32(self.gamma_v), gl=f32(self.gamma_l))
        return self._pack.get(self, dt, build)

    def packed_dense(self, dt):
        def build(_key):
            a = self.attn
            nh, hd, vd = a.num_heads, a.head_dim, a.v_dim
            gv = f32(self.gamma_v)
            wov = f32(a.out_v_proj.weight) * gv[:, None]                         # gamma_v folded into out_v_proj
            return dict(
                wl=pack_matrix(a.l_proj.weight, dt), bl=f32(a.l_proj.bias),
                wvl=pack_matrix(a.values_l_proj.weight, dt), bvl=f32(a.values_l_proj.bias),
                # per head: W_v,h^T  [v_dim, hd]  (W operand of  U_h = k_h W_v,h)
                wvT=(f32(a.v_proj.weight).view(nh, hd, vd).transpose(1, 2) * a.scale).contiguous().to(dt),
                bv=f32(a.v_proj.bias).view(nh, hd),
                # per head: gamma_v * W_ov[:, h]  [v_dim, hd]  (W operand of  Z_h = vl_h W_ov,h^T)
                wov=wov.view(vd, nh, hd).permute(1, 0, 2).contiguous().to(dt), bov=(gv * f32(a.out_v_proj.bias)).contiguous(),
                wvv=f32(a.values_v_proj.weight).view(nh, hd, vd).contiguous().to(dt), bvv=f32(a.values_v_proj.bias).view(nh, hd).contiguous(),
                wol=pack_matrix(a.out_l_proj.weight, dt), bol=f32(a.out_l_proj.bias))
        return self._pack.get(self, ("dense", dt), build)

    def forward_tokens_dense(self, x, lvl_pos, l, dt):
        """x [T,256] vision tokens, l [L, l_dim] fp32 text tokens (L > 1) -> (v_new, v_new + lvl_pos, l_new [L, l_dim]).
        fuse_helper.py:67-166 + 224-231; no language mask (the banks are reduced to one token per phrase, :303)."""
        P, Pd = self.packed(dt), self.packed_dense(dt)
        a = self.attn
        nh, hd, L, T = a.num_heads, a.head_dim, l.shape[0], x.shape[0]
        dev = x.device
        l_n = ops.layernorm(l, P["lnl"][0], P["lnl"][1], P["lnl"][2], out_dtype=torch.float32)
        l_c = l_n.to(dt)
        k = ops.gemm(l_c, Pd["wl"], Pd["bl"], out_dtype=dt)                      # l_proj          [L, E]
        vl = ops.gemm(l_c, Pd["wvl"], Pd["bvl"], out_dtype=dt)                   # values_l_proj   [L, E]
        v_n = ops.layernorm(x, P["lnv"][0], P["lnv"][1], P["lnv"][2], out_dtype=dt)
        U = torch.empty((nh * L, a.v_dim), dtype=dt, device=dev)                 # rows (h, l)
        Zt = torch.empty((a.v_dim, nh * L), dtype=dt, device=dev)                # Z^T: columns (h, l)
        for h in range(nh):
            seg = slice(h * hd, (h + 1) * hd)
            ops.gemm(k[:, seg], Pd["wvT"][h], None, out=U[h * L:(h + 1) * L])
            ops.gemm(vl[:, seg], Pd["wov"][h], None, trans_out=True, out=Zt[:, h * L:(h + 1) * L])
        c = (a.scale * (k.float().view(L, nh, hd) * Pd["bv"][None]).sum(-1)).t().reshape(-1).contiguous()   # b_v,h . k_h[l]
        S = ops.gemm(v_n, U, c, out_dtype=torch.float32)                         # [T, 8L] scores (scale folded into U)
        gmax = S.max().reshape(1)                                                # stable_softmax_2d: one global max (:89-90)
        pv = ops.segment_softmax(S, nh, gmax, dt)                                # softmax over l per head       (:131)
        v_new = ops.gemm(pv, Zt, Pd["bov"], residual=v_n, out_dtype=dt)          # LN_v(v) + gamma_v out_v(...)  (:225-229)
        qp = v_new + lvl_pos
        plT = ops.col_softmax_t(S, gmax, dt, pad=8)                              # softmax over t, [8L, Tp]      (:101-116)
        vT = ops.transpose(v_n, pad=8)                                           # [256, Tp]
        G = ops.gemm(plT, vT, None, out_dtype=dt)                                # sum_t p[(h,l),t] LN_v(v)[t]   [8L, 256]
        ol = torch.empty((L, nh * hd), dtype=dt, device=dev)
        for h in range(nh):
            ops.gemm(G[h * L:(h + 1) * L], Pd["wvv"][h], Pd["bvv"][h], out=ol[:, h * hd:(h + 1) * hd])   # values_v_proj
        dl = ops.gemm(ol, Pd["wol"], Pd["bol"], out_dtype=torch.float32)
        l_new = l_n + P["gl"] * dl
        return v_new, qp, l_new

    def forward_tokens(self, x, lvl_pos, l, dt, defer_language=False):
        """-> (v_new, v_new + lvl_pos, l_new).  defer_language: l_new comes back as a handle whose .join() yields it -- in the
        single-token path the language update only feeds the NEXT layer's fusion, so it runs as a parallel graph branch
        next to this layer's deformable attention / FFN."""
        if l.shape[0] == 1:
            v_new, qp, ljob = self.forward_tokens_single(x, lvl_pos, l, dt)
        else:
            v_new, qp, l_new = self.forward_tokens_dense(x, lvl_pos, l, dt)
            ljob = ops._Joined(l_new)
        return (v_new, qp, ljob) if defer_language else (v_new, qp, ljob.join())

    def forward_tokens_single(self, x, lvl_pos, l, dt):
        """x [T,256] vision tokens, lvl_pos [T,256], l [1, l_dim] fp32 ->
        (v_new [T,256], v_new + lvl_pos [T,256], handle of l_new [1, l_dim])"""
        P = self.packed(dt)
        a = self.attn
        l_n = ops.layernorm(l, P["lnl"][0], P["lnl"][1], P["lnl"][2], out_dtype=torch.float32)
        k = ops.gemv(l_n, P["wl"], P["bl"])                      # l_proj            [1, E]
        vl = ops.gemv(l_n, P["wvl"], P["bvl"])                   # values_l_proj     [1, E]
        # out_v_proj (softmax over one token == 1) with gamma_v and the LayerNorm bias in the epilogue:
        # gdv = gamma_v * delta_v [1, v_dim], lnb = LN_v bias + gdv
        gdv, lnb = ops.gemv(vl, P["wov"], P["bov"], scale=P["gv"], add=P["lnv"][1][None, :])
        v_new, qp = ops.layernorm(x, P["lnv"][0], lnb[0], P["lnv"][2], out_dtype=dt, add=lvl_pos)

        def language_side():
            kh = k.view(a.num_heads, a.head_dim)
            u, u_c = ops.head_gemv(kh, P["wvT"], bf16_copy=dt if dt in ops.HALF16 else True)     # W_v,h^T k_h       [8, v_dim] (+ the GEMM operand copy)
            c = ops.head_gemv(kh, P["bv1"], alpha=a.scale)           # scale * b_v,h . k_h   [8, 1]
            # scores are taken on LN_v(v) = v_new - gdv: bias_h = scale * (c_h - u_h . gamma_v delta_v)
            _, sbias = ops.gemv(gdv, u, alpha=-a.scale, add=c.view(1, -1))
            S = ops.gemm(v_new, u_c if dt in ops.HALF16 else u, sbias[0], alpha=a.scale, out_dtype=torch.float32)   # [T, 8]
            pooled = ops.vl_pool(S, v_new, gdv[0])                   # sum_t p[t,h] LN_v(v)[t]     [8, v_dim]
            ol = ops.head_gemv(pooled, P["wvv"], P["bvv"])           # values_v_proj per head   [8, hd]
            return ops.gemv(ol.view(1, -1), P["wol"], P["bol"], scale=P["gl"], add=l_n)[1]       # l_n + gamma_l * delta_l

        return v_new, qp, ops.fork(language_side)

    def forward(self, v, l, attention_mask_v=None, attention_mask_l=None):
        """reference signature (fuse_helper.py:221-232); the text bank carries no mask (reduced tokens, :266/:303)"""
        if attention_mask_l is not None:
            raise NotImplementedError("ape_amd BiAttentionBlock: per-token language masks (un-reduced expression tokens, "
                                      "text_feature_reduce_before_fusion=False) are not implemented")
        dt = getattr(self, "compute_dtype", torch.bfloat16)
        vs, ls = [], []
        for b in range(v.shape[0]):
            zeros = torch.zeros_like(v[b], dtype=dt)
            vn, _, ln = self.forward_tokens(v[b].to(dt).contiguous(), zeros, l[b].float().contiguous(), dt)
            vs.append(vn.to(v.dtype))
            ls.append(ln.to(l.dtype))
        return torch.stack(vs), torch.stack(ls)
