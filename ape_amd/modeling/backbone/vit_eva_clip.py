"""EVA-02-CLIP ViT backbone + SimpleFeaturePyramid on the HIP kernels.

Host-side mirror of ape/modeling/backbone/vit_eva_clip.py (ViT :570-754, Block :505-523, Attention :218-268,
SwiGLU :125-132, SimpleFeaturePyramid :757-922) and the helpers it uses from utils_eva02.py (PatchEmbed :190-216,
get_abs_pos :158-187, window_partition :19-63, VisionRotaryEmbeddingFast :307-346): same class names, constructor
kwargs and state-dict keys, so the reference's LazyConfig (configs/common/backbone/vitl_eva02_clip.py) instantiates
it unchanged.  The arithmetic runs through ape_amd.ops (C-ABI -> HIP); there is no PyTorch fallback.

MI355X-first layout: tokens stay token-major [N, C] and in WINDOW-MAJOR order through all blocks, so window
partition / unpartition are free (global blocks just use a RoPE table permuted to the same order); q|k are one
GEMM with the RoPE rotation fused in its epilogue, V is produced transposed for the flash-attention kernel,
w1|w2 are one interleaved GEMM with the SwiGLU epilogue, and the FPN's deconvolutions / 1x1 / 3x3 convolutions are
GEMMs over token-major maps (the pixel-shuffle and window orders are folded into the 3x3 im2col gather).
"""
import math
from functools import partial

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...packing import attach_cache, f32, pack_matrix, round_up
from ...stagetap import tap

__all__ = ["ViT", "SimpleFeaturePyramid"]


class VisionRotaryEmbeddingFast(nn.Module):
    """Persistent cos/sin tables [ft*ft, 2*dim] (utils_eva02.py:307-344); applied inside the q|k GEMM epilogue."""

    def __init__(self, dim, pt_seq_len=16, ft_seq_len=None, theta=10000):
        super().__init__()
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: (dim // 2)].float() / dim))
        if ft_seq_len is None:
            ft_seq_len = pt_seq_len
        t = torch.arange(ft_seq_len) / ft_seq_len * pt_seq_len
        freqs = (t[:, None] * freqs[None, :]).repeat_interleave(2, dim=-1)
        fh = freqs[:, None, :].expand(ft_seq_len, ft_seq_len, dim)
        fw = freqs[None, :, :].expand(ft_seq_len, ft_seq_len, dim)
        full = torch.cat((fh, fw), dim=-1)
        self.register_buffer("freqs_cos", full.cos().reshape(-1, 2 * dim))
        self.register_buffer("freqs_sin", full.sin().reshape(-1, 2 * dim))


class PatchEmbed(nn.Module):
    def __init__(self, kernel_size=(16, 16), stride=(16, 16), padding=(0, 0), in_chans=3, embed_dim=768):
        super().__init__()
        assert tuple(kernel_size) == (16, 16) and tuple(stride) == (16, 16) and in_chans == 3, "HIP patchify is 16x16/16, RGB"
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=kernel_size, stride=stride, padding=padding)


class SwiGLU(nn.Module):
    def __init__(self, in_features, hidden_features, norm_layer, subln=True):
        super().__init__()
        self.w1 = nn.Linear(in_features, hidden_features)
        self.w2 = nn.Linear(in_features, hidden_features)
        self.ffn_ln = norm_layer(hidden_features) if subln else nn.Identity()
        self.w3 = nn.Linear(hidden_features, in_features)


class Mlp(nn.Module):
    """fc1 -> GELU -> (ffn_ln) -> fc2 (vit_eva_clip.py:67-98): the MLP of the non-SwiGLU configurations (ViT-e)"""

    def __init__(self, in_features, hidden_features, norm_layer, subln=False):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.ffn_ln = norm_layer(hidden_features) if subln else nn.Identity()
        self.fc2 = nn.Linear(hidden_features, in_features)


def padded_head_dim(hd):
    """the flash-attention kernel's head widths: 32 / 64 / 128; narrower heads are zero-padded by the weight packing (zero q / k
    columns add nothing to the scores, zero v columns produce zero outputs that meet zero columns of the out projection)"""
    for w in (32, 64, 128):
        if hd <= w:
            return w
    raise ValueError(f"ape_amd ViT: head width {hd} > 128")


class Attention(nn.Module):
    def __init__(self, dim, num_heads, rope, norm_layer, subln=True, qkv_bias=True):
        super().__init__()
        self.num_heads = num_heads
        self.subln = subln
        if subln:
            self.q_proj = nn.Linear(dim, dim, bias=False)
            self.k_proj = nn.Linear(dim, dim, bias=False)
            self.v_proj = nn.Linear(dim, dim, bias=False)
        else:
            self.qkv = nn.Linear(dim, dim * 3, bias=False)
        if qkv_bias:
            self.q_bias = nn.Parameter(torch.zeros(dim))
            self.v_bias = nn.Parameter(torch.zeros(dim))
        else:
            self.q_bias = self.v_bias = None
        self.inner_attn_ln = norm_layer(dim) if subln else nn.Identity()
        self.proj = nn.Linear(dim, dim)
        self.rope = rope

    def qkv_weights(self):
        """(Wq, Wk, Wv [E, E], q bias, v bias [E] or None) of either parameterisation (vit_eva_clip.py:236-260)"""
        if self.subln:
            wq, wk, wv = self.q_proj.weight, self.k_proj.weight, self.v_proj.weight
        else:
            E = self.qkv.weight.shape[1]
            wq, wk, wv = self.qkv.weight[:E], self.qkv.weight[E:2 * E], self.qkv.weight[2 * E:]
        return wq.detach().float(), wk.detach().float(), wv.detach().float(), self.q_bias, self.v_bias


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, norm_layer, window_size, rope, postnorm=False, subln=True, naiveswiglu=True,
                 qkv_bias=True):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads, rope, norm_layer, subln=subln, qkv_bias=qkv_bias)
        self.norm2 = norm_layer(dim)
        hidden = int(dim * mlp_ratio)
        self.mlp = SwiGLU(dim, hidden, norm_layer, subln=subln) if naiveswiglu else Mlp(dim, hidden, norm_layer, subln=subln)
        self.window_size = window_size
        self.postnorm, self.subln, self.naiveswiglu = postnorm, subln, naiveswiglu
        attach_cache(self)

    def packed(self, dt):
        def build(dt):
            a, m = self.attn, self.mlp
            wq, wk, wv, qb, vb = a.qkv_weights()
            E = wq.shape[1]
            nh = a.num_heads
            hd = E // nh
            hdp = padded_head_dim(hd)
            dev = wq.device

            def pad_heads(w):                      # [nh * hd, ...] -> [nh * hdp, ...], zero rows behind every head
                if hdp == hd:
                    return w
                out = torch.zeros((nh, hdp) + tuple(w.shape[1:]), dtype=torch.float32, device=dev)
                out[:, :hd] = w.reshape((nh, hd) + tuple(w.shape[1:]))
                return out.reshape((nh * hdp,) + tuple(w.shape[1:]))

            zeros = torch.zeros(E, dtype=torch.float32, device=dev)
            qb = qb.detach().float() if qb is not None else zeros
            vb = vb.detach().float() if vb is not None else zeros
            wproj = a.proj.weight.detach().float()
            if hdp != hd:                          # zero COLUMNS of the out projection where the attention output is padding
                wproj = pad_heads(wproj.t().contiguous()).t().contiguous()
            P = dict(
                hd=hd, hdp=hdp, Ep=nh * hdp,
                wqk=pack_matrix(torch.cat([pad_heads(wq), pad_heads(wk)], 0), dt),
                bqk=torch.cat([pad_heads(qb), torch.zeros(nh * hdp, dtype=torch.float32, device=dev)]).contiguous(),
                wv=pack_matrix(pad_heads(wv), dt), bv=pad_heads(vb).contiguous(),
                wproj=pack_matrix(wproj, dt), bproj=f32(a.proj.bias),
                n1=(f32(self.norm1.weight), f32(self.norm1.bias), self.norm1.eps),
                n2=(f32(self.norm2.weight), f32(self.norm2.bias), self.norm2.eps))
            if self.subln:
                P["nin"] = (f32(a.inner_attn_ln.weight), f32(a.inner_attn_ln.bias), a.inner_attn_ln.eps)
                P["nffn"] = (f32(m.ffn_ln.weight), f32(m.ffn_ln.bias), m.ffn_ln.eps)
                P.update(self._folded_inner_ln(a, dt, hd == hdp))
            if self.naiveswiglu:
                hid = m.w1.weight.shape[0]
                hid_pad = round_up(hid, 64)
                n12 = 2 * hid_pad          # zero (gate, up) rows beyond 2*hid: the fused SwiGLU epilogue writes exact zeros into the
                w12 = torch.zeros((n12, E), dtype=torch.float32, device=dev)   # K padding of the down projection
                w12[0:2 * hid:2] = m.w1.weight.detach().float()
                w12[1:2 * hid:2] = m.w2.weight.detach().float()
                b12 = torch.zeros((n12,), dtype=torch.float32, device=dev)
                b12[0:2 * hid:2] = m.w1.bias.detach().float()
                b12[1:2 * hid:2] = m.w2.bias.detach().float()
                P.update(w12=pack_matrix(w12, dt), b12=b12, hid=hid, hid_pad=hid_pad,
                         w3=pack_matrix(m.w3.weight, dt, kpad=64), b3=f32(m.w3.bias), **self._folded_subln(m, dt))
            else:
                hid = m.fc1.weight.shape[0]
                hid_pad = round_up(hid, 64)
                w1 = torch.zeros((hid_pad, E), dtype=torch.float32, device=dev)     # zero rows: gelu(0) = 0 exactly in the K padding
                w1[:hid] = m.fc1.weight.detach().float()
                b1 = torch.zeros((hid_pad,), dtype=torch.float32, device=dev)
                b1[:hid] = m.fc1.bias.detach().float()
                P.update(w1=pack_matrix(w1, dt), b1=b1, hid=hid, hid_pad=hid_pad,
                         w2=pack_matrix(m.fc2.weight, dt, kpad=64), b2=f32(m.fc2.bias))
            return P
        return self._pack.get(self, dt, build)

    @staticmethod
    def _folded_subln(m, dt):
        """bf16 production mode: the SwiGLU sub-LayerNorm (vit_eva_clip.py:129-131) is folded into the down projection,
        LN(h) W3^T = rstd (h W'^T) - rstd mean rowsum(W') + W3 b_ln  with  W' = W3 diag(gamma)  (ApeGemmArgs.rowscale ...):
        the 2730-wide activation is read once (statistics) instead of read + written + read again."""
        if dt not in ops.HALF16 or not isinstance(m.ffn_ln, nn.LayerNorm) or os.environ.get("APE_NO_LNFOLD") == "1":
            return {}
        w3 = m.w3.weight.detach().float()
        g, b = m.ffn_ln.weight.detach().float(), m.ffn_ln.bias.detach().float()
        w3f = pack_matrix(w3 * g[None, :], dt, kpad=64)
        return dict(w3f=w3f, c1=w3f.float().sum(dim=1).contiguous(),                 # row sums of the ROUNDED folded weight
                    c2=(w3 @ b + m.w3.bias.detach().float()).contiguous())

    @staticmethod
    def _folded_inner_ln(a, dt, unpadded):
        """16-bit production mode (round 6): the attention's inner LayerNorm (vit_eva_clip.py:258-262, `inner_attn_ln` between the
        attention and its output projection) folded into that projection exactly like the SwiGLU sub-LayerNorm below:
        LN(o) Wp^T + bp = rstd (o W'^T) - rstd mean rowsum(W') + (Wp b_ln + bp)  with  W' = Wp diag(gamma).  The row statistics of the
        stored attention output come from the projection's own launch (ops.gemm rowstats=): no LayerNorm launch, no second copy of o."""
        if dt not in ops.HALF16 or not unpadded or not isinstance(a.inner_attn_ln, nn.LayerNorm) or os.environ.get("APE_NO_LNFOLD") == "1":
            return {}
        wp = a.proj.weight.detach().float()
        g, b = a.inner_attn_ln.weight.detach().float(), a.inner_attn_ln.bias.detach().float()
        wpf = pack_matrix(wp * g[None, :], dt)
        return dict(wprojf=wpf, cp1=wpf.float().sum(dim=1).contiguous(),                 # row sums of the ROUNDED folded weight
                    cp2=(wp @ b + a.proj.bias.detach().float()).contiguous())

    def _attention(self, xn, P, rope, nwin, ntok_win, vt_buf, images):
        """the attention branch on the normalised (pre-norm) or raw (post-norm) tokens xn [rows, E] -> [rows, Ep]"""
        nh = self.attn.num_heads
        hd, hdp, Ep = P["hd"], P["hdp"], P["Ep"]
        # V^T on a parallel graph branch: two M = 4096 GEMMs fill the chip better together than one after the other
        vjob = ops.fork(lambda: ops.gemm(xn, P["wv"], P["bv"], trans_out=True, out=vt_buf))
        if rope is not None:
            qk = ops.gemm(xn, P["wqk"], P["bqk"], rope=(rope[0], rope[1], rope[2], hd, 2 * Ep) + tuple(rope[3:]))
        else:
            qk = ops.gemm(xn, P["wqk"], P["bqk"])
        vt = vjob.join()
        if self.window_size > 0:
            o = ops.attention(qk[:, :Ep], qk[:, Ep:], vt, batch=nwin, n=ntok_win, heads=nh, head_dim=hdp, scale=hd ** -0.5)
        else:
            o = ops.attention(qk[:, :Ep], qk[:, Ep:], vt, batch=images, n=xn.shape[0] // images, heads=nh, head_dim=hdp, scale=hd ** -0.5)
        if self.subln and "wprojf" not in P:
            o = ops.layernorm(o, P["nin"][0], P["nin"][1], P["nin"][2], out_dtype=xn.dtype)
        return o

    def _out_proj(self, o, P, residual, out_dtype):
        """the attention's output projection (+ residual), with the inner LayerNorm folded in when packed so"""
        if "wprojf" in P:
            return ops.gemm(o, P["wprojf"], P["cp2"], residual=residual, out_dtype=out_dtype, rowstats=(o.shape[1], P["nin"][2], P["cp1"]))
        return ops.gemm(o, P["wproj"], P["bproj"], residual=residual, out_dtype=out_dtype)

    def _mlp(self, xn, P, dt, residual, out_dtype):
        """the MLP branch: residual + mlp(xn) (residual None: mlp(xn) alone) in out_dtype"""
        hbuf = torch.empty((xn.shape[0], P["hid_pad"]), dtype=dt, device=xn.device)
        if self.naiveswiglu:
            ops.gemm(xn, P["w12"], P["b12"], act=ops.ACT_SWIGLU, out=hbuf)
            if "w3f" in P:
                # the row statistics of the stored hidden activation come from the down projection's own launch where the 256 x 128 tile
                # kernel runs it (ops.gemm rowstats=; else a row_stats launch as in rounds 3-5): the K padding of hbuf is exact zeros
                return ops.gemm(hbuf, P["w3f"], P["c2"], residual=residual, rowstats=(P["hid"], P["nffn"][2], P["c1"]), out_dtype=out_dtype)
            if self.subln:
                hbuf = ops.layernorm(hbuf[:, :P["hid"]], P["nffn"][0], P["nffn"][1], P["nffn"][2], out_dtype=dt, cpad=P["hid_pad"])
            return ops.gemm(hbuf, P["w3"], P["b3"], residual=residual, out_dtype=out_dtype)
        ops.gemm(xn, P["w1"], P["b1"], act=ops.ACT_GELU, out=hbuf)
        if self.subln:
            hbuf = ops.layernorm(hbuf[:, :P["hid"]], P["nffn"][0], P["nffn"][1], P["nffn"][2], out_dtype=dt, cpad=P["hid_pad"])
        return ops.gemm(hbuf, P["w2"], P["b2"], residual=residual, out_dtype=out_dtype)

    def forward_tokens(self, x, dt, rope, nwin, ntok_win, vt_buf, last=False, images=1):
        """pre-norm block (vit_eva_clip.py:519-523): x [images * N, E] fp32 residual stream (window-major per image).
        rope = (cos, sin, rows) or None; nwin = windows of ALL images.  Returns the new stream."""
        P = self.packed(dt)
        xn = ops.layernorm(x, P["n1"][0], P["n1"][1], P["n1"][2], out_dtype=dt)
        o = self._attention(xn, P, rope, nwin, ntok_win, vt_buf, images)
        x = self._out_proj(o, P, x, torch.float32)
        xn = ops.layernorm(x, P["n2"][0], P["n2"][1], P["n2"][2], out_dtype=dt)
        return self._mlp(xn, P, dt, x, dt if last else torch.float32)

    def forward_tokens_postnorm(self, x32, xb, dt, rope, nwin, ntok_win, vt_buf, images=1):
        """post-norm block (vit_eva_clip.py:505-517): x += norm1(attn(x)); x += norm2(mlp(x)).  x32: the fp32 residual stream,
        updated IN PLACE; xb: the same values in the GEMM operand type (x32 itself in fp32 mode).  Returns the new xb."""
        P = self.packed(dt)
        cdt = None if dt == torch.float32 else dt
        o = self._attention(xb, P, rope, nwin, ntok_win, vt_buf, images)
        t = self._out_proj(o, P, None, torch.float32)
        xb = ops.postnorm_residual(x32, t, P["n1"], copy_dtype=cdt)
        xb = x32 if xb is None else xb
        t = self._mlp(xb, P, dt, None, torch.float32)
        xb = ops.postnorm_residual(x32, t, P["n2"], copy_dtype=cdt)
        return x32 if xb is None else xb


def _with_packed(cos, sin, rows):
    """(cos, sin, rows[, packed]) -- packed = [rows, hd / 2, 2] (cos, sin) per rotate_half pair when both columns of every pair share
    an angle (VisionRotaryEmbeddingFast repeats each frequency twice, vit_eva_clip.py:179-216): the form the 256-row GEMM tile stages
    through LDS (ApeGemmArgs.rope_cs); checked, not assumed"""
    if torch.equal(cos[:, 0::2], cos[:, 1::2]) and torch.equal(sin[:, 0::2], sin[:, 1::2]):
        return (cos, sin, rows, torch.stack([cos[:, 0::2], sin[:, 0::2]], dim=-1).contiguous())
    return (cos, sin, rows)


def window_major_order(ht, wt, ws):
    """token order used inside the backbone: (window row, window col, row in window, col in window).
    returns (tok2raster int32 [N], raster2tok int32 [N])"""
    r = torch.arange(ht * wt).view(ht // ws, ws, wt // ws, ws).permute(0, 2, 1, 3).reshape(-1)
    inv = torch.empty_like(r)
    inv[r] = torch.arange(ht * wt)
    return r.to(torch.int32), inv.to(torch.int32)


class Backbone(nn.Module):
    """minimal stand-in for detectron2.modeling.backbone.Backbone (output_shape / size_divisibility contract)"""

    @property
    def size_divisibility(self):
        return 0

    @property
    def padding_constraints(self):
        return {}

    def output_shape(self):
        from types import SimpleNamespace
        return {n: SimpleNamespace(channels=self._out_feature_channels[n], stride=self._out_feature_strides[n])
                for n in self._out_features}


class ViT(Backbone):
    def __init__(self, img_size=1024, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0,
                 qkv_bias=False, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0,
                 norm_layer=partial(nn.LayerNorm, eps=1e-6), init_values=None, use_abs_pos=True, use_rel_pos=False,
                 rope=False, postnorm=False, pt_hw_seq_len=16, intp_freq=False, naiveswiglu=False, subln=False,
                 window_size=0, window_block_indexes=(), residual_block_indexes=(), use_act_checkpoint=False,
                 pretrain_img_size=224, pretrain_use_cls_token=True, out_feature="last_feat", xattn=False, frozen_stages=-1):
        super().__init__()
        # the HIP path implements the EVA-02-CLIP configurations of the APE configs: rope + sub-LN + SwiGLU + pre-norm (ViT-L,
        # vitl_eva02_clip.py:9-48) and packed-qkv + GELU MLP + post-norm without rope (ViT-e, vite_eva02_clip_1024.py:9-49)
        assert use_abs_pos and init_values is None, "ape_amd ViT: absolute position embedding, no layer scale (every APE config)"
        assert len(residual_block_indexes) == 0 and patch_size == 16
        assert not rope or intp_freq, "ape_amd ViT: rope tables are interpolated to the token grid (intp_freq=True) in the APE configs"
        assert (img_size // patch_size) % window_size == 0, "token grid must be a multiple of the window size"
        assert embed_dim <= 2048 or not postnorm, "ape_amd ViT: the post-norm residual kernel holds rows of <= 2048 channels"
        self.pretrain_use_cls_token = pretrain_use_cls_token
        self.img_size, self.patch_size, self.embed_dim, self.window_size = img_size, patch_size, embed_dim, window_size
        self.postnorm = postnorm
        self.patch_embed = PatchEmbed(in_chans=in_chans, embed_dim=embed_dim)
        num_patches = (pretrain_img_size // patch_size) ** 2
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + (1 if pretrain_use_cls_token else 0), embed_dim))
        half_head_dim = embed_dim // num_heads // 2
        hw = img_size // patch_size
        if rope:
            assert padded_head_dim(2 * half_head_dim) == 2 * half_head_dim, "ape_amd ViT: rope needs a head width of 32 / 64 / 128"
            self.rope_win = VisionRotaryEmbeddingFast(half_head_dim, pt_seq_len=pt_hw_seq_len, ft_seq_len=window_size)
            self.rope_glb = VisionRotaryEmbeddingFast(half_head_dim, pt_seq_len=pt_hw_seq_len, ft_seq_len=hw)
        else:
            self.rope_win = self.rope_glb = None
        self.blocks = nn.ModuleList([
            Block(embed_dim, num_heads, mlp_ratio, norm_layer, window_size if i in window_block_indexes else 0,
                  self.rope_win if i in window_block_indexes else self.rope_glb, postnorm=postnorm, subln=subln,
                  naiveswiglu=naiveswiglu, qkv_bias=qkv_bias) for i in range(depth)])
        self._out_feature_channels = {out_feature: embed_dim}
        self._out_feature_strides = {out_feature: patch_size}
        self._out_features = [out_feature]
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        self.compute_dtype = torch.bfloat16
        attach_cache(self)

    def token_order(self, hw):
        """(tok2raster, raster2tok) of the feature this ViT hands to the pyramid: window-major order"""
        return window_major_order(hw, hw, self.window_size)

    def packed(self, dt):
        def build(dt):
            hw = self.img_size // self.patch_size
            dev = self.pos_embed.device
            t2r, r2t = window_major_order(hw, hw, self.window_size)
            t2r, r2t = t2r.to(dev), r2t.to(dev)
            # get_abs_pos (utils_eva02.py:158-187): drop cls, bicubic resize to the token grid -- a per-model constant
            pos = self.pos_embed.detach().float()
            if self.pretrain_use_cls_token:
                pos = pos[:, 1:]
            size = int(math.sqrt(pos.shape[1]))
            if size != hw:
                pos = F.interpolate(pos.reshape(1, size, size, -1).permute(0, 3, 1, 2), size=(hw, hw), mode="bicubic",
                                    align_corners=False).permute(0, 2, 3, 1)
            pos = pos.reshape(hw * hw, -1)[t2r.long()].contiguous()
            w = self.patch_embed.proj.weight
            return dict(
                hw=hw, t2r=t2r, r2t=r2t, pos=pos, wpe=pack_matrix(w.reshape(w.shape[0], -1), dt), bpe=f32(self.patch_embed.proj.bias),
                rope_win=None if self.rope_win is None else _with_packed(f32(self.rope_win.freqs_cos), f32(self.rope_win.freqs_sin), self.window_size ** 2),
                rope_glb=None if self.rope_glb is None else _with_packed(f32(self.rope_glb.freqs_cos)[t2r.long()].contiguous(),
                                                                         f32(self.rope_glb.freqs_sin)[t2r.long()].contiguous(), hw * hw),
            )
        return self._pack.get(self, dt, build)

    def forward_tokens(self, image, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0), stages=None):
        """image [3,h,w] fp32 (h,w <= img_size) -> last feature [N, E] in the compute dtype, WINDOW-MAJOR token order.
        stages: optional dict / StageTap (stagetap.py) recording "vit_embed" and every block's output "vit_blk<i>".
        A list of B images (sizes may differ) runs as ONE pass over [B * N, E]: every linear sees B x 4096 rows (the big-tile
        GEMM kernels need that many to fill 256 CUs), windows / global attention stay per image.  Rows are independent in
        every op of the ViT, so each image's result is bit-identical to its own single-image pass."""
        dt = self.compute_dtype
        P = self.packed(dt)
        hw = P["hw"]
        n = hw * hw
        images = list(image) if isinstance(image, (list, tuple)) else [image]
        B = len(images)
        if B == 1:
            patches = ops.patchify(images[0], P["t2r"], hw, hw, mean, std, out_dtype=dt)
            pos = P["pos"]
        else:
            patches = torch.empty((B * n, 768), dtype=dt, device=images[0].device)
            for b, im in enumerate(images):
                ops.patchify(im, P["t2r"], hw, hw, mean, std, out_dtype=dt, out=patches[b * n:(b + 1) * n])
            if ("pos", B) not in P:
                P[("pos", B)] = P["pos"].repeat(B, 1).contiguous()
            pos = P[("pos", B)]
        x = tap(stages, "vit_embed", ops.gemm(patches, P["wpe"], P["bpe"], residual=pos, out_dtype=torch.float32))
        nwin = (hw // self.window_size) ** 2 * B
        Ep = self.blocks[0].packed(dt)["Ep"]
        vt_buf = ops.zeros((Ep, round_up(B * n, 64)), dt, x.device)
        if self.postnorm:
            # the fp32 stream is updated IN PLACE by the post-norm residual kernel: with taps, work on a private copy (a forced
            # tensor belongs to the teacher) and record copies
            x32 = x.clone() if stages is not None else x
            cdt = None if dt == torch.float32 else dt
            xb = x32 if cdt is None else ops.postnorm_residual(x32, None, None, copy_dtype=cdt)
            for i, blk in enumerate(self.blocks):
                rope = P["rope_win"] if blk.window_size > 0 else P["rope_glb"]
                xb = blk.forward_tokens_postnorm(x32, xb, dt, rope, nwin, self.window_size ** 2, vt_buf, images=B)
                if stages is not None:
                    rec = x32.clone()
                    forced = tap(stages, f"vit_blk{i}", rec)
                    if forced is not rec:                            # teacher forcing: the next block starts from the teacher's stream
                        x32 = forced.float().clone()
                        xb = x32 if cdt is None else ops.postnorm_residual(x32, None, None, copy_dtype=cdt)
            return xb
        for i, blk in enumerate(self.blocks):
            rope = P["rope_win"] if blk.window_size > 0 else P["rope_glb"]
            x = blk.forward_tokens(x, dt, rope, nwin, self.window_size ** 2, vt_buf, last=(i == len(self.blocks) - 1), images=B)
            x = tap(stages, f"vit_blk{i}", x)
        return x

    def forward(self, x):
        """reference signature: normalised, padded NCHW batch -> {"last_feat": [B, E, h/16, w/16]}"""
        outs = []
        P = self.packed(self.compute_dtype)
        for b in range(x.shape[0]):
            t = self.forward_tokens(x[b].float().contiguous())
            outs.append(t[P["r2t"].long()].float().reshape(P["hw"], P["hw"], -1).permute(2, 0, 1))
        return {self._out_features[0]: torch.stack(outs)}


class _LN2d(nn.Module):
    """parameter holder for detectron2's channel LayerNorm (get_norm("LN"), eps 1e-6)"""

    def __init__(self, c, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.eps = eps


class _ConvLN(nn.Module):
    """detectron2 Conv2d(bias=False, norm=LN): holds weight [Cout,Cin,k,k] and norm.{weight,bias}"""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        self.norm = _LN2d(cout)
        self.k = k


class LastLevelMaxPool(nn.Module):
    """detectron2 LastLevelMaxPool: p6 = p5[:, :, ::2, ::2] (max_pool2d kernel 1, stride 2)"""

    def __init__(self):
        super().__init__()
        self.num_levels = 1
        self.in_feature = "p5"


class SimpleFeaturePyramid(Backbone):
    def __init__(self, net, in_feature, out_channels, scale_factors, top_block=None, norm="LN", square_pad=0):
        super().__init__()
        assert tuple(scale_factors) == (4.0, 2.0, 1.0, 0.5) and norm == "LN", "ape_amd FPN: the ViTDet (4,2,1,0.5)/LN pyramid"
        self.scale_factors = scale_factors
        dim = net.embed_dim
        self.simfp_2 = nn.Sequential(nn.ConvTranspose2d(dim, dim // 2, 2, 2), _LN2d(dim // 2), nn.GELU(),
                                     nn.ConvTranspose2d(dim // 2, dim // 4, 2, 2), _ConvLN(dim // 4, out_channels, 1),
                                     _ConvLN(out_channels, out_channels, 3))
        self.simfp_3 = nn.Sequential(nn.ConvTranspose2d(dim, dim // 2, 2, 2), _ConvLN(dim // 2, out_channels, 1),
                                     _ConvLN(out_channels, out_channels, 3))
        self.simfp_4 = nn.Sequential(_ConvLN(dim, out_channels, 1), _ConvLN(out_channels, out_channels, 3))
        self.simfp_5 = nn.Sequential(nn.MaxPool2d(2, 2), _ConvLN(dim, out_channels, 1), _ConvLN(out_channels, out_channels, 3))
        self.net = net
        self.in_feature = in_feature
        self.top_block = top_block
        strides = [int(net.patch_size / s) for s in scale_factors]
        self._out_feature_strides = {f"p{int(math.log2(s))}": s for s in strides}
        if top_block is not None:
            last = int(math.log2(strides[-1]))
            self._out_feature_strides[f"p{last + 1}"] = 2 ** (last + 1)
        self._out_features = list(self._out_feature_strides.keys())
        self._out_feature_channels = {k: out_channels for k in self._out_features}
        self._size_divisibility = strides[-1]
        self._square_pad = square_pad
        self.out_channels = out_channels
        attach_cache(self)

    @property
    def padding_constraints(self):
        return {"size_divisiblity": self._size_divisibility, "square_size": self._square_pad}  # sic (vit_eva_clip.py:864-869)

    @property
    def compute_dtype(self):
        return self.net.compute_dtype

    @staticmethod
    def _deconv_matrix(m, dt):
        """ConvTranspose2d(k=2,s=2) weight [Cin,Cout,2,2] -> GEMM weight [(i*2+j)*Cout + co, ci], bias x4"""
        w = m.weight.detach().float()
        cin, cout = w.shape[0], w.shape[1]
        return pack_matrix(w.permute(2, 3, 1, 0).reshape(4 * cout, cin), dt), m.bias.detach().float().repeat(4).contiguous()

    @staticmethod
    def _conv_matrix(m, dt):
        """Conv2d weight [Cout,Cin,k,k] -> [Cout, (ky*k+kx)*Cin + ci] (matches ops.im2col3x3's column order)"""
        w = m.weight.detach().float()
        return pack_matrix(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1), dt)

    def packed(self, dt):
        def build(dt):
            hw = self.net.img_size // self.net.patch_size
            dev = self.simfp_4[0].weight.device
            _, r2t = self.net.token_order(hw)          # vit_eva_clip: window-major, vit_eva02: raster
            r2t = r2t.long()
            # raster (Y,X) of the x2 / x4 maps -> row of the nested (token, i, j[, i2, j2]) layouts the deconv GEMMs produce
            Y2, X2 = torch.meshgrid(torch.arange(2 * hw), torch.arange(2 * hw), indexing="ij")
            p3 = (r2t[(Y2 // 2) * hw + X2 // 2] * 4 + (Y2 % 2) * 2 + X2 % 2).reshape(-1)
            Y4, X4 = torch.meshgrid(torch.arange(4 * hw), torch.arange(4 * hw), indexing="ij")
            p2 = ((r2t[(Y4 // 4) * hw + X4 // 4] * 4 + ((Y4 // 2) % 2) * 2 + (X4 // 2) % 2) * 4 + (Y4 % 2) * 2 + X4 % 2).reshape(-1)
            Y6, X6 = torch.meshgrid(torch.arange(0, hw // 2, 2), torch.arange(0, hw // 2, 2), indexing="ij")
            p6 = (Y6 * (hw // 2) + X6).reshape(-1)
            d = dict(hw=hw, perm4=r2t.to(torch.int32).to(dev), perm3=p3.to(torch.int32).to(dev), perm2=p2.to(torch.int32).to(dev),
                     idx6=p6.to(torch.int32).to(dev))
            d["s2_d0"] = self._deconv_matrix(self.simfp_2[0], dt)
            d["s2_ln"] = (f32(self.simfp_2[1].weight), f32(self.simfp_2[1].bias), self.simfp_2[1].eps)
            d["s2_d1"] = self._deconv_matrix(self.simfp_2[3], dt)
            d["s3_d0"] = self._deconv_matrix(self.simfp_3[0], dt)
            for name, seq, i1, i3 in (("s2", self.simfp_2, 4, 5), ("s3", self.simfp_3, 1, 2), ("s4", self.simfp_4, 0, 1), ("s5", self.simfp_5, 1, 2)):
                d[name + "_c1"] = (self._conv_matrix(seq[i1], dt), f32(seq[i1].norm.weight), f32(seq[i1].norm.bias), seq[i1].norm.eps)
                d[name + "_c3"] = (self._conv_matrix(seq[i3], dt), f32(seq[i3].norm.weight), f32(seq[i3].norm.bias), seq[i3].norm.eps)
            return d
        return self._pack.get(self, dt, build)

    def _conv_ln_pair(self, x, perm, H, W, c1, c3, dt):
        """1x1 conv + LN (any row order), then 3x3 conv + LN through the gathering im2col -> raster [H*W, C]"""
        y = ops.layernorm(ops.gemm(x, c1[0], None), c1[1], c1[2], c1[3], out_dtype=dt)
        # large 256-channel maps (p2: 256 x 256 pixels): implicit GEMM, the im2col matrix (302 MB) is never written; else im2col + gemm
        return ops.layernorm(ops.conv3x3(y, perm, H, W, c3[0], None), c3[1], c3[2], c3[3], out_dtype=dt)

    def forward_tokens(self, image, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0), vit_feat=None, stages=None):
        """-> dict name -> ([H*W, C] raster token-major map in the compute dtype, (H, W)).  vit_feat: this image's rows of a
        batched ViT pass (ViT.forward_tokens on a list) -- the pyramid then starts from them."""
        dt = self.compute_dtype
        P = self.packed(dt)
        hw = P["hw"]
        if vit_feat is not None:
            x = vit_feat
        elif stages is not None:
            x = self.net.forward_tokens(image, mean, std, stages=stages)
        else:
            x = self.net.forward_tokens(image, mean, std)                                             # [hw*hw, E] window-major
        out = {}
        # The four scales only share the ViT output: stride 8 / 16 / 32 run as parallel graph branches next to the heavy
        # stride-4 chain (their kernels are small launches that leave most of the chip idle).
        def s3():
            t = ops.gemm(x, P["s3_d0"][0], P["s3_d0"][1]).view(hw * hw * 4, -1)
            return self._conv_ln_pair(t, P["perm3"], 2 * hw, 2 * hw, P["s3_c1"], P["s3_c3"], dt)

        def s5():
            t = ops.maxpool2x2(x, P["perm4"], hw, hw)
            p5 = self._conv_ln_pair(t, None, hw // 2, hw // 2, P["s5_c1"], P["s5_c3"], dt)
            return p5, (ops.gather_rows(p5, P["idx6"]) if self.top_block is not None else None)

        j3 = ops.fork(s3)
        j4 = ops.fork(lambda: self._conv_ln_pair(x, P["perm4"], hw, hw, P["s4_c1"], P["s4_c3"], dt))
        j5 = ops.fork(s5)
        # stride 4: deconv -> LN -> GELU -> deconv -> 1x1+LN -> 3x3+LN   (rows stay in nested order until the im2col)
        t = ops.gemm(x, P["s2_d0"][0], P["s2_d0"][1])
        t = t.view(hw * hw * 4, -1)
        t = ops.layernorm(t, P["s2_ln"][0], P["s2_ln"][1], P["s2_ln"][2], out_dtype=dt, act=ops.ACT_GELU)
        t = ops.gemm(t, P["s2_d1"][0], P["s2_d1"][1]).view(hw * hw * 16, -1)
        out["p2"] = (self._conv_ln_pair(t, P["perm2"], 4 * hw, 4 * hw, P["s2_c1"], P["s2_c3"], dt), (4 * hw, 4 * hw))
        out["p3"] = (j3.join(), (2 * hw, 2 * hw))
        out["p4"] = (j4.join(), (hw, hw))
        p5, p6 = j5.join()
        out["p5"] = (p5, (hw // 2, hw // 2))
        if self.top_block is not None:
            out["p6"] = (p6, (hw // 4, hw // 4))
        return out

    def forward(self, x):
        """reference signature: NCHW batch -> {"p2".."p6": NCHW}"""
        res = {}
        for b in range(x.shape[0]):
            maps = self.forward_tokens(x[b].float().contiguous())
            for k, (t, (H, W)) in maps.items():
                res.setdefault(k, []).append(t.float().reshape(H, W, -1).permute(2, 0, 1))
        return {k: torch.stack(v) for k, v in res.items()}
