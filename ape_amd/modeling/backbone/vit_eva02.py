"""EVA-02 (MIM-pretrained) ViT backbone of APE-Ti / APE-L_A..C on the HIP kernels.

Host-side mirror of ape/modeling/backbone/vit_eva02.py (ViT :464-634, Block :357-461, Attention :206-291, xops_SwiGLU
:43-176 / SwiGLU :179-203) with the constructor of configs/common/backbone/vitt_eva02.py:10-41: same class names,
kwargs and state-dict keys (`attn.qkv.weight`, `attn.q_bias`, `attn.v_bias`, `mlp.w12.*`, `mlp.w3.*`).  Differences from
the EVA-02-CLIP ViT of APE-L_D (vit_eva_clip.py): one packed qkv Linear and no sub-LayerNorms (subln=False), a packed
w1|w2 SwiGLU without ffn_ln (swiglu=True), and -- with window_size 14 on a 64 x 64 token grid -- windows that do NOT
tile the grid: window_partition (utils_eva02.py:19-40) zero-pads the NORMALISED tokens to 70 x 70, and the 804 padding
tokens take part in the attention of their windows as keys with k = 0 and v = v_bias.

MI355X mapping.  The residual stream stays raster-ordered [4096, E]; a windowed block gathers the LayerNorm output into
window-major order (25 windows x 196 slots, stored at a stride of 200 rows so that every window starts 16-byte aligned
in V^T), padding / slack slots reading an all-zero row -- so the q|k GEMM (RoPE in its epilogue) and the V^T GEMM give
exactly q = rope(q_bias), k = 0, v = v_bias there -- runs the strided flash-attention kernel and gathers the real
tokens back.  SimpleFeaturePyramid is the one of vit_eva_clip.py (identical in the reference, vit_eva02.py:637-804).
"""
import math
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...packing import attach_cache, f32, pack_matrix, round_up
from ...stagetap import tap
from .vit_eva_clip import Backbone, PatchEmbed, SimpleFeaturePyramid, VisionRotaryEmbeddingFast  # noqa: F401

__all__ = ["ViT", "SimpleFeaturePyramid"]


class xops_SwiGLU(nn.Module):
    """parameter holder of the packed SwiGLU (vit_eva02.py:43-83): w12 = [w1; w2] stacked along the output axis"""

    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.w12 = nn.Linear(in_features, 2 * hidden_features)
        self.w3 = nn.Linear(hidden_features, in_features)
        self.hidden_features = hidden_features


class SwiGLU(nn.Module):
    """naiveswiglu=True variant (vit_eva02.py:179-203): separate w1 / w2, sub-LayerNorm before w3"""

    def __init__(self, in_features, hidden_features, norm_layer):
        super().__init__()
        self.w1 = nn.Linear(in_features, hidden_features)
        self.w2 = nn.Linear(in_features, hidden_features)
        self.ffn_ln = norm_layer(hidden_features)
        self.w3 = nn.Linear(hidden_features, in_features)
        self.hidden_features = hidden_features


class Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias, rope, subln):
        super().__init__()
        self.num_heads, self.subln, self.rope = num_heads, subln, rope
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
        self.proj = nn.Linear(dim, dim)

    def qkv_weights(self):
        if self.subln:
            return self.q_proj.weight, self.k_proj.weight, self.v_proj.weight
        return self.qkv.weight.chunk(3, dim=0)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, qkv_bias, norm_layer, window_size, rope, subln, swiglu, naiveswiglu):
        super().__init__()
        assert swiglu or naiveswiglu, "vit_eva02.Block: swiglu or naiveswiglu (vit_eva02.py:414-433)"
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads, qkv_bias, rope, subln)
        self.norm2 = norm_layer(dim)
        hidden = int(dim * mlp_ratio)
        self.mlp = xops_SwiGLU(dim, hidden) if swiglu else SwiGLU(dim, hidden, norm_layer)
        self.window_size = window_size
        attach_cache(self)

    def packed(self, dt):
        def build(dt):
            a, m = self.attn, self.mlp
            wq, wk, wv = (w.detach().float() for w in a.qkv_weights())
            E = wq.shape[0]
            zeros = torch.zeros(E, dtype=torch.float32, device=wq.device)
            hid = m.hidden_features
            hid_pad = round_up(hid, 64)
            if isinstance(m, xops_SwiGLU):          # _ordered_params (:107-149): w1 = first half of w12, w2 = second half
                w1, w2 = m.w12.weight.detach().float().chunk(2, dim=0)
                b1, b2 = m.w12.bias.detach().float().chunk(2, dim=0)
            else:
                w1, w2, b1, b2 = m.w1.weight.detach().float(), m.w2.weight.detach().float(), m.w1.bias.detach().float(), m.w2.bias.detach().float()
            w12 = torch.zeros((2 * hid_pad, E), dtype=torch.float32, device=wq.device)     # (gate, up) rows interleaved
            b12 = torch.zeros((2 * hid_pad,), dtype=torch.float32, device=wq.device)
            w12[0:2 * hid:2], w12[1:2 * hid:2] = w1, w2
            b12[0:2 * hid:2], b12[1:2 * hid:2] = b1, b2
            d = dict(
                wqk=pack_matrix(torch.cat([wq, wk], 0), dt),
                bqk=torch.cat([f32(a.q_bias) if a.q_bias is not None else zeros, zeros]).contiguous(),
                wv=pack_matrix(wv, dt), bv=f32(a.v_bias) if a.v_bias is not None else zeros,
                wproj=pack_matrix(a.proj.weight, dt), bproj=f32(a.proj.bias),
                w12=pack_matrix(w12, dt), b12=b12, hid=hid, hid_pad=hid_pad,
                w3=pack_matrix(m.w3.weight, dt, kpad=64), b3=f32(m.w3.bias),
                n1=(f32(self.norm1.weight), f32(self.norm1.bias), self.norm1.eps),
                n2=(f32(self.norm2.weight), f32(self.norm2.bias), self.norm2.eps),
            )
            if isinstance(m, SwiGLU):
                d["nffn"] = (f32(m.ffn_ln.weight), f32(m.ffn_ln.bias), m.ffn_ln.eps)
            return d
        return self._pack.get(self, dt, build)

    def forward_tokens(self, x, dt, rope, win, xn_buf, vt_buf, last=False):
        """x [N, E] fp32 residual stream in RASTER order; rope = (cos, sin, rows); win = window tables or None (global block);
        xn_buf [N + 1, E] (row N stays zero: the padding token after norm1)."""
        P = self.packed(dt)
        N, E = x.shape
        nh = self.attn.num_heads
        hd = E // nh
        ops.layernorm(x, P["n1"][0], P["n1"][1], P["n1"][2], out=xn_buf[:N])
        if win is not None:
            xw = ops.gather_rows(xn_buf, win["slot2tok"])                    # window-major, padding / slack slots = zero row
            qk = ops.gemm(xw, P["wqk"], P["bqk"], rope=(rope[0], rope[1], rope[2], hd, 2 * E))
            ops.gemm(xw, P["wv"], P["bv"], trans_out=True, out=vt_buf)
            o = ops.attention(qk[:, :E], qk[:, E:], vt_buf, batch=win["nwin"], n=win["ntok"], heads=nh, head_dim=hd, scale=hd ** -0.5,
                              stride=win["stride"])
            o = ops.gather_rows(o, win["tok2slot"])                          # window_unpartition + crop (:43-63)
        else:
            xn = xn_buf[:N]
            qk = ops.gemm(xn, P["wqk"], P["bqk"], rope=(rope[0], rope[1], rope[2], hd, 2 * E))
            ops.gemm(xn, P["wv"], P["bv"], trans_out=True, out=vt_buf)
            o = ops.attention(qk[:, :E], qk[:, E:], vt_buf, batch=1, n=N, heads=nh, head_dim=hd, scale=hd ** -0.5)
        x = ops.gemm(o, P["wproj"], P["bproj"], residual=x, out_dtype=torch.float32)
        xn2 = ops.layernorm(x, P["n2"][0], P["n2"][1], P["n2"][2], out_dtype=dt)
        h = torch.empty((N, P["hid_pad"]), dtype=dt, device=x.device)
        ops.gemm(xn2, P["w12"], P["b12"], act=ops.ACT_SWIGLU, out=h)
        if "nffn" in P:
            h = ops.layernorm(h[:, :P["hid"]], P["nffn"][0], P["nffn"][1], P["nffn"][2], out_dtype=dt, cpad=P["hid_pad"])
        return ops.gemm(h, P["w3"], P["b3"], residual=x, out_dtype=dt if last else torch.float32)


class ViT(Backbone):
    def __init__(self, img_size=1024, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4 * 2 / 3,
                 qkv_bias=True, drop_path_rate=0.0, norm_layer=partial(nn.LayerNorm, eps=1e-6), act_layer=nn.GELU,
                 use_abs_pos=True, use_rel_pos=False, rope=True, pt_hw_seq_len=16, intp_freq=True, window_size=0,
                 window_block_indexes=(), residual_block_indexes=(), use_act_checkpoint=False, pretrain_img_size=224,
                 pretrain_use_cls_token=True, out_feature="last_feat", xattn=True, subln=False, swiglu=False,
                 naiveswiglu=False, frozen_stages=-1):
        super().__init__()
        assert rope and use_abs_pos and patch_size == 16 and len(residual_block_indexes) == 0, \
            "ape_amd vit_eva02.ViT: the rope + abs-pos configuration without residual conv blocks is implemented"
        assert (embed_dim // num_heads) in (32, 64), "the HIP attention kernel handles head_dim 32 / 64"
        self.pretrain_use_cls_token = pretrain_use_cls_token
        self.img_size, self.patch_size, self.embed_dim, self.window_size = img_size, patch_size, embed_dim, window_size
        self.patch_embed = PatchEmbed(in_chans=in_chans, embed_dim=embed_dim)
        num_patches = (pretrain_img_size // patch_size) ** 2
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + (1 if pretrain_use_cls_token else 0), embed_dim))
        half_head_dim = embed_dim // num_heads // 2
        hw = img_size // patch_size
        self.rope_win = VisionRotaryEmbeddingFast(half_head_dim, pt_seq_len=pt_hw_seq_len, ft_seq_len=window_size if intp_freq else None)
        self.rope_glb = VisionRotaryEmbeddingFast(half_head_dim, pt_seq_len=pt_hw_seq_len, ft_seq_len=hw if intp_freq else None)
        self.blocks = nn.ModuleList([
            Block(embed_dim, num_heads, mlp_ratio, qkv_bias, norm_layer, window_size if i in window_block_indexes else 0,
                  self.rope_win if i in window_block_indexes else self.rope_glb, subln, swiglu, naiveswiglu)
            for i in range(depth)])
        self._out_feature_channels = {out_feature: embed_dim}
        self._out_feature_strides = {out_feature: patch_size}
        self._out_features = [out_feature]
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        self.compute_dtype = torch.bfloat16
        attach_cache(self)

    def token_order(self, hw):
        """(tok2raster, raster2tok) of the feature this ViT hands to the pyramid: raster order"""
        r = torch.arange(hw * hw, dtype=torch.int32)
        return r, r.clone()

    def packed(self, dt):
        def build(dt):
            hw = self.img_size // self.patch_size
            dev = self.pos_embed.device
            n = hw * hw
            pos = self.pos_embed.detach().float()
            if self.pretrain_use_cls_token:
                pos = pos[:, 1:]
            size = int(math.sqrt(pos.shape[1]))
            if size != hw:                             # get_abs_pos (utils_eva02.py:158-187)
                pos = F.interpolate(pos.reshape(1, size, size, -1).permute(0, 3, 1, 2), size=(hw, hw), mode="bicubic",
                                    align_corners=False).permute(0, 2, 3, 1)
            w = self.patch_embed.proj.weight
            d = dict(hw=hw, ident=torch.arange(n, dtype=torch.int32, device=dev), pos=pos.reshape(n, -1).contiguous(),
                     wpe=pack_matrix(w.reshape(w.shape[0], -1), dt), bpe=f32(self.patch_embed.proj.bias),
                     rope_glb=(f32(self.rope_glb.freqs_cos), f32(self.rope_glb.freqs_sin), n), win=None)
            ws = self.window_size
            if ws > 0:
                # window_partition (utils_eva02.py:19-40): pad to a multiple of ws, windows row-major, tokens row-major in a
                # window.  slot (window, i) lives at row window * stride + i; stride = ntok rounded up to 8 (V^T alignment).
                hp = round_up(hw, ws)
                nside, ntok = hp // ws, ws * ws
                stride = round_up(ntok, 8)
                slot2tok = torch.full((nside * nside * stride,), n, dtype=torch.int32)        # default: the zero row
                tok2slot = torch.empty(n, dtype=torch.int32)
                Y, X = torch.meshgrid(torch.arange(hw), torch.arange(hw), indexing="ij")
                slot = ((Y // ws) * nside + X // ws) * stride + (Y % ws) * ws + X % ws
                slot2tok[slot.reshape(-1)] = torch.arange(n, dtype=torch.int32)
                tok2slot[:] = slot.reshape(-1).to(torch.int32)
                cos = torch.zeros((stride, self.rope_win.freqs_cos.shape[1]), dtype=torch.float32)
                sin = torch.zeros_like(cos)
                cos[:ntok], sin[:ntok] = self.rope_win.freqs_cos.float().cpu(), self.rope_win.freqs_sin.float().cpu()
                d["win"] = dict(nwin=nside * nside, ntok=ntok, stride=stride, slot2tok=slot2tok.to(dev), tok2slot=tok2slot.to(dev))
                d["rope_win"] = (cos.to(dev), sin.to(dev), stride)
            return d
        return self._pack.get(self, dt, build)

    def forward_tokens(self, image, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0), stages=None):
        """image [3,h,w] fp32 (h,w <= img_size) -> last feature [N, E] in the compute dtype, RASTER token order"""
        dt = self.compute_dtype
        P = self.packed(dt)
        hw = P["hw"]
        n = hw * hw
        if isinstance(image, (list, tuple)):
            return torch.cat([self.forward_tokens(im, mean, std) for im in image], 0)
        patches = ops.patchify(image, P["ident"], hw, hw, mean, std, out_dtype=dt)
        x = tap(stages, "vit_embed", ops.gemm(patches, P["wpe"], P["bpe"], residual=P["pos"], out_dtype=torch.float32))
        xn_buf = torch.zeros((n + 1, self.embed_dim), dtype=dt, device=x.device)
        rows = n if P["win"] is None else max(n, P["win"]["nwin"] * P["win"]["stride"])
        # the attention kernel reads V^T in 64-column tiles from every window's first column: the last window's last tile ends
        # at (nwin - 1) * stride + round_up(ntok, 64) <= round_up(rows, 64) + 64 (ops.attention checks the bound)
        vt_buf = ops.zeros((self.embed_dim, round_up(rows, 64) + 64), dt, x.device)
        for i, blk in enumerate(self.blocks):
            last = i == len(self.blocks) - 1
            if blk.window_size > 0:
                x = blk.forward_tokens(x, dt, P["rope_win"], P["win"], xn_buf, vt_buf, last)
            else:
                x = blk.forward_tokens(x, dt, P["rope_glb"], None, xn_buf, vt_buf, last)
            x = tap(stages, f"vit_blk{i}", x)
        return x

    def forward(self, x):
        """reference signature: normalised, padded NCHW batch -> {"last_feat": [B, E, h/16, w/16]}"""
        outs = []
        hw = self.img_size // self.patch_size
        for b in range(x.shape[0]):
            t = self.forward_tokens(x[b].float().contiguous())
            outs.append(t.float().reshape(hw, hw, -1).permute(2, 0, 1))
        return {self._out_features[0]: torch.stack(outs)}
