"""Parameter containers with detrex's attribute names (detrex @776058e: BaseTransformerLayer.attentions / ffns /
norms, FFN.layers = Sequential(Sequential(Linear, ReLU, Dropout), Linear, Dropout), MLP.layers, nn.MultiheadAttention
inside detrex MultiheadAttention.attn, ChannelMapper.convs[i].{conv,norm}) so checkpoints of the reference load
unchanged.  They hold weights and packing logic only -- the arithmetic is in the HIP kernels.
"""
import os

import torch
import torch.nn as nn

from ... import ops
from ...packing import attach_cache, f32, pack_matrix, permute_ffn_w2


class FFN(nn.Module):
    def __init__(self, embed_dim=256, feedforward_dim=1024, output_dim=None, num_fcs=2, ffn_drop=0.0, **kwargs):
        super().__init__()
        assert num_fcs == 2
        output_dim = embed_dim if output_dim is None else output_dim
        self.layers = nn.Sequential(nn.Sequential(nn.Linear(embed_dim, feedforward_dim), nn.ReLU(inplace=True), nn.Dropout(ffn_drop)),
                                    nn.Linear(feedforward_dim, output_dim), nn.Dropout(ffn_drop))
        self.embed_dim = embed_dim
        attach_cache(self)

    def packed(self, dt):
        def build(dt):
            l0, l1 = self.layers[0][0], self.layers[1]
            d = dict(w1=pack_matrix(l0.weight, dt), b1=f32(l0.bias), w2=pack_matrix(l1.weight, dt), b2=f32(l1.bias))
            if dt in ops.HALF16 and l0.weight.shape[0] % 64 == 0:
                d["w2p"] = permute_ffn_w2(d["w2"])            # hidden columns in the fused kernel's k order
            return d
        return self._pack.get(self, dt, build)

    # rows from which the one-kernel FFN (csrc/ffn_fused.hip) replaces the two GEMMs: the encoder's 87 296 tokens write and
    # re-read a 357 MB hidden tensor per layer in the two-GEMM form; APE_FFN_FUSED=0 restores it, =b64 selects the kernel
    # variant on the un-permuted W2 (A/B)
    FUSED_MIN_ROWS = 2048

    def forward_tokens(self, x, dt, out_dtype=None, norm=None):
        """x + Linear(ReLU(Linear(x)))  (detrex FFN with add_identity); norm = (weight, bias, eps): followed by that LayerNorm (the
        layer's post-FFN norm) -- in the same launch on the one-kernel path"""
        P = self.packed(dt)
        mode = os.environ.get("APE_FFN_FUSED", "1")
        if (mode != "0" and "w2p" in P and (out_dtype or dt) in ops.HALF16 and (out_dtype or dt) == dt and x.shape[0] >= self.FUSED_MIN_ROWS
                and x.shape[1] == 256 and P["w2"].shape[0] == 256 and P["w1"].shape[0] <= 4096):
            fuse_ln = norm if os.environ.get("APE_FFN_LN") != "0" else None
            if mode == "b64":
                y = ops.ffn_fused(x, P["w1"], P["b1"], P["w2"], P["b2"], residual=x, norm=fuse_ln)
            else:
                y = ops.ffn_fused(x, P["w1"], P["b1"], P["w2p"], P["b2"], residual=x, w2_permuted=True, norm=fuse_ln)
            return y if (norm is None or fuse_ln is not None) else ops.layernorm(y, norm[0], norm[1], norm[2], out_dtype=dt)
        h = ops.gemm(x, P["w1"], P["b1"], act=ops.ACT_RELU)
        y = ops.gemm(h, P["w2"], P["b2"], residual=x, out_dtype=out_dtype or dt)
        return y if norm is None else ops.layernorm(y, norm[0], norm[1], norm[2], out_dtype=dt)


class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))
        attach_cache(self)

    def packed(self, dt):
        def build(dt):
            return [(pack_matrix(l.weight, dt), f32(l.bias)) for l in self.layers]
        return self._pack.get(self, dt, build)

    def forward_tokens(self, x, dt, out_dtype=torch.float32):
        """ReLU between layers, none after the last; last layer written in `out_dtype`"""
        P = self.packed(dt)
        for i, (w, b) in enumerate(P):
            last = i == len(P) - 1
            x = ops.gemm(x, w, b, act=ops.ACT_NONE if last else ops.ACT_RELU, out_dtype=out_dtype if last else dt)
        return x


class SelfAttention(nn.Module):
    """detrex MultiheadAttention: holds nn.MultiheadAttention as `.attn` (in_proj_weight / in_proj_bias / out_proj)"""

    def __init__(self, embed_dim, num_heads, attn_drop=0.0, proj_drop=0.0, batch_first=False, **kwargs):
        super().__init__()
        self.embed_dim, self.num_heads, self.batch_first = embed_dim, num_heads, batch_first
        self.attn = nn.MultiheadAttention(embed_dim=embed_dim, num_heads=num_heads, dropout=attn_drop, batch_first=batch_first)
        self.proj_drop = nn.Dropout(proj_drop)
        attach_cache(self)

    def packed(self, dt):
        def build(dt):
            E = self.embed_dim
            w, b = self.attn.in_proj_weight, self.attn.in_proj_bias
            return dict(wqk=pack_matrix(w[:2 * E], dt), bqk=f32(b[:2 * E]), wv=pack_matrix(w[2 * E:], dt), bv=f32(b[2 * E:]),
                        wo=pack_matrix(self.attn.out_proj.weight, dt), bo=f32(self.attn.out_proj.bias))
        return self._pack.get(self, dt, build)

    def forward_tokens(self, x, x_pos, dt, vt_buf, out_dtype=None):
        """x [Q,E], x_pos = x + query_pos: q = k = x_pos, v = x; returns x + out_proj(attention)"""
        P = self.packed(dt)
        E, nh = self.embed_dim, self.num_heads
        hd = E // nh
        vjob = ops.fork(lambda: ops.gemm(x, P["wv"], P["bv"], trans_out=True, out=vt_buf))   # parallel graph branch
        qk = ops.gemm(x_pos, P["wqk"], P["bqk"])
        vt = vjob.join()
        o = ops.attention(qk[:, :E], qk[:, E:], vt, batch=1, n=x.shape[0], heads=nh, head_dim=hd, scale=hd ** -0.5)
        return ops.gemm(o, P["wo"], P["bo"], residual=x, out_dtype=out_dtype or dt)


class TransformerLayer(nn.Module):
    """detrex BaseTransformerLayer as a container: attentions / ffns / norms ModuleLists"""

    def __init__(self, attentions, ffn, num_norms, embed_dim=256):
        super().__init__()
        self.attentions = nn.ModuleList(attentions)
        self.ffns = nn.ModuleList([ffn])
        self.norms = nn.ModuleList([nn.LayerNorm(embed_dim) for _ in range(num_norms)])
        self.embed_dim = embed_dim
        self.pre_norm = False

    def norm_params(self, i):
        n = self.norms[i]
        return f32(n.weight), f32(n.bias), n.eps


class ConvNormAct(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=1, bias=True, norm_layer=None):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, padding=(kernel_size - 1) // 2, bias=bias)
        self.norm = norm_layer


class ChannelMapper(nn.Module):
    """stand-in for detrex.modeling.neck.ChannelMapper (1x1 conv + GroupNorm per level; config
    ape_deta_vitl_eva02_clip_vlf_lsj1024_cp_16x4_1080k.py:42-55).  Accepts ShapeSpec-like objects or ints."""

    def __init__(self, input_shapes, in_features, out_channels, kernel_size=1, stride=1, bias=True, groups=1, dilation=1,
                 norm_layer=None, activation=None, num_outs=None, **kwargs):
        super().__init__()
        import copy
        assert kernel_size == 1 and activation is None
        chans = [getattr(input_shapes[f], "channels", input_shapes[f]) for f in in_features]
        assert num_outs is None or num_outs == len(chans), "extra downsampling convs are not used by the APE configs"
        self.convs = nn.ModuleList(ConvNormAct(c, out_channels, 1, bias, copy.deepcopy(norm_layer)) for c in chans)
        self.in_features, self.out_channels, self.input_shapes = in_features, out_channels, input_shapes


class PositionEmbeddingSine(nn.Module):
    """settings holder for detrex.layers.PositionEmbeddingSine (ape_deta_r50.py:35-40)"""

    def __init__(self, num_pos_feats=64, temperature=10000, scale=2 * 3.141592653589793, eps=1e-6, offset=0.0, normalize=False):
        super().__init__()
        self.num_pos_feats, self.temperature, self.scale, self.eps, self.offset, self.normalize = (
            num_pos_feats, temperature, scale, eps, offset, normalize)
