"""CPU restatement (fp32, plain torch) of the reference's APE-L_D inference forward pass.

TEST INFRASTRUCTURE ONLY -- the checker, never the product: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg import this file; nothing under ape_amd/ does.

Every function cites the reference lines it restates (paths relative to shenyunhang/APE).  Third-party pieces
(detrex / detectron2 / torchvision) come from oracle/thirdparty.py.  Pinning: tests/test_oracle_vs_reference.py
executes the reference's own files (oracle/refshim.py, this container only) on the same seeded weights/inputs
and compares every stage; tests/golden/ holds outputs of that reference run for the GPU box.

Parameters are addressed by the reference's state-dict names (prefix "model_vision."), so the oracle also
checks the checkpoint-key contract.

Defined tie rule (the reference's is unspecified): per-level proposal top-k (deformable_transformer_vl.py:582-589)
breaks ties -- in practice the exact zeros of other levels' tokens -- by lowest token index.
"""
import math

import torch
import torch.nn.functional as F

from . import thirdparty as tp
from .configs import window_block_indexes

PIXEL_MEAN = (123.675, 116.280, 103.530)  # ape_deta_r50.py:118-119
PIXEL_STD = (58.395, 57.120, 57.375)


# ------------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------------
def stable_topk(values, k):
    """top-k indices by (value desc, index asc) -- the oracle's defined tie rule"""
    return torch.sort(values, descending=True, stable=True)[1][:k]


def rotate_half(x):
    """utils_eva02.py:248-252: pairs (2i, 2i+1) -> (-x[2i+1], x[2i])"""
    x1, x2 = x[..., 0::2], x[..., 1::2]
    return torch.stack((-x2, x1), dim=-1).flatten(-2)


def rope_tables(ft_seq_len, half_head_dim=32, pt_seq_len=16, theta=10000.0):
    """VisionRotaryEmbeddingFast.__init__ (utils_eva02.py:307-344): cos/sin tables [ft*ft, 2*half_head_dim]"""
    dim = half_head_dim
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
    t = torch.arange(ft_seq_len) / ft_seq_len * pt_seq_len
    f = t[:, None] * freqs[None, :]
    f = f.repeat_interleave(2, dim=-1)  # '... n -> ... (n r)', r=2
    fh = f[:, None, :].expand(ft_seq_len, ft_seq_len, dim)
    fw = f[None, :, :].expand(ft_seq_len, ft_seq_len, dim)
    full = torch.cat((fh, fw), dim=-1)
    return full.cos().reshape(-1, 2 * dim), full.sin().reshape(-1, 2 * dim)


def window_partition(x, ws):
    """utils_eva02.py:19-40 (H, W multiples of ws for every configuration used here)"""
    B, H, W, C = x.shape
    assert H % ws == 0 and W % ws == 0
    x = x.view(B, H // ws, ws, W // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)


def window_unpartition(w, ws, H, W):
    """utils_eva02.py:43-63"""
    B = w.shape[0] // (H * W // ws // ws)
    x = w.view(B, H // ws, W // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


def msda_core(value, spatial_shapes, sampling_locations, attention_weights):
    """multi_scale_deformable_attn_pytorch (ape/layers/multi_scale_deform_attn.py:84-124)"""
    bs, _, num_heads, dims = value.shape
    _, nq, _, nl, npnt, _ = sampling_locations.shape
    value_list = value.split([h * w for h, w in spatial_shapes], dim=1)
    grids = 2 * sampling_locations - 1
    sampled = []
    for lvl, (h, w) in enumerate(spatial_shapes):
        v = value_list[lvl].flatten(2).transpose(1, 2).reshape(bs * num_heads, dims, h, w)
        g = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)
        sampled.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    aw = attention_weights.transpose(1, 2).reshape(bs * num_heads, 1, nq, nl * npnt)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * aw).sum(-1).view(bs, num_heads * dims, nq)
    return out.transpose(1, 2).contiguous()


class ApeOracle:
    def __init__(self, cfg, state_dict, prefix="model_vision."):
        self.cfg = dict(cfg)
        self.sd = state_dict
        self.prefix = prefix
        self.num_heads_vit = cfg["num_heads"]
        self.depth = cfg["depth"]
        self.ws = cfg["window_size"]
        ge = cfg.get("global_every", 3)      # every ge-th block is global (vitl_eva02_clip.py:21-28: 3; vitl_eva02.py:21-24: 6)
        self.win_blocks = set(window_block_indexes(self.depth)) if ge == 3 else {i for i in range(self.depth) if i % ge != ge - 1}
        self.vl = bool(cfg.get("vl", True))  # False: APE-L_A/B/C -- DeformableDETRSegm on the plain DeformableDetrTransformer
        self.num_levels = 5
        self.nq = cfg["num_queries"]
        self.enc_layers, self.dec_layers = cfg["enc_layers"], cfg["dec_layers"]
        self.topk_eval = cfg["topk_eval"]
        self.pre_nms_topk, self.nms_thresh_enc = 1000, 0.9  # deformable_transformer_vl.py:279-280
        self.test_nms_thresh, self.test_score_thresh = 0.7, 0.0  # deformable_detr.py:83-84
        hw = cfg["img_size"] // 16
        self.rope_win = rope_tables(self.ws)
        self.rope_glb = rope_tables(hw)
        self.stages = {}
        self.timers = {}

    def _tick(self, key, t0):
        import time
        self.timers[key] = self.timers.get(key, 0.0) + (time.perf_counter() - t0)

    def p(self, name):
        return self.sd[self.prefix + name]

    def lin(self, x, name, bias=True):
        return F.linear(x, self.p(name + ".weight"), self.p(name + ".bias") if bias else None)

    def ln(self, x, name, eps=1e-5):
        w = self.p(name + ".weight")
        return F.layer_norm(x, (w.shape[0],), w, self.p(name + ".bias"), eps)

    # --------------------------------------------------------------------------------------------
    # a1: preprocess (deformable_detr_segm_vl.py:846-855, 365-368)
    # --------------------------------------------------------------------------------------------
    def preprocess(self, image):
        mean = torch.tensor(PIXEL_MEAN).view(3, 1, 1)
        std = torch.tensor(PIXEL_STD).view(3, 1, 1)
        x = (image.float() - mean) / std
        x, (h, w) = tp.pad_to_square(x, self.cfg["img_size"])
        S = x.shape[-1]
        img_mask = torch.ones(1, S, S)
        img_mask[0, :h, :w] = 0
        return x[None], img_mask, (h, w)

    # --------------------------------------------------------------------------------------------
    # a2-a5: ViT (vit_eva_clip.py:743-754, Block :505-523, Attention :218-268, SwiGLU :125-132)
    # --------------------------------------------------------------------------------------------
    def vit_attention(self, x, i, rope):
        pre = f"backbone.net.blocks.{i}.attn."
        B, H, W, C = x.shape
        N = H * W
        x = x.reshape(B, N, C)
        nh = self.num_heads_vit
        q = F.linear(x, self.p(pre + "q_proj.weight"), self.p(pre + "q_bias"))
        k = F.linear(x, self.p(pre + "k_proj.weight"), None)
        v = F.linear(x, self.p(pre + "v_proj.weight"), self.p(pre + "v_bias"))
        q = q.reshape(B, N, nh, -1).permute(0, 2, 1, 3)
        k = k.reshape(B, N, nh, -1).permute(0, 2, 1, 3)
        v = v.reshape(B, N, nh, -1).permute(0, 2, 1, 3)
        cos, sin = rope
        q = q * cos + rotate_half(q) * sin
        k = k * cos + rotate_half(k) * sin
        scale = q.shape[-1] ** -0.5
        att = ((q * scale) @ k.transpose(-2, -1)).softmax(dim=-1)
        o = (att @ v).permute(0, 2, 1, 3).reshape(B, N, -1)
        o = self.ln(o, pre + "inner_attn_ln", 1e-6)
        o = self.lin(o, pre + "proj")
        return o.view(B, H, W, C)

    def vit_block(self, x, i):
        pre = f"backbone.net.blocks.{i}."
        shortcut = x
        x = self.ln(x, pre + "norm1", 1e-6)
        if i in self.win_blocks:
            H, W = x.shape[1], x.shape[2]
            x = window_partition(x, self.ws)
            x = self.vit_attention(x, i, self.rope_win)
            x = window_unpartition(x, self.ws, H, W)
        else:
            x = self.vit_attention(x, i, self.rope_glb)
        x = shortcut + x
        h = self.ln(x, pre + "norm2", 1e-6)
        hidden = F.silu(self.lin(h, pre + "mlp.w1")) * self.lin(h, pre + "mlp.w2")
        hidden = self.ln(hidden, pre + "mlp.ffn_ln", 1e-6)
        return x + self.lin(hidden, pre + "mlp.w3")

    # --------------------------------------------------------------------------------------------
    # config 1 (APE-Ti): the EVA-02 MIM ViT of vit_eva02.py -- packed qkv without sub-LN (Attention :206-291, subln=False),
    # packed SwiGLU without ffn_ln (xops_SwiGLU :85-98, :107-149), zero-padded windows (Block :437-458 with
    # utils_eva02.py:19-63: the padding is applied AFTER norm1, padded tokens are keys with k = 0 and v = v_bias)
    # --------------------------------------------------------------------------------------------
    def vit_attention_eva02(self, x, i, rope):
        pre = f"backbone.net.blocks.{i}.attn."
        B, H, W, C = x.shape
        N = H * W
        x = x.reshape(B, N, C)
        nh = self.num_heads_vit
        q_bias, v_bias = self.p(pre + "q_bias"), self.p(pre + "v_bias")
        if self.cfg.get("subln"):            # vit_eva02.py:250-257: separate projections (q_bias / no k bias / v_bias), no inner LayerNorm
            q = F.linear(x, self.p(pre + "q_proj.weight"), q_bias).reshape(B, N, nh, -1).permute(0, 2, 1, 3)
            k = F.linear(x, self.p(pre + "k_proj.weight"), None).reshape(B, N, nh, -1).permute(0, 2, 1, 3)
            v = F.linear(x, self.p(pre + "v_proj.weight"), v_bias).reshape(B, N, nh, -1).permute(0, 2, 1, 3)
        else:
            qkv = F.linear(x, self.p(pre + "qkv.weight"), torch.cat((q_bias, torch.zeros_like(v_bias), v_bias)))
            qkv = qkv.reshape(B, N, 3, nh, -1).permute(2, 0, 3, 1, 4)
            q, k, v = qkv[0], qkv[1], qkv[2]
        cos, sin = rope
        q = q * cos + rotate_half(q) * sin
        k = k * cos + rotate_half(k) * sin
        att = ((q * q.shape[-1] ** -0.5) @ k.transpose(-2, -1)).softmax(dim=-1)
        o = (att @ v).permute(0, 2, 1, 3).reshape(B, N, -1)
        return self.lin(o, pre + "proj").view(B, H, W, C)

    def vit_block_eva02(self, x, i):
        pre = f"backbone.net.blocks.{i}."
        shortcut = x
        x = self.ln(x, pre + "norm1", 1e-6)
        if i in self.win_blocks:
            ws = self.ws
            B, H, W, C = x.shape
            ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
            xp = F.pad(x, (0, 0, 0, pw, 0, ph))                                   # utils_eva02.py:31-35
            Hp, Wp = H + ph, W + pw
            xw = window_partition(xp, ws)
            xw = self.vit_attention_eva02(xw, i, self.rope_win)
            x = window_unpartition(xw, ws, Hp, Wp)[:, :H, :W, :].contiguous()    # :57-62
        else:
            x = self.vit_attention_eva02(x, i, self.rope_glb)
        x = shortcut + x
        h = self.ln(x, pre + "norm2", 1e-6)
        if self.cfg.get("subln"):            # SwiGLU with ffn_ln (vit_eva02.py:183-204)
            hidden = F.silu(self.lin(h, pre + "mlp.w1")) * self.lin(h, pre + "mlp.w2")
            hidden = self.ln(hidden, pre + "mlp.ffn_ln", 1e-6)
            return x + self.lin(hidden, pre + "mlp.w3")
        w12, b12 = self.p(pre + "mlp.w12.weight"), self.p(pre + "mlp.w12.bias")
        hid = w12.shape[0] // 2
        hidden = F.silu(F.linear(h, w12[:hid], b12[:hid])) * F.linear(h, w12[hid:], b12[hid:])
        return x + self.lin(hidden, pre + "mlp.w3")

    # --------------------------------------------------------------------------------------------
    # ViT-e (configs/common/backbone/vite_eva02_clip_1024.py:9-49: the SAME classes of vit_eva_clip.py with subln=False,
    # naiveswiglu=False, rope=False, postnorm=True): packed qkv with q_bias / v_bias (Attention :250-262), no rotary embedding,
    # Mlp fc1 -> GELU -> fc2 (:67-98), and the post-norm residual order of Block :505-517:
    #     x = x + norm1(attn(x));  x = x + norm2(mlp(x))
    # --------------------------------------------------------------------------------------------
    def vit_attention_packed(self, x, i):
        pre = f"backbone.net.blocks.{i}.attn."
        B, H, W, C = x.shape
        N = H * W
        x = x.reshape(B, N, C)
        nh = self.num_heads_vit
        q_bias, v_bias = self.p(pre + "q_bias"), self.p(pre + "v_bias")
        qkv = F.linear(x, self.p(pre + "qkv.weight"), torch.cat((q_bias, torch.zeros_like(v_bias), v_bias)))
        qkv = qkv.reshape(B, N, 3, nh, -1).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        att = ((q * q.shape[-1] ** -0.5) @ k.transpose(-2, -1)).softmax(dim=-1)          # == scaled_dot_product_attention (:264-266)
        o = (att @ v).permute(0, 2, 1, 3).reshape(B, N, -1)
        return self.lin(o, pre + "proj").view(B, H, W, C)

    def vit_block_prenorm_packed(self, x, i):
        """EVA-01-CLIP ViT-g (vitg_eva01_clip_1024.py:9-45): the pre-norm order of Block :519-523 around the packed-qkv attention and
        the GELU Mlp -- x = x + attn(norm1(x)); x = x + mlp(norm2(x))"""
        pre = f"backbone.net.blocks.{i}."
        shortcut = x
        x = self.ln(x, pre + "norm1", 1e-6)
        if i in self.win_blocks:
            H, W = x.shape[1], x.shape[2]
            xw = window_partition(x, self.ws)
            xw = self.vit_attention_packed(xw, i)
            x = window_unpartition(xw, self.ws, H, W)
        else:
            x = self.vit_attention_packed(x, i)
        x = shortcut + x
        return x + self.lin(F.gelu(self.lin(self.ln(x, pre + "norm2", 1e-6), pre + "mlp.fc1")), pre + "mlp.fc2")

    # --------------------------------------------------------------------------------------------
    # EVA-01 MIM ViT-g (ape/modeling/backbone/vit_eva.py: Attention :72-146, Block :210-309): the packed-qkv / GELU-Mlp pre-norm block
    # above PLUS decomposed relative positions (utils_eva.py:65-161): attn[q, (kh, kw)] += q . Rh[qh - kh] + q . Rw[qw - kw] with
    # the UNSCALED q; tables of 2 * size - 1 rows (linear interpolation, "vitdet", when the stored length differs)
    # --------------------------------------------------------------------------------------------
    @staticmethod
    def rel_pos_table(q_size, k_size, rel_pos):
        """get_rel_pos (utils_eva.py:65-129, interp_type="vitdet"): -> [q_size, k_size, C]"""
        max_rel_dist = int(2 * max(q_size, k_size) - 1)
        if rel_pos.shape[0] != max_rel_dist:
            rel_pos = F.interpolate(rel_pos.reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1), size=max_rel_dist, mode="linear")
            rel_pos = rel_pos.reshape(-1, max_rel_dist).permute(1, 0)
        q_coords = torch.arange(q_size)[:, None] * max(k_size / q_size, 1.0)
        k_coords = torch.arange(k_size)[None, :] * max(q_size / k_size, 1.0)
        rel = (q_coords - k_coords) + (k_size - 1) * max(q_size / k_size, 1.0)
        return rel_pos[rel.long()]

    def vit_attention_relpos(self, x, i):
        pre = f"backbone.net.blocks.{i}.attn."
        B, H, W, C = x.shape
        N = H * W
        nh = self.num_heads_vit
        q_bias, v_bias = self.p(pre + "q_bias"), self.p(pre + "v_bias")
        qkv = F.linear(x.reshape(B, N, C), self.p(pre + "qkv.weight"), torch.cat((q_bias, torch.zeros_like(v_bias), v_bias)))
        qkv = qkv.reshape(B, N, 3, nh, -1).permute(2, 0, 3, 1, 4)
        q, k, v = (t.reshape(B * nh, N, -1) for t in (qkv[0], qkv[1], qkv[2]))
        att = (q * q.shape[-1] ** -0.5) @ k.transpose(-2, -1)
        Rh, Rw = self.rel_pos_table(H, H, self.p(pre + "rel_pos_h")), self.rel_pos_table(W, W, self.p(pre + "rel_pos_w"))
        r_q = q.reshape(B * nh, H, W, -1)
        rel_h = torch.einsum("bhwc,hkc->bhwk", r_q, Rh)
        rel_w = torch.einsum("bhwc,wkc->bhwk", r_q, Rw)
        att = (att.view(B * nh, H, W, H, W) + rel_h[:, :, :, :, None] + rel_w[:, :, :, None, :]).view(B * nh, N, N).softmax(dim=-1)
        o = (att @ v).view(B, nh, H, W, -1).permute(0, 2, 3, 1, 4).reshape(B, H, W, -1)
        return self.lin(o, pre + "proj")

    def vit_block_eva01(self, x, i):
        pre = f"backbone.net.blocks.{i}."
        shortcut = x
        x = self.ln(x, pre + "norm1", 1e-6)
        if i in self.win_blocks:
            H, W = x.shape[1], x.shape[2]
            xw = window_partition(x, self.ws)
            xw = self.vit_attention_relpos(xw, i)
            x = window_unpartition(xw, self.ws, H, W)
        else:
            x = self.vit_attention_relpos(x, i)
        x = shortcut + x
        return x + self.lin(F.gelu(self.lin(self.ln(x, pre + "norm2", 1e-6), pre + "mlp.fc1")), pre + "mlp.fc2")

    def vit_block_postnorm(self, x, i):
        pre = f"backbone.net.blocks.{i}."
        shortcut = x
        if i in self.win_blocks:
            H, W = x.shape[1], x.shape[2]
            xw = window_partition(x, self.ws)
            xw = self.vit_attention_packed(xw, i)
            x = window_unpartition(xw, self.ws, H, W)
        else:
            x = self.vit_attention_packed(x, i)
        x = shortcut + self.ln(x, pre + "norm1", 1e-6)
        h = self.lin(F.gelu(self.lin(x, pre + "mlp.fc1")), pre + "mlp.fc2")
        return x + self.ln(h, pre + "norm2", 1e-6)

    def abs_pos(self, hw):
        """get_abs_pos (utils_eva02.py:158-187): drop cls, bicubic resize to the token grid"""
        pos = self.p("backbone.net.pos_embed")[:, 1:]
        size = int(math.sqrt(pos.shape[1]))
        if size != hw[0] or size != hw[1]:
            pos = F.interpolate(pos.reshape(1, size, size, -1).permute(0, 3, 1, 2), size=hw, mode="bicubic",
                                align_corners=False).permute(0, 2, 3, 1)
        else:
            pos = pos.reshape(1, hw[0], hw[1], -1)
        return pos

    def vit(self, images):
        x = F.conv2d(images, self.p("backbone.net.patch_embed.proj.weight"), self.p("backbone.net.patch_embed.proj.bias"),
                     stride=16)
        x = x.permute(0, 2, 3, 1)
        x = x + self.abs_pos((x.shape[1], x.shape[2]))
        self.stages["vit_embed"] = x
        import time
        for i in range(self.depth):
            t0 = time.perf_counter()
            kind = self.cfg.get("backbone")
            x = (self.vit_block_eva02(x, i) if kind == "eva02" else self.vit_block_postnorm(x, i) if kind == "clip_e"
                 else self.vit_block_prenorm_packed(x, i) if kind == "clip_g" else self.vit_block_eva01(x, i) if kind == "eva01"
                 else self.vit_block(x, i))
            self._tick("vit_win_block" if i in self.win_blocks else "vit_glb_block", t0)
            self.stages[f"vit_block{i}"] = x
        return x.permute(0, 3, 1, 2)

    # --------------------------------------------------------------------------------------------
    # a6: SimpleFeaturePyramid (vit_eva_clip.py:804-847, 871-922)
    # --------------------------------------------------------------------------------------------
    def conv_ln(self, x, name, k):
        x = F.conv2d(x, self.p(name + ".weight"), None, padding=k // 2)
        return tp.layer_norm_2d(x, self.p(name + ".norm.weight"), self.p(name + ".norm.bias"), 1e-6)

    def fpn(self, feat):
        pb = "backbone."
        x = F.conv_transpose2d(feat, self.p(pb + "simfp_2.0.weight"), self.p(pb + "simfp_2.0.bias"), stride=2)
        x = tp.layer_norm_2d(x, self.p(pb + "simfp_2.1.weight"), self.p(pb + "simfp_2.1.bias"), 1e-6)
        x = F.gelu(x)
        x = F.conv_transpose2d(x, self.p(pb + "simfp_2.3.weight"), self.p(pb + "simfp_2.3.bias"), stride=2)
        p2 = self.conv_ln(self.conv_ln(x, pb + "simfp_2.4", 1), pb + "simfp_2.5", 3)
        x = F.conv_transpose2d(feat, self.p(pb + "simfp_3.0.weight"), self.p(pb + "simfp_3.0.bias"), stride=2)
        p3 = self.conv_ln(self.conv_ln(x, pb + "simfp_3.1", 1), pb + "simfp_3.2", 3)
        p4 = self.conv_ln(self.conv_ln(feat, pb + "simfp_4.0", 1), pb + "simfp_4.1", 3)
        x = F.max_pool2d(feat, kernel_size=2, stride=2)
        p5 = self.conv_ln(self.conv_ln(x, pb + "simfp_5.1", 1), pb + "simfp_5.2", 3)
        p6 = tp.last_level_max_pool(p5)
        return {"p2": p2, "p3": p3, "p4": p4, "p5": p5, "p6": p6}

    # a7: neck (detrex ChannelMapper: 1x1 conv + bias, GroupNorm(32))
    def neck(self, feats):
        outs = []
        for i, f in enumerate(["p2", "p3", "p4", "p5", "p6"]):
            x = F.conv2d(feats[f], self.p(f"neck.convs.{i}.conv.weight"), self.p(f"neck.convs.{i}.conv.bias"))
            outs.append(F.group_norm(x, 32, self.p(f"neck.convs.{i}.norm.weight"), self.p(f"neck.convs.{i}.norm.bias"), 1e-5))
        return outs

    # --------------------------------------------------------------------------------------------
    # a10: vision-language fusion (fuse_helper.py:67-166, 221-232)
    # --------------------------------------------------------------------------------------------
    def vl_fusion(self, v, l, i):
        pre = f"transformer.encoder.vl_layers.{i}.b_attn."
        v = self.ln(v, pre + "layer_norm_v")
        l = self.ln(l, pre + "layer_norm_l")
        nh, E = 8, 2048
        hd = E // nh
        bsz, tgt, _ = v.shape

        def shape(t):
            return t.view(bsz, -1, nh, hd).transpose(1, 2).contiguous().view(bsz * nh, -1, hd)

        q = shape(self.lin(v, pre + "attn.v_proj") * hd ** -0.5)
        k = shape(self.lin(l, pre + "attn.l_proj"))
        vv = shape(self.lin(v, pre + "attn.values_v_proj"))
        vl = shape(self.lin(l, pre + "attn.values_l_proj"))
        w = torch.bmm(q, k.transpose(1, 2))
        w = w - w.max()                       # stable_softmax_2d: one global max (:89-90)
        w = w.clamp(min=-50000).clamp(max=50000)
        wT = w.transpose(1, 2)
        wl = wT - wT.max(dim=-1, keepdim=True)[0]
        wl = wl.clamp(min=-50000).clamp(max=50000).softmax(dim=-1)   # vision padding NOT masked
        wv = w.softmax(dim=-1)
        ov = torch.bmm(wv, vl).view(bsz, nh, tgt, hd).transpose(1, 2).reshape(bsz, tgt, E)
        ol = torch.bmm(wl, vv).view(bsz, nh, -1, hd).transpose(1, 2).reshape(bsz, -1, E)
        dv = self.lin(ov, pre + "attn.out_v_proj")
        dl = self.lin(ol, pre + "attn.out_l_proj")
        v = v + self.p(pre + "gamma_v") * dv      # residual on the NORMALISED tensors (:224-231)
        l = l + self.p(pre + "gamma_l") * dl
        return v, l

    # --------------------------------------------------------------------------------------------
    # a11: MultiScaleDeformableAttention.forward (multi_scale_deform_attn.py:215-358)
    # --------------------------------------------------------------------------------------------
    def msda(self, pre, query, value, identity, query_pos, key_padding_mask, reference_points, spatial_shapes):
        if query_pos is not None:
            query = query + query_pos
        bs, nq, _ = query.shape
        nv = value.shape[1]
        value = self.lin(value, pre + "value_proj")
        if key_padding_mask is not None:
            value = value.masked_fill(key_padding_mask[..., None], 0.0)
        value = value.view(bs, nv, 8, -1)
        L = len(spatial_shapes)
        off = self.lin(query, pre + "sampling_offsets").view(bs, nq, 8, L, 4, 2)
        aw = self.lin(query, pre + "attention_weights").view(bs, nq, 8, L * 4).softmax(-1).view(bs, nq, 8, L, 4)
        if reference_points.shape[-1] == 2:
            norm = torch.tensor([[w, h] for h, w in spatial_shapes], dtype=torch.float32)
            loc = reference_points[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
        else:
            loc = reference_points[:, :, None, :, None, :2] + off / 4 * reference_points[:, :, None, :, None, 2:] * 0.5
        out = msda_core(value, spatial_shapes, loc, aw)
        return self.lin(out, pre + "output_proj") + identity

    def ffn(self, x, pre):
        """detrex FFN: Linear-ReLU-Linear + identity"""
        return x + self.lin(F.relu(self.lin(x, pre + "layers.0.0")), pre + "layers.1")

    # a9 helpers (deformable_transformer_vl.py:371-410)
    @staticmethod
    def valid_ratio(mask):
        _, H, W = mask.shape
        valid_H = torch.sum(~mask[:, :, 0], 1)
        valid_W = torch.sum(~mask[:, 0, :], 1)
        return torch.stack([valid_W.float() / W, valid_H.float() / H], -1)

    @staticmethod
    def encoder_reference_points(spatial_shapes, valid_ratios):
        pts = []
        for lvl, (H, W) in enumerate(spatial_shapes):
            ref_y, ref_x = torch.meshgrid(torch.linspace(0.5, H - 0.5, H), torch.linspace(0.5, W - 0.5, W), indexing="ij")
            ref_y = ref_y.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * H)
            ref_x = ref_x.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * W)
            pts.append(torch.stack((ref_x, ref_y), -1))
        ref = torch.cat(pts, 1)
        return ref[:, :, None] * valid_ratios[:, None]

    # a13 (deformable_transformer_vl.py:321-369)
    def gen_proposals(self, memory, mask_flat, spatial_shapes, mask_prompt_flat=None):
        N = memory.shape[0]
        proposals, level_ids = [], []
        cur = 0
        for lvl, (H, W) in enumerate(spatial_shapes):
            m = mask_flat[:, cur:cur + H * W].view(N, H, W, 1)
            valid_H = torch.sum(~m[:, :, 0, 0], 1)
            valid_W = torch.sum(~m[:, 0, :, 0], 1)
            gy, gx = torch.meshgrid(torch.linspace(0, H - 1, H), torch.linspace(0, W - 1, W), indexing="ij")
            grid = torch.cat([gx.unsqueeze(-1), gy.unsqueeze(-1)], -1)
            scale = torch.cat([valid_W.unsqueeze(-1), valid_H.unsqueeze(-1)], 1).view(N, 1, 1, 2)
            grid = (grid.unsqueeze(0).expand(N, -1, -1, -1) + 0.5) / scale
            wh = torch.ones_like(grid) * 0.05 * (2.0 ** lvl)
            proposals.append(torch.cat((grid, wh), -1).view(N, -1, 4))
            cur += H * W
            level_ids.append(torch.full((H * W,), lvl, dtype=torch.long))
        prop = torch.cat(proposals, 1)
        valid = ((prop > 0.01) & (prop < 0.99)).all(-1, keepdim=True)
        prop = torch.log(prop / (1 - prop))
        prop = prop.masked_fill(mask_flat.unsqueeze(-1), float("inf")).masked_fill(~valid, float("inf"))
        om = memory.masked_fill(mask_flat.unsqueeze(-1), 0.0).masked_fill(~valid, 0.0)
        if mask_prompt_flat is not None:                 # (:356-358, 364-365) tokens outside the prompted region are no proposals
            prop = prop.masked_fill(~mask_prompt_flat.unsqueeze(-1), float("inf"))
            om = om.masked_fill(~mask_prompt_flat.unsqueeze(-1), 0.0)
        om = self.ln(self.lin(om, "transformer.enc_output"), "transformer.enc_output_norm")
        return om, prop, torch.cat(level_ids)

    def mlp(self, x, pre, n=3):
        """detrex MLP: ReLU between layers, none after the last"""
        for j in range(n):
            x = self.lin(x, f"{pre}.layers.{j}")
            if j < n - 1:
                x = F.relu(x)
        return x

    # a14 (deformable_transformer_vl.py:565-627), batch element b
    def select_proposals(self, logits_b, boxes_b, level_ids, num_levels):
        topk = self.nq
        pre = []
        for lvl in range(num_levels):
            lvl_mask = level_ids == lvl
            pre.append(stable_topk(logits_b.sigmoid() * lvl_mask, min(self.pre_nms_topk, logits_b.shape[0])))
        pre = torch.cat(pre)
        post = tp.batched_nms(boxes_b[pre], logits_b[pre], level_ids[pre], self.nms_thresh_enc)
        keep = pre[post]
        if len(keep) < topk:  # :600-606
            keep = stable_topk(logits_b, min(topk, logits_b.shape[0]))
        q_per_l = topk // num_levels
        is_lvl = level_ids[keep][None] == torch.arange(num_levels)[:, None]
        kmask = (is_lvl & (is_lvl.cumsum(1) <= q_per_l)).any(0)
        if kmask.sum() < topk:
            num_to_add = topk - kmask.sum()
            pad = (~kmask).nonzero()[:num_to_add]
            kmask[pad] = True
        return keep[kmask]

    @staticmethod
    def proposal_pos_embed(proposals, num_pos_feats=128, temperature=10000):
        """deformable_transformer_vl.py:412-420"""
        scale = 2 * math.pi
        dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
        dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_pos_feats)
        proposals = proposals.sigmoid() * scale
        pos = proposals[:, :, :, None] / dim_t
        return torch.stack((pos[:, :, :, 0::2].sin(), pos[:, :, :, 1::2].cos()), dim=4).flatten(2)

    # a17 (vision_language_align.py:27-52)
    def vl_align(self, x, emb, pre):
        e = F.normalize(emb, p=2, dim=-1)
        tok = self.lin(e / 2.0, pre + ".dot_product_projection_text")
        bias = torch.matmul(e, self.p(pre + ".bias_lang")) + self.p(pre + ".bias0")
        logit = torch.matmul(x, tok.transpose(-1, -2)) / self.p(pre + ".log_scale").exp() + bias.unsqueeze(1)
        return logit.clamp(max=50000).clamp(min=-50000)

    # --------------------------------------------------------------------------------------------
    # a9-a16: DeformableDetrTransformerVL.forward (deformable_transformer_vl.py:422-699)
    # --------------------------------------------------------------------------------------------
    def transformer(self, feats, masks, pos_embeds, query_l, forced_topk=None, masks_prompt=None):
        S = self.stages
        spatial_shapes = [(f.shape[2], f.shape[3]) for f in feats]
        feat = torch.cat([f.flatten(2).transpose(1, 2) for f in feats], 1)
        mask = torch.cat([m.flatten(1) for m in masks], 1)
        lvl_pos = torch.cat([p.flatten(2).transpose(1, 2) + self.p("transformer.level_embeds")[i].view(1, 1, -1)
                             for i, p in enumerate(pos_embeds)], 1)
        valid_ratios = torch.stack([self.valid_ratio(m) for m in masks], 1)
        ref = self.encoder_reference_points(spatial_shapes, valid_ratios)
        S["enc_input"], S["lvl_pos"], S["valid_ratios"] = feat, lvl_pos, valid_ratios

        # encoder (:84-115): VL fusion, then BaseTransformerLayer("self_attn","norm","ffn","norm")
        import time
        x, l = feat, query_l
        for i in range(self.enc_layers):
            t0 = time.perf_counter()
            if self.vl:                     # deformable_transformer_vl.py:84-91; the plain encoder (deformable_transformer.py:78) has no fusion
                x, l = self.vl_fusion(x, l, i)
                S[f"enc{i}_fused_v"], S[f"enc{i}_fused_l"] = x, l
            pre = f"transformer.encoder.layers.{i}."
            x = self.msda(pre + "attentions.0.", x, x, x, lvl_pos, mask, ref, spatial_shapes)
            x = self.ln(x, pre + "norms.0")
            x = self.ffn(x, pre + "ffns.0.")
            x = self.ln(x, pre + "norms.1")
            self._tick("enc_layer", t0)
            S[f"enc{i}_out"] = x
        memory = x
        S["memory"], S["query_l"] = memory, l

        # two-stage proposals (:495-533)
        mpf = None if masks_prompt is None else torch.cat([m.flatten(1) for m in masks_prompt], 1)      # (:465-470)
        om, props, level_ids = self.gen_proposals(memory, mask, spatial_shapes, mpf)
        nd = self.dec_layers
        cls = self.lin(om, f"transformer.decoder.class_embed.{nd}")
        box = self.mlp(om, f"transformer.decoder.bbox_embed.{nd}") + props
        if (self.prefix + "transformer.decoder.class_embed_ambiguous.0.weight") in self.sd:    # proposal_ambiguous = 1 (:508-533)
            cls_a = self.lin(om, "transformer.decoder.class_embed_ambiguous.0")
            box_a = self.mlp(om, "transformer.decoder.bbox_embed_ambiguous.0") + props
            cls2 = torch.stack([cls, cls_a], dim=1)
            box2 = torch.stack([box, box_a], dim=1)
            idx = torch.argmax(cls2, dim=1, keepdim=True)
            enc_class = torch.gather(cls2, 1, idx).squeeze(1)
            enc_coord = torch.gather(box2, 1, idx.repeat(1, 1, 1, 4)).squeeze(1)
        else:                                                                                  # proposal_ambiguous = 0 (APE-L_A/B/C)
            enc_class, enc_coord = cls, box
        S["output_memory"], S["enc_class"], S["enc_coord_unact"] = om, enc_class, enc_coord

        logit = enc_class[..., 0]
        boxes = tp.box_cxcywh_to_xyxy(enc_coord.sigmoid()).clamp(0, 1)
        if forced_topk is not None:
            topk = forced_topk
        else:
            topk = torch.stack([self.select_proposals(logit[b], boxes[b], level_ids, len(spatial_shapes))
                                for b in range(feat.shape[0])])
        S["topk_proposals"] = topk

        # query init (:629-645)
        coords = torch.gather(enc_coord, 1, topk.unsqueeze(-1).repeat(1, 1, 4))
        reference = coords.sigmoid()
        init_reference = reference
        pt = self.ln(self.lin(self.proposal_pos_embed(coords), "transformer.pos_trans"), "transformer.pos_trans_norm")
        query_pos, query = torch.split(pt, 256, dim=2)
        feats_topk = torch.stack([om[b][topk[b]] for b in range(om.shape[0])])
        query = query + self.ln(self.lin(feats_topk, "transformer.pix_trans"), "transformer.pix_trans_norm")
        S["query_init"], S["query_pos"] = query, query_pos

        # decoder (:195-250)
        inter, inter_ref = [], []
        out = query
        for i in range(self.dec_layers):
            t0 = time.perf_counter()
            ref_in = reference[:, :, None] * torch.cat([valid_ratios, valid_ratios], -1)[:, None]
            pre = f"transformer.decoder.layers.{i}."
            # self attention: nn.MultiheadAttention(256, 8), q = k = x + pos, v = x (detrex MultiheadAttention)
            qk = out + query_pos
            E = 256
            w_in, b_in = self.p(pre + "attentions.0.attn.in_proj_weight"), self.p(pre + "attentions.0.attn.in_proj_bias")
            q = F.linear(qk, w_in[:E], b_in[:E])
            k = F.linear(qk, w_in[E:2 * E], b_in[E:2 * E])
            v = F.linear(out, w_in[2 * E:], b_in[2 * E:])
            B, Q, _ = q.shape

            def heads(t):
                return t.view(B, Q, 8, 32).transpose(1, 2)

            att = ((heads(q) * 32 ** -0.5) @ heads(k).transpose(-1, -2)).softmax(-1)
            sa = (att @ heads(v)).transpose(1, 2).reshape(B, Q, E)
            out = out + self.lin(sa, pre + "attentions.0.attn.out_proj")
            out = self.ln(out, pre + "norms.0")
            out = self.msda(pre + "attentions.1.", out, memory, out, query_pos, mask, ref_in, spatial_shapes)
            out = self.ln(out, pre + "norms.1")
            out = self.ffn(out, pre + "ffns.0.")
            out = self.ln(out, pre + "norms.2")
            tmp = self.mlp(out, f"transformer.decoder.bbox_embed.{i}")
            reference = (tmp + tp.inverse_sigmoid(reference)).sigmoid()
            inter.append(out)
            inter_ref.append(reference)
            self._tick("dec_layer", t0)
        return (torch.stack(inter), init_reference, torch.stack(inter_ref), enc_class, enc_coord, props.sigmoid(), memory, l,
                spatial_shapes)

    # a18 (deformable_detr_segm_vl.py:728-750)
    def mask_features(self, memory, p2, spatial_shapes):
        h, w = spatial_shapes[0]
        enc = memory[:, : h * w, :].permute(0, 2, 1).reshape(1, -1, h, w)
        x = F.conv2d(p2, self.p("lateral_conv.weight"))
        x = F.group_norm(x, 32, self.p("lateral_conv.norm.weight"), self.p("lateral_conv.norm.bias"), 1e-5)
        x = x + F.interpolate(enc, size=x.shape[-2:], mode="bilinear", align_corners=False)
        x = F.conv2d(x, self.p("output_conv.weight"), padding=1)
        x = F.relu(F.group_norm(x, 32, self.p("output_conv.norm.weight"), self.p("output_conv.norm.bias"), 1e-5))
        return F.conv2d(x, self.p("mask_conv.weight"))

    # a20 (deformable_detr_segm_vl.py:759-810; ape_deta/fast_rcnn.py:97-201)
    def inference(self, box_cls, box_pred, image_size):
        scores = box_cls.sigmoid()
        boxes = tp.box_cxcywh_to_xyxy(box_pred)
        h, w = image_size
        boxes = boxes * torch.tensor([w, h, w, h], dtype=torch.float32)
        valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores).all(dim=1)
        qidx = torch.arange(boxes.shape[0])
        if not valid.all():
            boxes, scores, qidx = boxes[valid], scores[valid], qidx[valid]
        boxes = torch.stack((boxes[:, 0].clamp(0, w), boxes[:, 1].clamp(0, h), boxes[:, 2].clamp(0, w), boxes[:, 3].clamp(0, h)), -1)
        filter_mask = scores > self.test_score_thresh
        filter_inds = filter_mask.nonzero()
        cand_boxes = boxes[filter_inds[:, 0]]
        cand_scores = scores[filter_mask]
        keep = tp.batched_nms(cand_boxes, cand_scores, filter_inds[:, 1], self.test_nms_thresh)
        keep = keep[: self.topk_eval]
        return cand_boxes[keep], cand_scores[keep], filter_inds[keep, 1], qidx[filter_inds[keep, 0]]

    # --------------------------------------------------------------------------------------------
    # whole forward, "name" prompt mode (deformable_detr_segm_vl.py:166-726)
    # --------------------------------------------------------------------------------------------
    # --------------------------------------------------------------------------------------------
    # a22: semantic branch (deformable_detr_segm_vl.py:628-666, 875-918, 1251-1271)
    # --------------------------------------------------------------------------------------------
    @staticmethod
    def stuff_score(box_cls, meta):
        """get_stuff_score (:1251-1271).  meta: dict(entity, thing_classes, stuff_classes)."""
        thing, stuff, entity = meta.get("thing_classes") or [], meta.get("stuff_classes") or [], meta["entity"]
        sem = box_cls.clone()
        if entity == "thing+stuff" and stuff[0] == "things":
            nt = len(thing)
            sem = torch.cat([box_cls[..., :nt].min(dim=-1, keepdim=True)[0], box_cls[..., nt:]], dim=-1)
        overlap = len(thing) > 0 and len(stuff) > 0 and (set(thing) <= set(stuff) or set(stuff) <= set(thing))  # :1215-1226
        if (entity == "thing+stuff" and overlap) or entity == "stuff":
            sem = box_cls.clone()
        return sem

    def semantic_branch(self, logits, coord, pred_masks, padded_size, image_size, height, width, meta, pano_temp=0.06):
        S = self.stages
        sem_cls = self.stuff_score(logits, meta)
        _, _, _, qidx = self.inference(sem_cls[0], coord[0], image_size)          # semantic_post_nms (:638-647)
        S["sem_query"], S["sem_box_cls"] = qidx, sem_cls
        up = F.interpolate(pred_masks[:, qidx], size=padded_size, mode="bilinear", align_corners=False)[0]   # (:569-572)
        mask_cls = F.softmax(sem_cls[0][qidx].sigmoid() / pano_temp, dim=-1)      # (:891-894)
        result = torch.einsum("qc,qhw->chw", mask_cls, up.sigmoid())              # (:895-899)
        r = tp.sem_seg_postprocess(result, image_size, height, width)             # (:916)
        if meta["entity"] == "stuff" and (meta.get("stuff_classes") or [""])[0] == "things" and meta.get("stuff_prob_thing", -1.0) > 0 \
                and meta.get("dataset_id", -1) >= 0:
            p = meta["stuff_prob_thing"]
            r[0, ...] = math.log(p / (1 - p))                                     # (:654-663)
        return r

    def panoptic_branch(self, logits, coord, pred_masks, padded_size, image_size, height, width, meta, cfg):
        """panoptic_post_nms + _postprocess_panoptic (deformable_detr_segm_vl.py:671-695, 921-998): the kept queries'
        masks are merged Mask2Former style.  meta carries thing_classes / stuff_classes / thing_dataset_id_to_contiguous_id."""
        S = self.stages
        _, _, _, qidx = self.inference(logits[0], coord[0], image_size)                         # third NMS, all K columns
        S["pan_query"] = qidx
        mask_cls = logits[0][qidx]
        mask_pred = F.interpolate(pred_masks[:, qidx], size=padded_size, mode="bilinear", align_corners=False)[0]   # (:569-572)
        mask_pred = tp.sem_seg_postprocess(mask_pred, image_size, height, width)                  # (:942) logits, crop + resize
        scores, labels = mask_cls.sigmoid().max(-1)
        mask_pred = mask_pred.sigmoid()
        keep = scores > cfg["object_mask_threshold"]
        if cfg["transform_eval"]:
            scores, labels = F.softmax(mask_cls.sigmoid() / cfg["pano_temp"], dim=-1).max(-1)
        cur_scores, cur_classes, cur_masks = scores[keep], labels[keep], mask_pred[keep]
        cur_prob_masks = cur_scores.view(-1, 1, 1) * cur_masks
        panoptic_seg = torch.zeros((height, width), dtype=torch.int32)
        segments_info = []
        current_segment_id = 0
        thing_ids = set((meta.get("thing_dataset_id_to_contiguous_id") or {}).values())
        if cur_masks.size(0) > 0:
            cur_mask_ids = cur_prob_masks.argmax(0)
            stuff_memory = {}
            for k in range(cur_classes.shape[0]):
                pred_class = int(cur_classes[k])
                isthing = pred_class in thing_ids
                mask_area = int((cur_mask_ids == k).sum())
                original_area = int((cur_masks[k] >= cfg["prob"]).sum())
                mask = (cur_mask_ids == k) & (cur_masks[k] >= cfg["prob"])
                if mask_area > 0 and original_area > 0 and int(mask.sum()) > 0:
                    if mask_area / original_area < cfg["overlap_threshold"]:
                        continue
                    if not isthing:
                        if pred_class in stuff_memory:
                            panoptic_seg[mask] = stuff_memory[pred_class]
                            continue
                        stuff_memory[pred_class] = current_segment_id + 1
                    current_segment_id += 1
                    panoptic_seg[mask] = current_segment_id
                    if not isthing and (meta.get("stuff_classes") or [""])[0] == "things":
                        pred_class = pred_class - len(meta["thing_classes"]) + 1
                    segments_info.append({"id": current_segment_id, "isthing": bool(isthing), "category_id": int(pred_class)})
        return panoptic_seg, segments_info

    @torch.no_grad()
    def forward(self, image, text_feats, height=None, width=None, forced_topk=None, with_masks=True, prompt="name",
                phrase_bank=256, semantic=None, detector_columns=None, panoptic=None, name_fusion_text=False, mask_prompt=None):
        """prompt="phrase" (also "expression" with text_feature_reduce_before_fusion): the text bank, zero-padded to
        the phrase-bank size (:304-327 with text_feature_bank + text_feature_bank_reset), is FUSED with the vision
        tokens in the encoder and the fused tokens are the classifier's vocabulary (:356-358, 448)."""
        S = self.stages = {}
        images, img_mask, (h, w) = self.preprocess(image)
        height = height or h
        width = width or w
        features_l = text_feats.float()[None]                       # [1,K,1024]  (:279-280)
        if not self.vl:
            fusion = None                                                # deformable_detr_segm.py: no fusion tokens at all
        elif prompt == "name" and name_fusion_text:
            fusion = features_l                                          # (:343-347) name_prompt_fusion_text[dataset_id]
        elif prompt == "name":
            fusion = self.p("name_prompt_fusion_feature").repeat(1, 1, 1)  # zeros [1,1,1024] (:349-352)
        else:
            K = text_feats.shape[0]
            # phrase_bank: int = zero rows (text_feature_bank_reset, :321-327); tensor = the persistent bank's rows (:310-320);
            # 0 = no bank (free-text prompts with the default config)
            extra = torch.zeros(phrase_bank, text_feats.shape[1]) if isinstance(phrase_bank, int) else phrase_bank.float()
            bank = torch.cat([text_feats.float(), extra], 0)[: max(K, extra.shape[0])]
            features_l = bank[None]                                  # (:321-335)
            fusion = features_l + 0.0 * self.p("name_prompt_fusion_feature")   # (:357-360)
        feat = self.vit(images)
        S["last_feat"] = feat
        fpn = self.fpn(feat)
        for k_, v_ in fpn.items():
            S[k_] = v_
        # neck = None (APE-L_A/B/C, ape_deta_vitl_eva02_lsj1024_cp_12ep.py:21): the pyramid maps feed the transformer directly
        ml_feats = self.neck(fpn) if (self.prefix + "neck.convs.0.conv.weight") in self.sd else [fpn[k_] for k_ in ("p2", "p3", "p4", "p5", "p6")]
        masks, pos = [], []
        for f in ml_feats:
            masks.append(F.interpolate(img_mask[None], size=f.shape[-2:]).to(torch.bool).squeeze(0))
            pos.append(tp.position_embedding_sine(masks[-1]))
        masks_prompt = None
        if mask_prompt is not None:                      # deformable_detr_segm_vl.py:394-414
            mp, _ = tp.pad_to_square(mask_prompt.float()[None], self.cfg["img_size"])
            if mp.sum() == 0:
                mp[...] = 255
            masks_prompt = [F.interpolate(mp[None], size=f.shape[-2:], mode="bilinear").to(torch.bool).squeeze(0) for f in ml_feats]
        (inter, init_ref, inter_ref, enc_class, enc_coord, anchors, memory, l_out,
         spatial_shapes) = self.transformer(ml_feats, masks, pos, fusion, forced_topk, masks_prompt)
        S["inter_states"], S["inter_references"], S["init_reference"] = inter, inter_ref, init_ref
        mask_feat = self.mask_features(memory, fpn["p2"], spatial_shapes)
        S["mask_features"] = mask_feat
        if not self.vl:
            pass                                                     # the classifier sees the raw text bank
        elif prompt == "name":
            features_l = 1.0 * features_l + 0.0 * l_out              # (:446)
        else:
            features_l = 0.0 * features_l + 1.0 * l_out              # (:448)
        lvl = self.dec_layers - 1                                   # only the last level is consumed (:519-524)
        reference = tp.inverse_sigmoid(init_ref if lvl == 0 else inter_ref[lvl - 1])
        logits = self.vl_align(inter[lvl], features_l, f"class_embed.{lvl}")
        coord = (self.mlp(inter[lvl], f"bbox_embed.{lvl}") + reference).sigmoid()
        membed = self.mlp(inter[lvl], "mask_embed")
        pred_masks = torch.einsum("bqc,bchw->bqhw", membed, mask_feat)
        S["pred_logits"], S["pred_boxes"], S["pred_masks"] = logits, coord, pred_masks
        out = {"pred_logits": logits, "pred_boxes": coord, "pred_masks": pred_masks}
        det_logits = logits[0] if detector_columns is None else logits[0][:, :detector_columns]   # thing columns (:578-590)
        boxes, scores, classes, qidx = self.inference(det_logits, coord[0], (h, w))
        S["det_boxes"], S["det_scores"], S["det_classes"], S["det_query"] = boxes, scores, classes, qidx
        masks128 = None
        if with_masks:
            # :569-572 (only the kept queries: bilinear interpolation is per channel) and :600-613
            up = F.interpolate(pred_masks[:, qidx], size=images.shape[-2:], mode="bilinear", align_corners=False)[0]
            bit = up.sigmoid() > 0.5
            masks128 = tp.bitmasks_crop_and_resize(bit, boxes, 128)
            S["det_masks128"] = masks128
        fb, fs, fc, fm, keep = tp.detector_postprocess(boxes, scores, classes, masks128, (h, w), height, width)
        out["instances"] = {"pred_boxes": fb, "scores": fs, "pred_classes": fc, "pred_masks": fm, "query_index": qidx[keep]}
        if semantic is not None:
            out["sem_seg"] = S["sem_seg"] = self.semantic_branch(logits, coord, pred_masks, images.shape[-2:], (h, w), height, width,
                                                                 semantic)
        if panoptic is not None:
            out["panoptic_seg"] = self.panoptic_branch(logits, coord, pred_masks, images.shape[-2:], (h, w), height, width,
                                                       panoptic["meta"], panoptic["cfg"])
        return out
