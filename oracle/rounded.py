"""The oracle AT THE HIP PIPELINE'S ROUNDING POINTS (tier T2: "bf16 kernels vs the oracle run at the same bf16
rounding points").

TEST INFRASTRUCTURE ONLY -- the checker, never the product: tests/, __graft_entry__.smoke() and bench.py's `parity` leg import this file;
nothing under ape_amd/ does.

`ApeOracle` (oracle/ape_oracle.py) restates the reference in fp32 and is pinned to the reference's own code.  The benchmarked pipeline
stores its activations and GEMM operands in a 16-bit type (bf16, or IEEE half), so it cannot be within 1e-3 of an fp32 run end to end --
the reference's own modules under bf16 autocast are not either.  What CAN be asked of it, and is asked here: fed the same input, every
stage of the 16-bit HIP pipeline must agree with the same algorithm evaluated in fp32 arithmetic WITH A ROUNDING AT EXACTLY THE POINTS
WHERE THE PIPELINE STORES A 16-BIT TENSOR -- what is left between the two is accumulation order, fp32 transcendental accuracy and the
rare rounding flip they cause (<= 2e-3 relative rms per stage, asserted by tests/test_same_rounding.py on the MI355X).

`RoundedApeOracle` subclasses the fp32 oracle and overrides the stages of the APE-L_D path (EVA-02-CLIP ViT with sub-LN / SwiGLU / RoPE,
SimpleFPN, neck, vision-language encoder in name-prompt mode, two-stage heads, decoder, heads, mask features, semantic branch).  Every
override cites BOTH the reference lines it restates and the host / kernel code whose storage points it mirrors.  The storage points:

  * GEMM operands: activations and weights are 16-bit (ape_amd/packing.py pack_matrix); accumulation, bias, RoPE, activation, residual
    in fp32; ONE rounding at the store (include/ape_hip.h "epilogue order").  Biases, norm parameters, RoPE tables, position embeddings of
    the ViT stay fp32.
  * ViT residual stream fp32 (the last block's output is stored 16-bit); the outputs of norm1 / norm2 16-bit; attention probabilities are rounded
    before P.V and the row sum is taken over the ROUNDED probabilities (csrc/attention.hip: the sum is an MFMA of the packed tile with ones).
  * the attention's inner LayerNorm folded into its output projection the same way (round 6, Block._folded_inner_ln): statistics of the
    STORED attention output, gamma inside the rounded weight; no store between attention and projection.
  * SwiGLU sub-LayerNorm folded into the down projection: LN(h) W3^T = rstd (h W'^T) - rstd mean rowsum(W') + W3 beta with W' = W3 diag(gamma)
    ROUNDED to 16 bits and rowsum taken of the rounded W' (vit_eva_clip.py Block._folded_subln).
  * encoder / decoder streams 16-bit (post-norm layers: the residual is the LayerNorm output); the deformable attention's value projection
    and -- for >= 2048 queries -- its offsets | logits are IEEE HALF in both flavours (layers/multi_scale_deform_attn.py), saturating.
  * LayerNorm in the epilogue of the producing kernel (no rounding between linear and norm) where the product uses those kernels:
    the attention's output projection + first norm (csrc/gemm.hip gemm_kres_ln_kernel) and the FFN + last norm (csrc/ffn_fused.hip) for
    >= 2048 tokens; separate launches (one more rounding) below that.
  * name-prompt vision-language fusion in its single-token form (layers/fuse_helper.py forward_tokens_single): the language side is fp32
    except the [T, 8] score GEMM, whose operands are the 16-bit vision stream and a 16-bit copy of W_v^T k.

Layouts: this class keeps the oracle's layouts (raster token order, batch-first); `hip_stages()` returns the stage tensors under the HIP
pipeline's stage names in ITS layouts (token-major; ViT stages in raster order -- the test permutes them to window-major).
"""
import math

import torch
import torch.nn.functional as F

from . import thirdparty as tp
from .ape_oracle import ApeOracle, msda_core, rotate_half, window_partition, window_unpartition

HALF_MAX = 65504.0


class RoundedApeOracle(ApeOracle):
    FUSED_MIN_ROWS = 2048            # gemm_kres_ln / ffn_fused / half offsets exist from 2048 rows on (ops.gemm_norm_fusable, FFN.FUSED_MIN_ROWS)

    def __init__(self, cfg, state_dict, dtype=torch.bfloat16, prefix="model_vision."):
        super().__init__(cfg, state_dict, prefix)
        if cfg.get("backbone") is not None or not self.vl:
            raise NotImplementedError("RoundedApeOracle covers the APE-L_D path (EVA-02-CLIP ViT, vision-language encoder)")
        assert dtype in (torch.bfloat16, torch.float16)
        self.dt = dtype
        self._w = {}
        self.hip = {}
        self.debug = None                # set to {} to record per-op intermediates of the decoder layers

    # ------------------------------------------------------------------------------------------------ rounding primitives
    def R(self, x):
        """one store in the pipeline's 16-bit type (IEEE half stores saturate: csrc/common.h pack2h / stf<f16_t>)"""
        if self.dt == torch.float16:
            x = x.clamp(-HALF_MAX, HALF_MAX)
        return x.to(self.dt).float()

    @staticmethod
    def Rh(x):
        """one store in IEEE half (deformable attention values, encoder offsets | logits), saturating"""
        return x.clamp(-HALF_MAX, HALF_MAX).to(torch.float16).float()

    def W(self, name):
        """a GEMM weight as packed: rounded once (packing.pack_matrix)"""
        if name not in self._w:
            self._w[name] = self.R(self.p(name))
        return self._w[name]

    def rlin(self, x, name, bias=True):
        """fp32 accumulate over 16-bit operands + fp32 bias; the CALLER rounds the result where the pipeline stores it"""
        return F.linear(x, self.W(name + ".weight"), self.p(name + ".bias") if bias else None)

    # ------------------------------------------------------------------------------------------------ a2-a5: ViT
    def vit_attention(self, xn, i, rope, key_order=None):
        """Attention.forward (vit_eva_clip.py:218-268) at the storage points of Block._attention (ape_amd/modeling/backbone/vit_eva_clip.py):
        q|k = store(rope(xn Wqk^T + b)), V^T = store(xn Wv^T + b), flash attention on 16-bit q, k, v with 16-bit probabilities, its
        output stored, then the out projection with the inner LayerNorm folded in (fp32; the caller adds the residual stream)"""
        pre = f"backbone.net.blocks.{i}.attn."
        B, H, W, C = xn.shape
        N = H * W
        x = xn.reshape(B, N, C)
        nh = self.num_heads_vit
        q = F.linear(x, self.W(pre + "q_proj.weight"), self.p(pre + "q_bias"))
        k = F.linear(x, self.W(pre + "k_proj.weight"), None)
        v = self.R(F.linear(x, self.W(pre + "v_proj.weight"), self.p(pre + "v_bias")))
        q = q.reshape(B, N, nh, -1).permute(0, 2, 1, 3)
        k = k.reshape(B, N, nh, -1).permute(0, 2, 1, 3)
        v = v.reshape(B, N, nh, -1).permute(0, 2, 1, 3)
        cos, sin = rope
        q = self.R(q * cos + rotate_half(q) * sin)
        k = self.R(k * cos + rotate_half(k) * sin)
        if key_order is not None:               # global blocks: the pipeline keeps its tokens window-major (ViT.token_order), and a key's
            k, v = k[:, :, key_order], v[:, :, key_order]          # tile in the flash loop follows the STORED order
        o = self.attention16(q, k, v, q.shape[-1] ** -0.5)
        o = o.permute(0, 2, 1, 3).reshape(B, N, -1)
        # round 6: the inner LayerNorm is folded into the out projection (Block._folded_inner_ln): statistics of the STORED attention output,
        # gamma inside the rounded weight -- no store between the attention and the projection any more
        key = pre + "projf"
        if key not in self._w:
            wp, g, b = self.p(pre + "proj.weight"), self.p(pre + "inner_attn_ln.weight"), self.p(pre + "inner_attn_ln.bias")
            wpf = self.R(wp * g[None, :])
            self._w[key] = (wpf, wpf.sum(dim=1), wp @ b + self.p(pre + "proj.bias"))
        wpf, c1, c2 = self._w[key]
        mean = o.mean(dim=-1, keepdim=True)
        rstd = torch.rsqrt(o.var(dim=-1, unbiased=False, keepdim=True) + 1e-6)
        y = (o @ wpf.t()) * rstd + (-mean * rstd) * c1 + c2
        return y.view(B, H, W, C)

    def attention16(self, q, k, v, scale, tile=64):
        """softmax(scale q k^T) v on 16-bit operands as csrc/attention.hip evaluates it: a flash loop over KEY TILES of 64 in the kernel's key
        order -- scores in fp32, a running row maximum m (in the exp2 domain: scale * log2 e folded into one fma), the tile's probabilities
        p = 2^(s c - m_new) ROUNDED for the P.V product, the row sum accumulated over those rounded probabilities (an MFMA with a ones
        tile), earlier sums rescaled by 2^(m_old - m_new) in fp32, the quotient stored.  The rounding of a probability is therefore
        relative to the maximum SEEN SO FAR, not to the row's final maximum: with a peaked softmax that is a different realisation of the
        16-bit rounding (3e-4 of the attention output at the decoder's 900 queries, measured), so the tile structure is part of the
        kernel's rounding points.  q, k, v: [B, heads, N, d] fp32 holding 16-bit values; keys in the order the pipeline stores them."""
        c = scale * 1.4426950408889634
        B, nh, Nq, _ = q.shape
        Nk = k.shape[2]
        out = torch.empty((B, nh, Nq, v.shape[-1]), dtype=torch.float32)
        step = max(1, (1 << 25) // max(1, Nq * tile))                             # heads per chunk
        for h0 in range(0, nh, step):
            sl = slice(h0, h0 + step)
            qh = q[:, sl]
            m = torch.full((B, qh.shape[1], Nq, 1), float("-inf"))
            l = torch.zeros((B, qh.shape[1], Nq, 1))
            o = torch.zeros((B, qh.shape[1], Nq, v.shape[-1]))
            for t0 in range(0, Nk, tile):
                st = qh @ k[:, sl, t0:t0 + tile].transpose(-2, -1)                 # raw scores of this key tile
                m_new = torch.maximum(m, st.amax(dim=-1, keepdim=True) * c)
                alpha = torch.exp2(m - m_new)
                pt = self.R(torch.exp2(st * c - m_new))
                l = l * alpha + pt.sum(dim=-1, keepdim=True)
                o = o * alpha + pt @ v[:, sl, t0:t0 + tile]
                m = m_new
            out[:, sl] = self.R(o / l)
        return out

    def vit_block(self, x, i, last=False):
        """Block.forward (vit_eva_clip.py:505-523) at the storage points of Block.forward_tokens: x fp32 stream in, fp32 stream out
        (16-bit for the last block)"""
        pre = f"backbone.net.blocks.{i}."
        xn = self.R(self.ln(x, pre + "norm1", 1e-6))
        if i in self.win_blocks:
            H, W = xn.shape[1], xn.shape[2]
            o = window_unpartition(self.vit_attention(window_partition(xn, self.ws), i, self.rope_win), self.ws, H, W)
        else:
            H, W = xn.shape[1], xn.shape[2]
            order = torch.arange(H * W).view(H // self.ws, self.ws, W // self.ws, self.ws).permute(0, 2, 1, 3).reshape(-1)
            o = self.vit_attention(xn, i, self.rope_glb, key_order=order)
        x = x + o                                                                        # (the folded out projection is inside vit_attention) fp32 residual epilogue, fp32 store
        h = self.R(self.ln(x, pre + "norm2", 1e-6))
        hidden = self.R(F.silu(self.rlin(h, pre + "mlp.w1")) * self.rlin(h, pre + "mlp.w2"))      # fused SwiGLU epilogue, one store
        # sub-LayerNorm folded into the down projection (Block._folded_subln): statistics of the STORED hidden activation
        key = pre + "mlp.w3f"
        if key not in self._w:
            w3, g, b = self.p(pre + "mlp.w3.weight"), self.p(pre + "mlp.ffn_ln.weight"), self.p(pre + "mlp.ffn_ln.bias")
            w3f = self.R(w3 * g[None, :])
            self._w[key] = (w3f, w3f.sum(dim=1), w3 @ b + self.p(pre + "mlp.w3.bias"))
        w3f, c1, c2 = self._w[key]
        mean = hidden.mean(dim=-1, keepdim=True)
        rstd = torch.rsqrt(hidden.var(dim=-1, unbiased=False, keepdim=True) + 1e-6)
        y = (hidden @ w3f.t()) * rstd + (-mean * rstd) * c1 + c2 + x
        return self.R(y) if last else y

    def vit(self, images):
        """PatchEmbed + abs-pos + blocks (vit_eva_clip.py:743-754): normalised pixels and the patch-embedding weight are 16-bit operands,
        the position embedding rides as the fp32 residual of that GEMM (ViT.forward_tokens)"""
        x = F.conv2d(self.R(images), self.W("backbone.net.patch_embed.proj.weight"), self.p("backbone.net.patch_embed.proj.bias"), stride=16)
        x = x.permute(0, 2, 3, 1)
        x = x + self.abs_pos((x.shape[1], x.shape[2]))
        self.stages["vit_embed"] = x
        self.hip["vit_embed"] = x[0].reshape(-1, x.shape[-1])
        import time
        for i in range(self.depth):
            t0 = time.perf_counter()
            x = self.vit_block(x, i, last=(i == self.depth - 1))
            self._tick("vit_win_block" if i in self.win_blocks else "vit_glb_block", t0)
            self.stages[f"vit_block{i}"] = x
            self.hip[f"vit_blk{i}"] = x[0].reshape(-1, x.shape[-1])
        return x.permute(0, 3, 1, 2)

    # ------------------------------------------------------------------------------------------------ a6-a7: pyramid, neck
    def conv_ln(self, x, name, k):
        """detectron2 Conv2d(bias=False, norm=LN) (vit_eva_clip.py:806-842): conv output stored, channel-LayerNorm output stored
        (SimpleFeaturePyramid._conv_ln_pair)"""
        x = self.R(F.conv2d(x, self.W(name + ".weight"), None, padding=k // 2))
        return self.R(tp.layer_norm_2d(x, self.p(name + ".norm.weight"), self.p(name + ".norm.bias"), 1e-6))

    def fpn(self, feat):
        """SimpleFeaturePyramid.forward (vit_eva_clip.py:871-922) at the storage points of SimpleFeaturePyramid.forward_tokens"""
        pb = "backbone."
        x = self.R(F.conv_transpose2d(feat, self.W(pb + "simfp_2.0.weight"), self.p(pb + "simfp_2.0.bias"), stride=2))
        x = self.R(F.gelu(tp.layer_norm_2d(x, self.p(pb + "simfp_2.1.weight"), self.p(pb + "simfp_2.1.bias"), 1e-6)))
        x = self.R(F.conv_transpose2d(x, self.W(pb + "simfp_2.3.weight"), self.p(pb + "simfp_2.3.bias"), stride=2))
        p2 = self.conv_ln(self.conv_ln(x, pb + "simfp_2.4", 1), pb + "simfp_2.5", 3)
        x = self.R(F.conv_transpose2d(feat, self.W(pb + "simfp_3.0.weight"), self.p(pb + "simfp_3.0.bias"), stride=2))
        p3 = self.conv_ln(self.conv_ln(x, pb + "simfp_3.1", 1), pb + "simfp_3.2", 3)
        p4 = self.conv_ln(self.conv_ln(feat, pb + "simfp_4.0", 1), pb + "simfp_4.1", 3)
        x = F.max_pool2d(feat, kernel_size=2, stride=2)
        p5 = self.conv_ln(self.conv_ln(x, pb + "simfp_5.1", 1), pb + "simfp_5.2", 3)
        p6 = tp.last_level_max_pool(p5)
        out = {"p2": p2, "p3": p3, "p4": p4, "p5": p5, "p6": p6}
        for k_, v_ in out.items():
            self.hip[k_] = v_[0].permute(1, 2, 0).reshape(-1, v_.shape[1])
        return out

    def neck(self, feats):
        """detrex ChannelMapper: 1x1 conv (+ bias) stored, GroupNorm(32) stored (DeformableDETRSegmVL.forward_single neck_level)"""
        outs = []
        for i, f in enumerate(["p2", "p3", "p4", "p5", "p6"]):
            x = self.R(F.conv2d(feats[f], self.W(f"neck.convs.{i}.conv.weight"), self.p(f"neck.convs.{i}.conv.bias")))
            outs.append(self.R(F.group_norm(x, 32, self.p(f"neck.convs.{i}.norm.weight"), self.p(f"neck.convs.{i}.norm.bias"), 1e-5)))
        return outs

    # ------------------------------------------------------------------------------------------------ a10-a12: one encoder layer
    def vl_fusion_name(self, x, l, i, lvl_pos):
        """BiAttentionBlock.forward with ONE language token (fuse_helper.py:67-166, 221-232; name prompts, deformable_detr_segm_vl.py:349-352)
        in the form layers/fuse_helper.py forward_tokens_single evaluates: -> (v_new stored, v_new + pos stored, l_new fp32)"""
        pre = f"transformer.encoder.vl_layers.{i}.b_attn."
        nh, E = 8, 2048
        hd = E // nh
        scale = hd ** -0.5
        l_n = self.ln(l, pre + "layer_norm_l")                                             # fp32 language side
        k = self.lin(l_n, pre + "attn.l_proj")[0, 0]                                       # [E]
        vl = self.lin(l_n, pre + "attn.values_l_proj")
        gdv = (self.p(pre + "gamma_v") * self.lin(vl, pre + "attn.out_v_proj"))[0, 0]      # softmax over one token == 1   [256]
        v32 = self.ln(x, pre + "layer_norm_v") + gdv                                       # LayerNorm with bias beta + gamma_v delta_v
        v_new, qp = self.R(v32), self.R(v32 + lvl_pos)
        # language update: scores on LN_v(v) = v_new - gdv; the [T, 8] score GEMM runs on the STORED vision stream and a 16-bit copy of
        # u_h = W_v,h^T k_h; the bias term uses the fp32 u (ops.gemv)
        wv = self.p(pre + "attn.v_proj.weight").view(nh, hd, -1)                           # [h, hd, 256]
        u = torch.einsum("hd,hdc->hc", k.view(nh, hd), wv)                                 # [8, 256]
        c = scale * (self.p(pre + "attn.v_proj.bias").view(nh, hd) * k.view(nh, hd)).sum(-1)
        sbias = c - scale * (u @ gdv)
        S = scale * (v_new[0] @ self.R(u).t()) + sbias                                     # [T, 8]
        w = (S - S.max()).clamp(-50000, 50000)                                             # stable_softmax_2d (:89-90), clamps (:93-98)
        wl = (w - w.max(dim=0, keepdim=True)[0]).clamp(-50000, 50000).softmax(dim=0)       # over the vision tokens, padding NOT masked (:101-116)
        pooled = wl.t() @ v_new[0] - gdv[None, :]                                          # sum_t p[t,h] LN_v(v)[t]       [8, 256]
        wvv = self.p(pre + "attn.values_v_proj.weight").view(nh, hd, -1)
        ol = torch.einsum("hc,hdc->hd", pooled, wvv) + self.p(pre + "attn.values_v_proj.bias").view(nh, hd)
        dl = self.lin(ol.reshape(1, 1, E), pre + "attn.out_l_proj")
        l_new = l_n + self.p(pre + "gamma_l") * dl
        return v_new, qp, l_new

    def msda_rounded(self, pre, qp, value_src, identity, key_padding_mask, reference_points, spatial_shapes, value=None, norm=None):
        """MultiScaleDeformableAttention.forward (multi_scale_deform_attn.py:215-358) at the storage points of
        layers/multi_scale_deform_attn.py forward_tokens: value half, offsets | logits half from 2048 queries on (fp32 below), sampler
        output stored, output projection + identity (+ the following LayerNorm in the same kernel from 2048 rows on) stored"""
        bs, nq, _ = qp.shape
        if value is None:
            value = self.Rh(self.rlin(value_src, pre + "value_proj")) if self.dt == torch.bfloat16 and value_src.shape[1] >= self.FUSED_MIN_ROWS \
                else self.R(self.rlin(value_src, pre + "value_proj"))
            if key_padding_mask is not None:
                value = value.masked_fill(key_padding_mask[..., None], 0.0)
        nv = value.shape[1]
        value = value.view(bs, nv, 8, -1)
        L = len(spatial_shapes)
        off = self.rlin(qp, pre + "sampling_offsets")
        logit = self.rlin(qp, pre + "attention_weights")
        if nq >= self.FUSED_MIN_ROWS:
            off, logit = self.Rh(off), self.Rh(logit)
        off = off.view(bs, nq, 8, L, 4, 2)
        aw = logit.view(bs, nq, 8, L * 4).softmax(-1).view(bs, nq, 8, L, 4)
        if reference_points.shape[-1] == 2:
            nrm = torch.tensor([[w, h] for h, w in spatial_shapes], dtype=torch.float32)
            loc = reference_points[:, :, None, :, None, :] + off / nrm[None, None, None, :, None, :]
        else:
            loc = reference_points[:, :, None, :, None, :2] + off / 4 * reference_points[:, :, None, :, None, 2:] * 0.5
        samp = self.R(msda_core(value, spatial_shapes, loc, aw))
        y = self.rlin(samp, pre + "output_proj") + identity
        if norm is None:
            return self.R(y)
        if nq >= self.FUSED_MIN_ROWS:                       # gemm_kres_ln_kernel: LayerNorm of the fp32 sums
            return self.R(self.ln(y, norm))
        return self.R(self.ln(self.R(y), norm))

    def ffn_rounded(self, x, pre, norm):
        """detrex FFN + the layer's last norm (deformable_transformer_vl.py:45-54): hidden activation stored; from 2048 rows on the second
        linear, the identity and the LayerNorm are one kernel (csrc/ffn_fused.hip), below that the sum is stored before the norm"""
        h = self.R(F.relu(self.rlin(x, pre + "layers.0.0")))
        y = self.rlin(h, pre + "layers.1") + x
        if x.shape[1] >= self.FUSED_MIN_ROWS:
            return self.R(self.ln(y, norm))
        return self.R(self.ln(self.R(y), norm))

    # ------------------------------------------------------------------------------------------------ a13: two-stage heads
    def mlp(self, x, pre, n=3):
        """detrex MLP at the storage points of _containers.MLP.forward_tokens: hidden layers stored; the last layer is fp32 for the box
        heads and stored for the mask embedding"""
        for j in range(n):
            x = self.rlin(x, f"{pre}.layers.{j}")
            if j < n - 1:
                x = self.R(F.relu(x))
        return self.R(x) if pre.startswith("mask_embed") else x

    def gen_proposals(self, memory, mask_flat, spatial_shapes, mask_prompt_flat=None):
        """gen_encoder_output_proposals (:321-369) + enc_output / enc_output_norm at the storage points of
        DeformableDetrTransformerVL.forward_tokens: enc_output stored, its LayerNorm stored"""
        ln_name, lin_name = "transformer.enc_output_norm", "transformer.enc_output"
        # reuse the fp32 geometry of the parent through a probe that captures the masked memory instead of projecting it
        captured = {}
        orig_lin, orig_ln = self.lin, self.ln

        def lin(x, name, bias=True):
            if name == lin_name:
                captured["om"] = x
                return x
            return orig_lin(x, name, bias)

        def ln(x, name, eps=1e-5):
            return x if name == ln_name else orig_ln(x, name, eps)

        self.lin, self.ln = lin, ln
        try:
            _, prop, level_ids = super().gen_proposals(memory, mask_flat, spatial_shapes, mask_prompt_flat)
        finally:
            del self.lin, self.ln                 # back to the class's methods
        om = self.R(self.rlin(captured["om"], lin_name))
        om = self.R(self.ln(om, ln_name))
        return om, prop, level_ids

    # ------------------------------------------------------------------------------------------------ a17: classifier
    def vl_align(self, x, emb, pre):
        """VisionLanguageAlign.forward (vision_language_align.py:27-52): the projected text tokens are an fp32 product stored 16-bit
        (layers/vision_language_align.py text_side), the logits fp32"""
        e = F.normalize(emb, p=2, dim=-1)
        tok = self.R(self.lin(e / 2.0, pre + ".dot_product_projection_text"))
        bias = torch.matmul(e, self.p(pre + ".bias_lang")) + self.p(pre + ".bias0")
        logit = torch.matmul(x, tok.transpose(-1, -2)) / self.p(pre + ".log_scale").exp() + bias.unsqueeze(1)
        return logit.clamp(max=50000).clamp(min=-50000)

    # ------------------------------------------------------------------------------------------------ a9-a16: transformer
    def transformer(self, feats, masks, pos_embeds, query_l, forced_topk=None, masks_prompt=None):
        """DeformableDetrTransformerVL.forward (deformable_transformer_vl.py:422-699), name-prompt mode, at the storage points of
        ape_amd/modeling/ape_deta/deformable_transformer_vl.py"""
        S, Hs = self.stages, self.hip
        if query_l.shape[1] != 1 or masks_prompt is not None:
            raise NotImplementedError("RoundedApeOracle: name prompts (one fusion token), no mask prompt")
        spatial_shapes = [(f.shape[2], f.shape[3]) for f in feats]
        feat = torch.cat([f.flatten(2).transpose(1, 2) for f in feats], 1)
        mask = torch.cat([m.flatten(1) for m in masks], 1)
        lvl_pos = self.R(torch.cat([p.flatten(2).transpose(1, 2) + self.p("transformer.level_embeds")[i].view(1, 1, -1)
                                    for i, p in enumerate(pos_embeds)], 1))                  # stored once per size (lvl_pos())
        valid_ratios = torch.stack([self.valid_ratio(m) for m in masks], 1)
        ref = self.encoder_reference_points(spatial_shapes, valid_ratios)
        S["enc_input"], S["lvl_pos"], S["valid_ratios"] = feat, lvl_pos, valid_ratios
        Hs["enc_input"] = feat[0]
        import time
        x, l = feat, query_l
        for i in range(self.enc_layers):
            t0 = time.perf_counter()
            v_new, qp, l = self.vl_fusion_name(x, l, i, lvl_pos)
            pre = f"transformer.encoder.layers.{i}."
            x2 = self.msda_rounded(pre + "attentions.0.", qp, v_new, v_new, mask, ref, spatial_shapes, norm=pre + "norms.0")
            x = self.ffn_rounded(x2, pre + "ffns.0.", pre + "norms.1")
            self._tick("enc_layer", t0)
            S[f"enc{i}_fused_v"], S[f"enc{i}_fused_l"], S[f"enc{i}_out"] = v_new, l, x
            Hs[f"enc{i}_fused_l"], Hs[f"enc{i}_out"] = l[0], x[0]
        memory = x
        S["memory"], S["query_l"] = memory, l
        Hs["memory"] = memory[0]

        # two-stage heads (:495-533) -- main + ambiguous
        om, props, level_ids = self.gen_proposals(memory, mask, spatial_shapes, None)
        nd = self.dec_layers
        cls = self.rlin(om, f"transformer.decoder.class_embed.{nd}")
        d = self.mlp(om, f"transformer.decoder.bbox_embed.{nd}")
        cls_a = self.rlin(om, "transformer.decoder.class_embed_ambiguous.0")
        d_a = self.mlp(om, "transformer.decoder.bbox_embed_ambiguous.0")
        cls2 = torch.stack([cls, cls_a], dim=1)
        box2 = torch.stack([d + props, d_a + props], dim=1)
        idx = torch.argmax(cls2, dim=1, keepdim=True)
        enc_class = torch.gather(cls2, 1, idx).squeeze(1)
        enc_coord = torch.gather(box2, 1, idx.repeat(1, 1, 1, 4)).squeeze(1)
        S["output_memory"], S["enc_class"], S["enc_coord_unact"] = om, enc_class, enc_coord
        Hs["output_memory"], Hs["enc_cls2"], Hs["enc_delta8"] = om[0], torch.cat([cls[0], cls_a[0]], 1), torch.cat([d[0], d_a[0]], 1)
        Hs["enc_class"], Hs["enc_coord_unact"] = enc_class[0, :, 0], enc_coord[0]

        logit = enc_class[..., 0]
        boxes = tp.box_cxcywh_to_xyxy(enc_coord.sigmoid()).clamp(0, 1)
        if forced_topk is not None:
            topk = forced_topk
        else:
            topk = torch.stack([self.select_proposals(logit[b], boxes[b], level_ids, len(spatial_shapes)) for b in range(feat.shape[0])])
        S["topk_proposals"] = topk

        # query initialisation (:629-645): sine embedding stored, both linears fp32, LayerNorms + split + add, stored (query_finish)
        coords = torch.gather(enc_coord, 1, topk.unsqueeze(-1).repeat(1, 1, 4))
        reference = coords.sigmoid()
        init_reference = reference
        pe = self.R(self.proposal_pos_embed(coords))
        pt = self.ln(self.rlin(pe, "transformer.pos_trans"), "transformer.pos_trans_norm")
        qpos32, q32 = torch.split(pt, 256, dim=2)
        feats_topk = torch.stack([om[b][topk[b]] for b in range(om.shape[0])])
        query_pos = self.R(qpos32)
        query = self.R(q32 + self.ln(self.rlin(feats_topk, "transformer.pix_trans"), "transformer.pix_trans_norm"))
        S["query_init"], S["query_pos"] = query, query_pos
        Hs["query_init"], Hs["query_pos"], Hs["init_reference"] = query[0], query_pos[0], reference[0]

        # decoder (:195-250).  value_proj of every layer over the stored memory (one GEMM in the product), half, padded rows zero
        inter, inter_ref = [], []
        out = query
        E = 256
        for i in range(self.dec_layers):
            t0 = time.perf_counter()
            ref_in = reference[:, :, None] * torch.cat([valid_ratios, valid_ratios], -1)[:, None]
            pre = f"transformer.decoder.layers.{i}."
            outp = self.R(out + query_pos)
            w_in, b_in = self.W(pre + "attentions.0.attn.in_proj_weight"), self.p(pre + "attentions.0.attn.in_proj_bias")
            q = self.R(F.linear(outp, w_in[:E], b_in[:E]))
            k = self.R(F.linear(outp, w_in[E:2 * E], b_in[E:2 * E]))
            v = self.R(F.linear(out, w_in[2 * E:], b_in[2 * E:]))
            B, Q, _ = q.shape

            def heads(t):
                return t.view(B, Q, 8, 32).transpose(1, 2)

            sa = self.attention16(heads(q), heads(k), heads(v), 32 ** -0.5).transpose(1, 2).reshape(B, Q, E)
            x1 = self.R(out + self.rlin(sa, pre + "attentions.0.attn.out_proj"))
            n0 = self.ln(x1, pre + "norms.0")
            x2, x2p = self.R(n0), self.R(n0 + query_pos)                                # layernorm kernel: y and y + add from the fp32 y
            apre = pre + "attentions.1."
            val = self.rlin(memory, apre + "value_proj")
            val = (self.Rh(val) if self.dt == torch.bfloat16 and memory.shape[1] >= self.FUSED_MIN_ROWS else self.R(val)).masked_fill(mask[..., None], 0.0)
            x3 = self.msda_rounded(apre, x2p, None, x2, None, ref_in, spatial_shapes, value=val)
            x4 = self.R(self.ln(x3, pre + "norms.1"))
            if self.debug is not None:          # intermediates of a decoder layer (tools/gpu_t2_decoder_probe.py localises a mismatching op)
                self.debug[i] = dict(outp=outp[0], q=q[0], k=k[0], v=v[0], sa=sa[0], x1=x1[0], x2=x2[0], x2p=x2p[0], val=val[0], x3=x3[0], x4=x4[0],
                                     ref_in=ref_in[0])
            out = self.ffn_rounded(x4, pre + "ffns.0.", pre + "norms.2")
            tmp = self.mlp(out, f"transformer.decoder.bbox_embed.{i}")
            reference = (tmp + tp.inverse_sigmoid(reference)).sigmoid()
            inter.append(out)
            inter_ref.append(reference)
            Hs[f"dec{i}_out"], Hs[f"dec{i}_delta"], Hs[f"dec{i}_ref"] = out[0], tmp[0], reference[0]
            self._tick("dec_layer", t0)
        return (torch.stack(inter), init_reference, torch.stack(inter_ref), enc_class, enc_coord, props.sigmoid(), memory, l, spatial_shapes)

    # ------------------------------------------------------------------------------------------------ a18: mask features
    def mask_features(self, memory, p2, spatial_shapes):
        """maskdino_mask_features (deformable_detr_segm_vl.py:728-750) at the storage points of forward_single.mask_features: every conv
        output stored, GroupNorm (+ encoder memory / ReLU) stored"""
        h, w = spatial_shapes[0]
        enc = memory[:, : h * w, :].permute(0, 2, 1).reshape(1, -1, h, w)
        x = self.R(F.conv2d(p2, self.W("lateral_conv.weight")))
        x = F.group_norm(x, 32, self.p("lateral_conv.norm.weight"), self.p("lateral_conv.norm.bias"), 1e-5)
        x = self.R(x + F.interpolate(enc, size=x.shape[-2:], mode="bilinear", align_corners=False))
        x = self.R(F.conv2d(x, self.W("output_conv.weight"), padding=1))
        x = self.R(F.relu(F.group_norm(x, 32, self.p("output_conv.norm.weight"), self.p("output_conv.norm.bias"), 1e-5)))
        mf = self.R(F.conv2d(x, self.W("mask_conv.weight")))
        self.hip["mask_features"] = mf[0].permute(1, 2, 0).reshape(-1, mf.shape[1])
        return mf

    # ------------------------------------------------------------------------------------------------ a22: semantic branch
    def semantic_branch(self, logits, coord, pred_masks, padded_size, image_size, height, width, meta, pano_temp=0.06):
        """deformable_detr_segm_vl.py:628-666, 875-918 at the storage points of semantic_single: class weights and the per-pixel
        probabilities are 16-bit operands of the [K', k] x [k, pixels] product, whose fp32 result is resized"""
        S = self.stages
        sem_cls = self.stuff_score(logits, meta)
        _, _, _, qidx = self.inference(sem_cls[0], coord[0], image_size)
        S["sem_query"], S["sem_box_cls"] = qidx, sem_cls
        up = F.interpolate(pred_masks[:, qidx], size=padded_size, mode="bilinear", align_corners=False)[0]
        mask_cls = self.R(F.softmax(sem_cls[0][qidx].sigmoid() / pano_temp, dim=-1))
        h, w = image_size
        result = torch.einsum("qc,qhw->chw", mask_cls, self.R(up[:, :h, :w].sigmoid()))
        self.hip["sem_seg"] = result
        r = F.interpolate(result[None], size=(height, width), mode="bilinear", align_corners=False)[0]     # sem_seg_postprocess on the crop
        if meta["entity"] == "stuff" and (meta.get("stuff_classes") or [""])[0] == "things" and meta.get("stuff_prob_thing", -1.0) > 0 \
                and meta.get("dataset_id", -1) >= 0:
            p = meta["stuff_prob_thing"]
            r[0, ...] = math.log(p / (1 - p))
        return r

    # ------------------------------------------------------------------------------------------------ whole forward
    @torch.no_grad()
    def forward(self, image, text_feats, **kw):
        if kw.get("prompt", "name") != "name" or kw.get("name_fusion_text") or kw.get("mask_prompt") is not None or kw.get("panoptic") is not None:
            raise NotImplementedError("RoundedApeOracle: name prompts, instance + semantic branches")
        self.hip = {}
        out = super().forward(image, text_feats, **kw)
        S, Hs = self.stages, self.hip
        Hs["pred_logits"], Hs["pred_boxes"] = S["pred_logits"][0], S["pred_boxes"][0]
        Hs["mask_embed"] = self._mask_embed_of(S)
        Hs["topk_proposals"] = S["topk_proposals"][0]
        return out

    def _mask_embed_of(self, S):
        return self.mlp(S["inter_states"][self.dec_layers - 1], "mask_embed")[0]

    def hip_stages(self):
        """{HIP stage name: tensor in the HIP pipeline's layout} of the last forward (ViT stages in RASTER token order)"""
        return dict(self.hip)


# ----------------------------------------------------------------------------------------------------------------------
# Harness pieces shared by tests/test_same_rounding.py and bench.py's `parity.vs_same_rounding_oracle`
# ----------------------------------------------------------------------------------------------------------------------
VIT_KEYS = ("vit_embed", "vit_blk")
BOX_KEYS = ("init_reference", "pred_boxes", "_ref")
BIASED_KEYS = ("pred_logits", "enc_class", "enc_cls2")       # logits = <x, w> + a large constant bias (prior-probability init, -4.6)


def teacher_stages(orc, tok2raster):
    """stage tensors of the rounded oracle's last forward as a StageTap teacher for the HIP pipeline: HIP names, HIP layouts, the ViT
    stages permuted from raster to the pipeline's window-major token order (ViT.token_order)"""
    t2r = tok2raster.long().cpu()
    out = {}
    for k, v in orc.hip_stages().items():
        if not torch.is_tensor(v) or k in ("topk_proposals", "sem_seg"):
            continue
        out[k] = v[t2r].contiguous() if k.startswith(VIT_KEYS) else v.contiguous()
    return out


def stage_distances(got, teacher):
    """{stage: (relative rms, relative max, kind)} of the HIP pipeline's recorded stage outputs against the rounded oracle's.
    Plain stages: ||g - t|| / ||t||.  Classifier logits: the denominator is the logit WITHOUT its mean (the constant bias carries no
    rounding).  Non-finite entries (anchors of padded / out-of-range tokens, deformable_transformer_vl.py:352-357) must coincide and are
    left out of the norms."""
    res = {}
    for k, t in teacher.items():
        g = got.get(k)
        if g is None or not torch.is_tensor(g) or tuple(g.shape) != tuple(t.shape):
            continue
        g, t = g.detach().float().cpu(), t.detach().float().cpu()
        fin = torch.isfinite(t)
        if not torch.equal(torch.isfinite(g), fin):
            res[k] = (float("inf"), float("inf"), "finite-pattern differs")
            continue
        g, t = g[fin].double(), t[fin].double()
        ref = t - t.mean() if k.endswith(BIASED_KEYS) else t
        den = ref.pow(2).sum().sqrt().clamp_min(1e-300)
        res[k] = (float((g - t).pow(2).sum().sqrt() / den), float((g - t).abs().max() / ref.abs().max().clamp_min(1e-300)),
                  "bias-free" if k.endswith(BIASED_KEYS) else "rel")
    return res
