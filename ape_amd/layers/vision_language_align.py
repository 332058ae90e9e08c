"""VisionLanguageAlign (open-vocabulary classifier) -- mirror of ape/layers/vision_language_align.py:8-52."""
import math

import torch
import torch.nn as nn

from .. import ops
from ..packing import attach_cache, f32, pack_matrix


class VisionLanguageAlign(nn.Module):
    def __init__(self, embed_dim, embed_dim_language, prior_prob=0.01, log_scale=0.0, clamp_dot_product=True):
        super().__init__()
        bias_value = -math.log((1 - prior_prob) / prior_prob)
        self.dot_product_projection_image = nn.Identity()
        self.dot_product_projection_text = nn.Linear(embed_dim_language, embed_dim, bias=True)
        self.log_scale = nn.Parameter(torch.Tensor([log_scale]), requires_grad=True)
        self.bias_lang = nn.Parameter(torch.zeros(embed_dim_language), requires_grad=True)
        self.bias0 = nn.Parameter(torch.Tensor([bias_value]), requires_grad=True)
        self.clamp_dot_product = clamp_dot_product
        self.compute_dtype = torch.bfloat16
        attach_cache(self)

    def text_side(self, embedding, dt):
        """per-vocabulary constants: projected tokens [K,256] (compute dtype) and per-class bias [K] fp32
        (vision_language_align.py:35-40).  embedding [K, D_l] fp32 on the device."""
        def build(dt):
            return dict(w=pack_matrix(self.dot_product_projection_text.weight, torch.float32),
                        b=f32(self.dot_product_projection_text.bias))
        P = self._pack.get(self, torch.float32, build)
        e = torch.nn.functional.normalize(embedding.float(), p=2, dim=-1)
        tok = ops.gemm((e / 2.0).contiguous(), P["w"], P["b"], out_dtype=torch.float32)
        bias = (ops.gemv(f32(self.bias_lang).reshape(1, -1), e.contiguous())[0] + self.bias0.detach().float()).contiguous()
        # host scalar, read once per module (not per image / per vocabulary: phrase mode rebuilds the vocabulary per image)
        inv_scale = self._pack.get(self, "inv_scale", lambda _k: 1.0 / float(self.log_scale.detach().exp()))
        return tok.to(dt).contiguous(), bias, inv_scale

    def forward_tokens(self, x, tok, bias, inv_scale):
        """x [Q,256] compute dtype -> logits [Q,K] fp32 (:44-51)"""
        return ops.gemm(x, tok, bias, alpha=inv_scale, clamp=50000.0 if self.clamp_dot_product else 0.0, out_dtype=torch.float32)

    def forward(self, x, embedding):
        """reference signature: x [bs,Q,256], embedding [bs,K,D_l] -> [bs,Q,K]"""
        dt = self.compute_dtype
        outs = []
        for b in range(x.shape[0]):
            tok, bias, inv_scale = self.text_side(embedding[b], dt)
            outs.append(self.forward_tokens(x[b].to(dt).contiguous(), tok, bias, inv_scale).to(x.dtype))
        return torch.stack(outs)


class StillClassifier(nn.Module):
    def __init__(self, hidden_dim):
        super().__init__()
        self.body = nn.Linear(hidden_dim, 1)
