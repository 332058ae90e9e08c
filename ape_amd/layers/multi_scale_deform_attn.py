"""MultiScaleDeformableAttention on the HIP sampler.

Mirror of ape/layers/multi_scale_deform_attn.py: module `MultiScaleDeformableAttention` (:127-358, same
constructor kwargs, parameter names and forward signature), function `multi_scale_deformable_attn_pytorch`
(:84-124) and the operator `torch.ops.ape.ms_deform_attn_forward` (ape/layers/csrc/vision.cpp:76-79), all backed by
csrc/msda.hip through the C-ABI.  `pytorch_attn` is accepted and ignored: there is one path, the HIP one.
"""
import os

import torch
import torch.nn as nn

from .. import ops
from ..packing import attach_cache, f32, pack_matrix


def multi_scale_deformable_attn_pytorch(value, value_spatial_shapes, sampling_locations, attention_weights):
    """same contract as the reference's pure-PyTorch sampler (:84-124); runs the HIP kernel"""
    shapes = [(int(h), int(w)) for h, w in (value_spatial_shapes.tolist() if torch.is_tensor(value_spatial_shapes) else value_spatial_shapes)]
    starts = [0]
    for h, w in shapes[:-1]:
        starts.append(starts[-1] + h * w)
    return ops.ms_deform_attn_forward(value.contiguous(), shapes, starts, sampling_locations.to(value.dtype).contiguous(),
                                      attention_weights.to(value.dtype).contiguous())


def _register_torch_op():
    """make torch.ops.ape.ms_deform_attn_forward resolve (vision.cpp:76-79) -> C-ABI"""
    try:
        lib = torch.library.Library("ape", "DEF")
        lib.define("ms_deform_attn_forward(Tensor value, Tensor spatial_shapes, Tensor level_start_index, "
                   "Tensor sampling_loc, Tensor attn_weight, int im2col_step) -> Tensor")

        def impl(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
            return ops.ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)

        lib.impl("ms_deform_attn_forward", impl, "CUDA")
        return lib
    except RuntimeError as exc:
        # the only tolerated failure: the operator is already defined (the reference's compiled extension was imported first)
        if "ms_deform_attn_forward" in str(exc) and hasattr(torch.ops.ape, "ms_deform_attn_forward"):
            return None
        raise


_TORCH_LIB = _register_torch_op()


HALF_MAX = 65504.0


def half_value_kwargs(dt, rows):
    """production mode (bf16), >= 2048 value rows: the value projection is written as IEEE HALF, saturated at +-65504.  The sampler
    is VALU bound and consumes a half value with one v_fma_mix_f32 per channel (a bf16 value must be unpacked first: -40 % VALU
    work, csrc/msda.hip), and half keeps 11 significant bits instead of 8.  The K = 256 GEMM kernel produces half only from 2048
    rows on, which is every encoder / decoder value projection of the full-size models; APE_MSDA_BF16_VALUE=1 keeps bf16."""
    if dt == torch.bfloat16 and rows >= 2048 and os.environ.get("APE_MSDA_BF16_VALUE") != "1":
        # no clamp argument: EVERY half store of the library saturates at +-65504 (csrc/common.h pack2h / stf<f16_t>), so asking the GEMM
        # for clamp = 65504 as rounds 3-4 did changed nothing in the output -- but it selected the kernel's generic epilogue (alpha,
        # run-time activation switch, clamp per element): 43.7 instead of 32.9 us per encoder layer, 181 instead of 115 us for the decoder's
        # six-layer value projection (profiles/r05_kres_probe.log)
        return dict(out_dtype=torch.float16)
    return {}          # float16 flavour: the value projection is half already (every half store of the library saturates)


class MultiScaleDeformableAttention(nn.Module):
    def __init__(self, embed_dim=256, num_heads=8, num_levels=4, num_points=4, img2col_step=64, dropout=0.1,
                 batch_first=False, pytorch_attn=False):
        super().__init__()
        if embed_dim != 256 or num_heads != 8 or num_points != 4:
            raise ValueError("ape_amd MSDeformAttn kernel is built for embed_dim=256, 8 heads, 4 points (APE configs)")
        self.dropout = nn.Dropout(dropout)
        self.batch_first = batch_first
        self.im2col_step = img2col_step
        self.embed_dim, self.num_heads, self.num_levels, self.num_points = embed_dim, num_heads, num_levels, num_points
        self.sampling_offsets = nn.Linear(embed_dim, num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dim, num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dim, embed_dim)
        self.output_proj = nn.Linear(embed_dim, embed_dim)
        self.pytorch_attn = pytorch_attn
        self.compute_dtype = torch.bfloat16
        attach_cache(self)

    def packed(self, dt):
        def build(dt):
            # offsets | logits in ONE projection, its rows zero-padded to a multiple of 128: the K = 256 GEMM kernel walks the output in
            # 128-column chunks and a partial last chunk (5 levels: 480 = 3 x 128 + 96 columns) takes its predicated, drained epilogue for
            # a quarter of the work -- 63 us per encoder layer for 128 MB, against 29 us for the value projection's 89 MB.  The consumers
            # see the first 480 columns of the [Q, 512] result through a view (the sampler takes a row stride).
            w = torch.cat([self.sampling_offsets.weight, self.attention_weights.weight], 0).detach()
            b = torch.cat([self.sampling_offsets.bias.detach().float(), self.attention_weights.bias.detach().float()])
            n, npad = w.shape[0], -(-w.shape[0] // 128) * 128
            if npad != n:
                w = torch.cat([w, w.new_zeros((npad - n, w.shape[1]))], 0)
                b = torch.cat([b, b.new_zeros((npad - n,))])
            return dict(
                woffw=pack_matrix(w, dt), boffw=b.contiguous(), noffw=n,
                wval=pack_matrix(self.value_proj.weight, dt), bval=f32(self.value_proj.bias),
                wout=pack_matrix(self.output_proj.weight, dt), bout=f32(self.output_proj.bias))
        return self._pack.get(self, dt, build)

    def forward_tokens(self, query_pos_sum, identity, ref, shapes, starts, dt, *, value_src=None, value=None, mask=None,
                       out_dtype=None, norm=None):
        """query_pos_sum [Q,256] (= query + pos), identity [Q,256], ref [Q,L,2|4] fp32.
        Either value_src [S,256] (projected here, padded rows zeroed with `mask`) or a pre-projected `value`.
        norm = (weight, bias, eps): the LayerNorm that follows this attention in the transformer layer, applied to the result -- in
        the output projection's epilogue where that kernel exists (87 k encoder tokens), as its own launch otherwise."""
        P = self.packed(dt)
        if value is None:
            value = ops.gemm(value_src, P["wval"], P["bval"], rowmask=mask, mask_mode=ops.MASK_ZERO_OUTPUT,
                             **half_value_kwargs(dt, value_src.shape[0]))
        # offsets | logits: fp32 in validation mode and for the decoder's 900 queries; IEEE half for the encoder's 87 k tokens in
        # production mode -- that GEMM is bound by the bytes it writes (168 MB per layer in fp32) and the sampler reads them
        # back; half keeps 11 significant bits (offsets are a few pixels, logits feed a 20-way softmax)
        half = dt in ops.HALF16 and query_pos_sum.shape[0] >= 2048 and os.environ.get("APE_MSDA_F32_OFFSETS") != "1"
        offw = ops.gemm(query_pos_sum, P["woffw"], P["boffw"], out_dtype=torch.float16 if half else torch.float32)[:, :P["noffw"]]
        samp = ops.msda_fused(value, shapes, starts, offw, ref, out_dtype=dt)
        if norm is not None and ops.gemm_norm_fusable(samp, P["wout"], identity, out_dtype):
            return ops.gemm(samp, P["wout"], P["bout"], residual=identity, norm=norm)
        y = ops.gemm(samp, P["wout"], P["bout"], residual=identity, out_dtype=out_dtype or dt)
        return y if norm is None else ops.layernorm(y, norm[0], norm[1], norm[2], out_dtype=out_dtype or dt)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_padding_mask=None,
                reference_points=None, spatial_shapes=None, level_start_index=None, **kwargs):
        """reference signature (:215-358); runs batch elements one after the other through the HIP path"""
        if value is None:
            value = query
        if identity is None:
            identity = query
        if query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            query, value, identity_b = query.permute(1, 0, 2), value.permute(1, 0, 2), identity.permute(1, 0, 2)
        else:
            identity_b = identity
        bs, nq, _ = query.shape
        shapes = [(int(h), int(w)) for h, w in spatial_shapes.tolist()]
        assert sum(h * w for h, w in shapes) == value.shape[1]
        starts = [0]
        for h, w in shapes[:-1]:
            starts.append(starts[-1] + h * w)
        dt = self.compute_dtype
        outs = []
        for b in range(bs):
            m = key_padding_mask[b].to(torch.uint8).contiguous() if key_padding_mask is not None else None
            o = self.forward_tokens(query[b].to(dt).contiguous(), identity_b[b].to(dt).contiguous(),
                                    reference_points[b].float().contiguous(), shapes, starts, dt,
                                    value_src=value[b].to(dt).contiguous(), mask=m)
            outs.append(o.to(identity.dtype))
        out = torch.stack(outs)
        if not self.batch_first:
            out = out.permute(1, 0, 2)
        return out
