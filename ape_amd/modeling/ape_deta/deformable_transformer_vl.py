"""DETA-style two-stage deformable transformer with vision-language fusion, on the HIP kernels.

Mirror of ape/modeling/ape_deta/deformable_transformer_vl.py: DeformableDetrTransformerEncoderVL (:20-121),
DeformableDetrTransformerDecoderVL (:124-255), DeformableDetrTransformerVL (:258-699) with the same constructor
kwargs and (detrex-compatible) parameter names.  The token stream is a single [T, 256] token-major tensor (the
reference's flatten(2).transpose(1,2) layout is our native one), per-image-size constants come from geometry.py,
and all arithmetic goes through ape_amd.ops.  Batch = 1 per forward like the reference's evaluation
(ape/data/build.py:79; fuse_helper.py:89-90 couples images inside a batch through a global max).
"""
import copy

import torch
import torch.nn as nn

from ... import ops
from ...layers import MultiScaleDeformableAttention
from ...packing import attach_cache, f32, pack_matrix, round_up
from ...stagetap import forcing, tap
from . import geometry as G
from ._containers import FFN, SelfAttention, TransformerLayer


class DeformableDetrTransformerEncoderVL(nn.Module):
    def __init__(self, embed_dim=256, num_heads=8, feedforward_dim=1024, attn_dropout=0.1, ffn_dropout=0.1, num_layers=6,
                 post_norm=False, num_feature_levels=4, vl_layer=None, use_act_checkpoint=False, pytorch_attn=False):
        super().__init__()
        self.layers = nn.ModuleList([
            TransformerLayer([MultiScaleDeformableAttention(embed_dim=embed_dim, num_heads=num_heads, dropout=attn_dropout,
                                                            batch_first=True, num_levels=num_feature_levels,
                                                            pytorch_attn=pytorch_attn)],
                             FFN(embed_dim=embed_dim, feedforward_dim=feedforward_dim, output_dim=embed_dim, num_fcs=2,
                                 ffn_drop=ffn_dropout), 2, embed_dim) for _ in range(num_layers)])
        self.num_layers = num_layers
        self.embed_dim = embed_dim
        self.pre_norm = False
        self.post_norm_layer = nn.LayerNorm(embed_dim) if post_norm else None
        # vl_layer=None: the plain encoder of APE-L_A/B/C (deformable_transformer.py:20-106) -- no fusion, no `vl_layers` in the
        # state dict
        self.vl_layers = nn.ModuleList([copy.deepcopy(vl_layer) for _ in range(num_layers)]) if vl_layer is not None else None
        self.use_checkpoint = use_act_checkpoint

    def forward_tokens(self, x, geo, lvl_pos, l, dt, stages=None):
        """x [T,256], l [1, l_dim] fp32 -> (memory [T,256], l) -- reference loop :84-115"""
        xp = None
        for i, layer in enumerate(self.layers):
            if self.vl_layers is None:
                # plain encoder: the layer's input is its value, x + pos its query (the previous layer's last LayerNorm wrote both)
                v_new, qp, ljob = x, (xp if xp is not None else (x.float() + lvl_pos.float()).to(dt)), ops._Joined(l)
            else:
                # the language update feeds only the next layer's fusion: it runs as a parallel branch next to this layer's
                # deformable attention and FFN and is joined at the end of the layer
                v_new, qp, ljob = self.vl_layers[i].b_attn.forward_tokens(x, lvl_pos, l, dt, defer_language=True)
            # BaseTransformerLayer ("self_attn", "norm", "ffn", "norm"): value = fused tokens (no pos), identity = same
            # ... with the layer's first norm in the output projection's epilogue (csrc/gemm.hip gemm_kres_ln_kernel)
            x2 = layer.attentions[0].forward_tokens(qp, v_new, geo.enc_ref, geo.shapes, geo.starts, dt, value_src=v_new,
                                                    mask=geo.mask_u8, norm=layer.norm_params(0))
            # FFN + the layer's last norm: one launch on the one-kernel FFN path (csrc/ffn_fused.hip, LayerNorm in its epilogue)
            x, xp = layer.ffns[0].forward_tokens(x2, dt, norm=layer.norm_params(1)), None
            l = ljob.join()
            if stages is not None:
                if self.vl_layers is not None:
                    stages[f"enc{i}_fused_v"] = v_new
                    l = tap(stages, f"enc{i}_fused_l", l)
                x = tap(stages, f"enc{i}_out", x)
        if self.post_norm_layer is not None:
            x = ops.layernorm(x, f32(self.post_norm_layer.weight), f32(self.post_norm_layer.bias), self.post_norm_layer.eps, out_dtype=dt)
        return x, l


class DeformableDetrTransformerDecoderVL(nn.Module):
    def __init__(self, embed_dim=256, num_heads=8, feedforward_dim=1024, attn_dropout=0.1, ffn_dropout=0.1, num_layers=6,
                 return_intermediate=True, num_feature_levels=4, use_act_checkpoint=False, look_forward_twice=False,
                 pytorch_attn=False):
        super().__init__()
        self.layers = nn.ModuleList([
            TransformerLayer([SelfAttention(embed_dim=embed_dim, num_heads=num_heads, attn_drop=attn_dropout, batch_first=True),
                              MultiScaleDeformableAttention(embed_dim=embed_dim, num_heads=num_heads, dropout=attn_dropout,
                                                            batch_first=True, num_levels=num_feature_levels,
                                                            pytorch_attn=pytorch_attn)],
                             FFN(embed_dim=embed_dim, feedforward_dim=feedforward_dim, output_dim=embed_dim, ffn_drop=ffn_dropout),
                             3, embed_dim) for _ in range(num_layers)])
        self.num_layers = num_layers
        self.embed_dim = embed_dim
        self.return_intermediate = return_intermediate
        self.bbox_embed = None
        self.class_embed = None
        self.use_checkpoint = use_act_checkpoint
        self.look_forward_twice = look_forward_twice
        attach_cache(self)

    def packed(self, dt):
        def build(dt):
            ws = [l.attentions[1].value_proj.weight for l in self.layers]
            bs = [l.attentions[1].value_proj.bias.detach().float() for l in self.layers]
            return dict(wval=pack_matrix(torch.cat(ws, 0), dt), bval=torch.cat(bs).contiguous())
        return self._pack.get(self, dt, build)

    def forward_tokens(self, query, query_pos, memory, geo, reference, dt, stages=None, query_sum=None):
        """query/query_pos [Q,256], memory [T,256], reference [Q,4] fp32 (sigmoid space) -> (inter [list of Q,256],
        inter_ref [list of Q,4]) -- reference loop :195-250.  stages: per-layer taps "dec<i>_out", "dec<i>_delta" (box head
        output), "dec<i>_ref" (refined reference); under teacher forcing (stagetap.StageTap) every layer starts from the
        teacher's query stream and reference boxes.  query_sum = query + query_pos when the caller already has it."""
        P = self.packed(dt)
        E = self.embed_dim
        # value_proj of all layers in ONE pass over the encoder memory (the reference re-reads it per layer)
        from ...layers.multi_scale_deform_attn import half_value_kwargs
        value_all = ops.gemm(memory, P["wval"], P["bval"], rowmask=geo.mask_u8, mask_mode=ops.MASK_ZERO_OUTPUT,
                             **half_value_kwargs(dt, memory.shape[0]))
        Q = query.shape[0]
        vt_buf = ops.zeros((E, round_up(Q, 64)), dt, query.device)
        vr4 = geo.vr4                                                   # [L, 4] = cat(valid_ratios, valid_ratios)
        out = query
        outp = query_sum if query_sum is not None else (query.float() + query_pos.float()).to(dt)
        inter, inter_ref = [], []
        _, ref_in = ops.box_refine(None, reference.contiguous(), vr4)      # reference * valid ratios per level (:203-210)
        refjob = ops._Joined((reference, ref_in))
        for i, layer in enumerate(self.layers):
            x1 = layer.attentions[0].forward_tokens(out, outp, dt, vt_buf)
            x2, x2p = ops.layernorm(x1, *layer.norm_params(0), out_dtype=dt, add=query_pos)
            reference, ref_in = refjob.join()                  # the previous layer's box head ran next to this self-attention
            if i > 0:
                if stages is not None:
                    forced = tap(stages, f"dec{i - 1}_ref", reference)
                    if forced is not reference:                # teacher forcing: this layer samples around the teacher's boxes
                        reference = forced
                        _, ref_in = ops.box_refine(None, reference.contiguous(), vr4)
                inter_ref.append(reference)
            x3 = layer.attentions[1].forward_tokens(x2p, x2, ref_in, geo.shapes, geo.starts, dt,
                                                    value=value_all[:, i * E:(i + 1) * E])
            x4 = ops.layernorm(x3, *layer.norm_params(1), out_dtype=dt)
            x5 = layer.ffns[0].forward_tokens(x4, dt)
            out, outp = ops.layernorm(x5, *layer.norm_params(2), out_dtype=dt, add=query_pos)
            if stages is not None:
                forced = tap(stages, f"dec{i}_out", out)
                if forced is not out:
                    out, outp = forced, (forced.float() + query_pos.float()).to(dt)
            if self.bbox_embed is not None:
                # box head (3-layer MLP) + refinement (:232-246) feed only the NEXT layer's cross-attention: parallel branch
                def refine(i=i, out=out, reference=reference):
                    tmp = self.bbox_embed[i].forward_tokens(out, dt, out_dtype=torch.float32)
                    if stages is not None:
                        stages[f"dec{i}_delta"] = tmp
                    return ops.box_refine(tmp, reference, vr4)
                refjob = ops.fork(refine)
            inter.append(out)
        reference, _ = refjob.join()
        reference = tap(stages, f"dec{len(self.layers) - 1}_ref", reference)
        inter_ref.append(reference)
        return inter, inter_ref


class DeformableDetrTransformerVL(nn.Module):
    def __init__(self, encoder=None, decoder=None, num_feature_levels=4, as_two_stage=False, two_stage_num_proposals=300,
                 assign_first_stage=False, pre_nms_topk=1000, nms_thresh_enc=0.9, proposal_ambiguous=0):
        super().__init__()
        assert as_two_stage and assign_first_stage, "ape_amd implements the two-stage DETA selection used by every APE config"
        self.encoder, self.decoder = encoder, decoder
        self.num_feature_levels = num_feature_levels
        self.as_two_stage = as_two_stage
        self.two_stage_num_proposals = two_stage_num_proposals
        self.assign_first_stage = assign_first_stage
        self.pre_nms_topk, self.nms_thresh_enc = pre_nms_topk, nms_thresh_enc
        self.proposal_ambiguous = proposal_ambiguous
        self.compute_dtype = torch.bfloat16          # set by DeformableDETRSegmVL.set_compute_dtype; read by the reference-signature forward()
        self.embed_dim = self.encoder.embed_dim
        self.level_embeds = nn.Parameter(torch.Tensor(self.num_feature_levels, self.embed_dim))
        self.enc_output = nn.Linear(self.embed_dim, self.embed_dim)
        self.enc_output_norm = nn.LayerNorm(self.embed_dim)
        self.pos_trans = nn.Linear(self.embed_dim * 2, self.embed_dim * 2)
        self.pos_trans_norm = nn.LayerNorm(self.embed_dim * 2)
        self.pix_trans = nn.Linear(self.embed_dim, self.embed_dim)
        self.pix_trans_norm = nn.LayerNorm(self.embed_dim)
        nn.init.normal_(self.level_embeds)
        attach_cache(self)

    # ------------------------------------------------------------------ packing
    def packed(self, dt):
        def build(dt):
            dec = self.decoder
            nd = dec.num_layers
            # proposal_ambiguous = 0 (APE-L_A/B/C): the "ambiguous" copy is the main head itself -- the per-token maximum of two
            # identical logits keeps the first (= the main) pair, so the same kernels produce the single-head result
            amb = bool(self.proposal_ambiguous)
            be, bea = dec.bbox_embed[nd], (dec.bbox_embed_ambiguous[0] if amb else dec.bbox_embed[nd])
            ce, cea = dec.class_embed[nd], (dec.class_embed_ambiguous[0] if amb else dec.class_embed[nd])
            return dict(
                wenc=pack_matrix(self.enc_output.weight, dt), benc=f32(self.enc_output.bias),
                nenc=(f32(self.enc_output_norm.weight), f32(self.enc_output_norm.bias), self.enc_output_norm.eps),
                wpos=pack_matrix(self.pos_trans.weight, dt), bpos=f32(self.pos_trans.bias),
                npos=(f32(self.pos_trans_norm.weight), f32(self.pos_trans_norm.bias), self.pos_trans_norm.eps),
                wpix=pack_matrix(self.pix_trans.weight, dt), bpix=f32(self.pix_trans.bias),
                npix=(f32(self.pix_trans_norm.weight), f32(self.pix_trans_norm.bias), self.pix_trans_norm.eps),
                # two-stage heads: main + ambiguous copies share their input, so their first layers are one GEMM
                w1=pack_matrix(torch.cat([be.layers[0].weight, bea.layers[0].weight], 0), dt),
                b1=torch.cat([be.layers[0].bias, bea.layers[0].bias]).detach().float().contiguous(),
                w2=(pack_matrix(be.layers[1].weight, dt), pack_matrix(bea.layers[1].weight, dt)),
                b2=(f32(be.layers[1].bias), f32(bea.layers[1].bias)),
                w3=(pack_matrix(be.layers[2].weight, dt), pack_matrix(bea.layers[2].weight, dt)),
                b3=(f32(be.layers[2].bias), f32(bea.layers[2].bias)),
                wcls=pack_matrix(torch.cat([ce.weight, cea.weight], 0), dt),
                bcls=torch.cat([ce.bias, cea.bias]).detach().float().contiguous(),
                level_embeds=f32(self.level_embeds))
        return self._pack.get(self, dt, build)

    def lvl_pos(self, geo, dt):
        key = (dt, self.level_embeds.data_ptr(), self.level_embeds._version)
        if key not in geo._lvl_pos:
            geo._lvl_pos.clear()
            geo._lvl_pos[key] = (geo.pos + self.level_embeds.detach().float()[geo.level_ids]).to(dt).contiguous()
        return geo._lvl_pos[key]

    # ------------------------------------------------------------------ forward (:422-699), batch 1
    def forward_tokens(self, src, geo, l, dt, forced_topk=None, stages=None, after_encoder=None, mask_prompt=None):
        """src [T,256] neck output (token-major, levels concatenated), l [1, l_dim] fp32 fusion token(s).
        after_encoder(memory): hook called as soon as the encoder memory exists (the caller forks the mask-feature branch
        there, so that it runs next to the latency-bound selection + decoder)."""
        P = self.packed(dt)
        lvl_pos = self.lvl_pos(geo, dt)
        memory, l_out = self.encoder.forward_tokens(src, geo, lvl_pos, l, dt, stages)
        memory = tap(stages, "memory", memory)
        if after_encoder is not None:
            after_encoder(memory)
        # gen_encoder_output_proposals (:321-369): rows of padded / out-of-range anchors enter enc_output as zeros
        invalid, anchors = geo.invalid_u8, geo.proposals
        if mask_prompt is not None:
            # mask prompt (:356-358, 364-365): tokens outside the prompted region are no proposals -- anchors +inf, memory rows zero.
            # A per-request preprocessing step (not in the captured steady-state path): tensor-level
            outside = ~mask_prompt.reshape(-1).to(device=memory.device, dtype=torch.bool)
            invalid = (invalid.reshape(-1).bool() | outside).to(torch.uint8).reshape(invalid.shape).contiguous()
            anchors = anchors.masked_fill(outside[:, None], float("inf")).contiguous()
        om = ops.gemm(memory, P["wenc"], P["benc"], rowmask=invalid, mask_mode=ops.MASK_ZERO_INPUT)
        om = tap(stages, "output_memory", ops.layernorm(om, *P["nenc"], out_dtype=dt))
        E = self.embed_dim
        T = om.shape[0]
        h1 = ops.gemm(om, P["w1"], P["b1"], act=ops.ACT_RELU)                                    # [T, 2E]
        h2 = torch.empty((T, 2 * E), dtype=dt, device=om.device)
        ops.gemm(h1[:, :E], P["w2"][0], P["b2"][0], act=ops.ACT_RELU, out=h2[:, :E])
        ops.gemm(h1[:, E:], P["w2"][1], P["b2"][1], act=ops.ACT_RELU, out=h2[:, E:])
        d = torch.empty((T, 8), dtype=torch.float32, device=om.device)
        ops.gemm(h2[:, :E], P["w3"][0], P["b3"][0], out=d[:, :4])
        ops.gemm(h2[:, E:], P["w3"][1], P["b3"][1], out=d[:, 4:])
        cls2 = ops.gemm(om, P["wcls"], P["bcls"], out_dtype=torch.float32)                       # [T, 2]
        # ambiguous heads (:508-533): per token keep the (logit, box) pair with the larger logit (first on ties), add the
        # anchors, and produce the clamped corner boxes the proposal NMS works on -- one kernel (csrc/topk.hip)
        if stages is not None:
            cls2, d = tap(stages, "enc_cls2", cls2), tap(stages, "enc_delta8", d)
        enc_class, enc_coord, xyxy = ops.enc_finalize(cls2, d, anchors)
        if stages is not None:
            stages["query_l"] = l_out
            enc_class, enc_coord = tap(stages, "enc_class", enc_class), tap(stages, "enc_coord_unact", enc_coord)
        if forced_topk is not None:
            topk = forced_topk.to(om.device).long()
        else:
            # two-stage selection (:565-627): per-level top-k, NMS 0.9, per-level quota, fallback -- fixed-shape device code
            topk = ops.select_proposals(enc_class, xyxy, geo.shapes, self.pre_nms_topk, self.two_stage_num_proposals,
                                        self.nms_thresh_enc)
        # query initialisation (:629-645): sigmoid + sine embedding of the selected proposals in one launch, both LayerNorms +
        # split + add + the first layer's query + query_pos in another (csrc/boxes.hip)
        reference, pe, topk32 = ops.query_init(enc_coord, topk, G.dim_t_table(128, 10000, om.device), dt)
        pos_raw = ops.gemm(pe, P["wpos"], P["bpos"], out_dtype=torch.float32)
        pix_raw = ops.gemm(ops.gather_rows(om, topk32), P["wpix"], P["bpix"], out_dtype=torch.float32)
        query_pos, query, query_sum = ops.query_finish(pos_raw, pix_raw, P["npos"], P["npix"], dt)
        if stages is not None:
            stages["topk_proposals"] = topk
            q0, p0 = query, query_pos
            query, query_pos = tap(stages, "query_init", query), tap(stages, "query_pos", query_pos)
            if query is not q0 or query_pos is not p0:
                query_sum = None                                          # teacher forcing: the decoder recomputes it
            reference = tap(stages, "init_reference", reference)
        inter, inter_ref = self.decoder.forward_tokens(query, query_pos, memory, geo, reference, dt, stages, query_sum=query_sum)
        return dict(inter_states=inter, init_reference=reference, inter_references=inter_ref, enc_class=enc_class,
                    enc_coord_unact=enc_coord, memory=memory, query_l=l_out, topk_proposals=topk)

    def forward(self, multi_level_feats, multi_level_masks, multi_level_pos_embeds, query_embed=None, query_l=None, attention_mask_l=None,
                multi_level_masks_prompt=None, **kwargs):
        """reference signature (deformable_transformer_vl.py:422-689), two-stage form: NCHW feature levels, their padding masks
        [B, H_l, W_l] and position embeddings [B, C, H_l, W_l], the language tokens query_l [B, L, l_dim] ->
        (inter_states [layers, B, Q, C], init_reference [B, Q, 4], inter_references [layers, B, Q, 4], enc_outputs_class
        [B, T, 1], enc_outputs_coord_unact [B, T, 4], anchors [B, T, 4] (sigmoid space), memory [B, T, C], query_l [B, L, l_dim]).
        Batch elements run one after the other through `forward_tokens` (the reference evaluates batch 1)."""
        if not self.as_two_stage or query_embed is not None:
            raise NotImplementedError("ape_amd: the two-stage transformer of the APE configs (as_two_stage=True, no query_embed)")
        if attention_mask_l is not None:
            raise NotImplementedError("ape_amd: language masks (un-reduced text tokens, text_feature_reduce_before_fusion=False) are not implemented")
        dt = getattr(self, "compute_dtype", torch.bfloat16)
        B = multi_level_feats[0].shape[0]
        rows = []
        for b in range(B):
            masks = [m[b].bool() for m in multi_level_masks]
            pos = [p[b].flatten(1).t() for p in multi_level_pos_embeds]
            geo = G.geometry_from_masks(masks, pos)
            src = torch.cat([f[b].flatten(1).t() for f in multi_level_feats]).to(dt).contiguous()
            # multi_level_masks_prompt (:430, :465-471): per-level bool maps [B, H_l, W_l], flattened and concatenated like the
            # features; tokens outside the prompt are no proposals (:356-365)
            mp = None if multi_level_masks_prompt is None else torch.cat([m[b].reshape(-1) for m in multi_level_masks_prompt]).to(torch.bool)
            tr = self.forward_tokens(src, geo, query_l[b].float().contiguous(), dt, mask_prompt=mp)
            anchors = geo.proposals if mp is None else geo.proposals.masked_fill(~mp.to(geo.proposals.device)[:, None], float("inf"))   # (:356-358)
            rows.append((torch.stack(tr["inter_states"]), tr["init_reference"], torch.stack(tr["inter_references"]),
                         tr["enc_class"][:, None], tr["enc_coord_unact"], anchors.sigmoid(), tr["memory"], tr["query_l"]))
        odt = multi_level_feats[0].dtype
        cat = lambda i, dim: torch.stack([r[i] for r in rows], dim)                      # noqa: E731
        return (cat(0, 1).to(odt), cat(1, 0).to(odt), cat(2, 1).to(odt), cat(3, 0).to(odt), cat(4, 0).to(odt), cat(5, 0).to(odt),
                cat(6, 0).to(odt), cat(7, 0).to(query_l.dtype))

