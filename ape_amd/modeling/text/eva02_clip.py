"""CLIP text tower on the HIP kernels -- mirror of `TextTransformer` / `Transformer` /
`ResidualAttentionBlock` in ape/modeling/text/eva02_clip/transformer.py:443-517,642-737 (same attribute names, so the
`net.text.*` keys of an EVA-CLIP checkpoint load unchanged) and of the text half of `CustomCLIP` (model.py:271-308).

Pre-norm blocks: x += out_proj(MHA(ln_1(x), causal)); x += c_proj(gelu(c_fc(ln_2(x)))); then ln_final and the text
projection.  Device mapping: token-major [texts * Lp, width] activations (Lp = context rounded up to 8 rows per text, the
attention kernel's batch stride), fp32 residual stream, bf16 GEMM operands (fp32 in validation mode); q|k from one GEMM,
V produced transposed (`trans_out`) for the flash-attention kernel, which takes the causal mask as a template argument;
GELU / bias / residual live in the GEMM epilogues.

Causality makes position t independent of every later position, and the tower's consumers read the feature at the
end-of-text token only (`last_hidden_state_eot`, clip_wrapper_eva02.py:117-128): with `context=None` the tower therefore
runs on the first max(eot) + 1 positions of the batch (rounded up to 8) instead of all 77 -- for class names (3-8 tokens)
that is an order of magnitude less work with bit-identical end-of-text features.  `all_positions=True` computes the
full context (the un-reduced `last_hidden_state`)."""
import torch
import torch.nn as nn

from ... import ops
from ...packing import attach_cache, f32, pack_matrix, round_up

# text_cfg / embed_dim of the reference's model_configs/*.json (head width 64 everywhere; nn.GELU; LayerNorm eps 1e-5)
TEXT_CONFIGS = {
    "EVA02-CLIP-bigE-14-plus": dict(width=1280, heads=20, layers=32, embed_dim=1024),      # APE-A/B/C/D (configs .../ape_deta_vitl_eva02_*:35-42)
    "EVA02-CLIP-bigE-14": dict(width=1024, heads=16, layers=24, embed_dim=1024),
    "EVA02-CLIP-L-14": dict(width=768, heads=12, layers=12, embed_dim=768),
    "EVA02-CLIP-L-14-336": dict(width=768, heads=12, layers=12, embed_dim=768),
    "EVA02-CLIP-B-16": dict(width=512, heads=8, layers=12, embed_dim=512),
    "EVA01-CLIP-g-14": dict(width=768, heads=12, layers=12, embed_dim=1024),
    "EVA01-CLIP-g-14-plus": dict(width=1024, heads=16, layers=24, embed_dim=1024),
}


class _MHAParams(nn.Module):
    """parameter holder with nn.MultiheadAttention's names (in_proj_weight / in_proj_bias / out_proj.*)"""

    def __init__(self, width):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * width, width))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * width))
        self.out_proj = nn.Linear(width, width)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model, n_head, mlp_ratio=4.0):
        super().__init__()
        self.n_head = n_head
        self.ln_1 = nn.LayerNorm(d_model)
        self.attn = _MHAParams(d_model)
        self.ln_2 = nn.LayerNorm(d_model)
        self.mlp = nn.Sequential()
        self.mlp.add_module("c_fc", nn.Linear(d_model, int(d_model * mlp_ratio)))
        self.mlp.add_module("gelu", nn.GELU())
        self.mlp.add_module("c_proj", nn.Linear(int(d_model * mlp_ratio), d_model))
        attach_cache(self)

    def packed(self, dt):
        def build(dt):
            a, W = self.attn, self.ln_1.weight.shape[0]
            return dict(
                wqk=pack_matrix(a.in_proj_weight[: 2 * W], dt), bqk=f32(a.in_proj_bias[: 2 * W]),
                wv=pack_matrix(a.in_proj_weight[2 * W:], dt), bv=f32(a.in_proj_bias[2 * W:]),
                wo=pack_matrix(a.out_proj.weight, dt), bo=f32(a.out_proj.bias),
                wfc=pack_matrix(self.mlp.c_fc.weight, dt), bfc=f32(self.mlp.c_fc.bias),
                wpr=pack_matrix(self.mlp.c_proj.weight, dt), bpr=f32(self.mlp.c_proj.bias),
                n1=(f32(self.ln_1.weight), f32(self.ln_1.bias), self.ln_1.eps),
                n2=(f32(self.ln_2.weight), f32(self.ln_2.bias), self.ln_2.eps))
        return self._pack.get(self, dt, build)

    def forward_tokens(self, x, dt, batch, length, stride, vt_buf):
        """x [batch * stride, W] fp32 residual stream -> the same after this block (transformer.py:480-483)"""
        P = self.packed(dt)
        W = x.shape[1]
        hd = W // self.n_head
        xn = ops.layernorm(x, P["n1"][0], P["n1"][1], P["n1"][2], out_dtype=dt)
        vjob = ops.fork(lambda: ops.gemm(xn, P["wv"], P["bv"], trans_out=True, out=vt_buf))
        qk = ops.gemm(xn, P["wqk"], P["bqk"])
        vt = vjob.join()
        o = ops.attention(qk[:, :W], qk[:, W:], vt, batch=batch, n=length, heads=self.n_head, head_dim=hd, scale=hd ** -0.5,
                          stride=stride, causal=True)
        x = ops.gemm(o, P["wo"], P["bo"], residual=x, out_dtype=torch.float32)
        xn = ops.layernorm(x, P["n2"][0], P["n2"][1], P["n2"][2], out_dtype=dt)
        h = ops.gemm(xn, P["wfc"], P["bfc"], act=ops.ACT_GELU)
        return ops.gemm(h, P["wpr"], P["bpr"], residual=x, out_dtype=torch.float32)


class Transformer(nn.Module):
    def __init__(self, width, layers, heads, mlp_ratio=4.0):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.ModuleList([ResidualAttentionBlock(width, heads, mlp_ratio) for _ in range(layers)])

    def get_cast_dtype(self):
        return self.resblocks[0].mlp.c_fc.weight.dtype


class TextTransformer(nn.Module):
    def __init__(self, context_length=77, vocab_size=49408, width=512, heads=8, layers=12, output_dim=512):
        super().__init__()
        if width // heads != 64 or width % heads:
            raise ValueError("ape_amd TextTransformer: head width 64 (every CLIP text tower of the reference)")
        self.context_length, self.vocab_size, self.width, self.output_dim = context_length, vocab_size, width, output_dim
        self.token_embedding = nn.Embedding(vocab_size, width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, width))
        self.transformer = Transformer(width, layers, heads)
        self.ln_final = nn.LayerNorm(width)
        self.text_projection = nn.Parameter(torch.empty(width, output_dim))
        self.compute_dtype = torch.bfloat16
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        nn.init.normal_(self.text_projection, std=width ** -0.5)
        for b in self.transformer.resblocks:
            nn.init.normal_(b.attn.in_proj_weight, std=width ** -0.5)
        attach_cache(self)

    def packed(self, dt):
        def build(dt):
            return dict(emb=self.token_embedding.weight.detach().to(dt).contiguous(),
                        pos=self.positional_embedding.detach().to(dt).contiguous(),
                        nf=(f32(self.ln_final.weight), f32(self.ln_final.bias), self.ln_final.eps),
                        wproj=pack_matrix(self.text_projection.detach().t(), dt))
        return self._pack.get(self, dt, build)

    @torch.no_grad()
    def forward_tokens(self, text, all_positions=False, project_all=True):
        """text int [B, context] token ids -> (eot features [B, output_dim] fp32, all-position features [B, L, output_dim] fp32
        or None, L).  L = the positions computed: the whole context with all_positions, else max(eot) + 1 rounded up to 8.
        project_all=False: the all-position features are ln_final(x) [B, L, width] WITHOUT the text projection (what the
        reference's TextTransformer.forward(return_all_features=True) returns, transformer.py:722-737)."""
        dt = self.compute_dtype
        P = self.packed(dt)
        B, ctx = text.shape
        tok = text.to(torch.int32).contiguous()
        eot = text.argmax(dim=-1)                                    # the end-of-text id is the largest (:735-736)
        L = ctx if all_positions else min(ctx, round_up(int(eot.max()) + 1, 8))
        Lp = round_up(L, 8)
        x = ops.embed_tokens(tok, P["emb"], P["pos"], L, Lp)         # [B * Lp, W] fp32
        # V^T is read in 64-column tiles from each text's first column: (B - 1) * Lp + round_up(L, 64) <= round_up(B * Lp, 64) + 64
        vt_buf = ops.zeros((self.width, round_up(B * Lp, 64) + 64), dt, x.device)
        for blk in self.transformer.resblocks:
            x = blk.forward_tokens(x, dt, B, L, Lp, vt_buf)
        rows = (torch.arange(B, device=x.device) * Lp + eot).to(torch.int32)
        xe = ops.layernorm(ops.gather_rows(x, rows), P["nf"][0], P["nf"][1], P["nf"][2], out_dtype=dt)
        feat = ops.gemm(xe, P["wproj"], None, out_dtype=torch.float32)
        full = None
        if all_positions:
            if project_all:
                xa = ops.layernorm(x, P["nf"][0], P["nf"][1], P["nf"][2], out_dtype=dt)
                full = ops.gemm(xa, P["wproj"], None, out_dtype=torch.float32).view(B, Lp, -1)[:, :L]
            else:
                full = ops.layernorm(x, P["nf"][0], P["nf"][1], P["nf"][2], out_dtype=torch.float32).view(B, Lp, -1)[:, :L]
        return feat, full, L

    def forward(self, text, return_all_features=False):
        """reference signature (transformer.py:722-737)"""
        feat, full, _ = self.forward_tokens(text, all_positions=return_all_features, project_all=False)
        return full if return_all_features else feat


class CustomCLIPText(nn.Module):
    """the text half of CustomCLIP (model.py:271-308): `.text` + `logit_scale`; the visual tower is deleted by the wrapper
    (clip_wrapper_eva02.py:29)"""

    def __init__(self, embed_dim, text_cfg):
        super().__init__()
        self.text = TextTransformer(width=text_cfg["width"], heads=text_cfg["heads"], layers=text_cfg["layers"],
                                    context_length=text_cfg.get("context_length", 77), vocab_size=text_cfg.get("vocab_size", 49408),
                                    output_dim=embed_dim)
        self.logit_scale = nn.Parameter(torch.ones([]) * 2.6592600369327783)        # log(1 / 0.07)

    def encode_text(self, text, normalize=False):
        f = self.text(text)
        return torch.nn.functional.normalize(f, dim=-1) if normalize else f
