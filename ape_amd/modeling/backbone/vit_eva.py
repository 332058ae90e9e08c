"""EVA-01 (MIM-pretrained) ViT backbone with decomposed relative positions, on the HIP kernels.

Host-side mirror of ape/modeling/backbone/vit_eva.py (ViT :311-480, Block :209-308, Attention :70-146) and the helpers it takes
from utils_eva.py (add_decomposed_rel_pos :132-161, get_rel_pos :65-129, get_abs_pos, window_partition, PatchEmbed): same class
names, constructor kwargs and state-dict keys (blocks.<i>.{norm1, attn.{qkv, q_bias, v_bias, proj, rel_pos_h, rel_pos_w}, norm2,
mlp.{fc1, fc2}[, gamma_1, gamma_2]}), so the reference's LazyConfig (configs/common/backbone/vitg_eva01.py:9-47) instantiates it
unchanged.  The arithmetic runs through ape_amd.ops (C-ABI -> HIP); there is no PyTorch fallback.

MI355X-first formulation of the relative positions: the reference materialises the [B*heads, N, N] score tensor and adds two
einsum terms to it (vit_eva.py:131-141).  Here the terms become EXTRA CHANNELS of the attention operands,
    scale q.k + q.Rh[qh - kh] + q.Rw[qw - kw] = [scale q | q.Rh[qh - .] | q.Rw[qw - .]] . [k | one-hot(kh) | one-hot(kw)],
so the flash-attention kernel runs unchanged with a q.k width of hd + Hk + Wk (zero-padded to 128 for the 16 x 16 windows of a
head width 88, to 256 / 288 / 320 for the global blocks) over the V width 128 and never holds an N x N tensor in HBM.  The products of
a query with ALL table rows are one MFMA GEMM (rows = (token, head), weight = [Rh ; Rw]); csrc/relpos.hip gathers them into place.
Tokens stay token-major and WINDOW-MAJOR through all blocks, as in vit_eva_clip.py.
"""
import math
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...packing import attach_cache, f32, pack_matrix, round_up
from ...stagetap import tap
from .vit_eva_clip import Backbone, LastLevelMaxPool, PatchEmbed, SimpleFeaturePyramid, padded_head_dim, window_major_order  # noqa: F401

__all__ = ["ViT", "SimpleFeaturePyramid"]

# q.k widths of csrc/attention.hip over a V width of 128 (ape_hip_attention_ext); 128 is the ordinary kernel
EXT_WIDTHS = (128, 256, 288, 320)


def ext_width(hd, hk, wk):
    need = hd + hk + wk
    for w in EXT_WIDTHS:
        if need <= w:
            return w
    raise ValueError(f"ape_amd vit_eva: head width {hd} + {hk} + {wk} relative-position channels > {EXT_WIDTHS[-1]} "
                     "(token grids beyond 96 x 96 at a head width of 88 need a wider attention tile)")


BEIT_RATIO = 1.0903078          # growth of the node spacing in the BEiT-style table resize (utils_eva.py:91)


def resized_rel_pos(rel_pos, size, interp_type="vitdet"):
    """get_rel_pos (utils_eva.py:65-129) for q_size == k_size == size: the table with 2 size - 1 rows; row (q - k) + size - 1 belongs to
    the offset q - k.  A checkpoint table of another length is resized ONCE, at weight-packing time:
      "vitdet": linear interpolation over the row index (:81-91);
      "beit"  : (:92-118) the source rows sit at the geometric-progression offsets 0, +-1, +-(1 + r), +-(1 + r + r^2), ... (r = 1.0903078),
                a cubic spline through them (scipy interp1d, extrapolating) is read at the integer offsets -(size - 1) .. size - 1 --
                a host-side weight transform exactly as in the reference, which imports scipy for it too."""
    want = 2 * size - 1
    rel_pos = rel_pos.detach().float()
    if rel_pos.shape[0] == want:
        return rel_pos.contiguous()
    if interp_type == "vitdet":
        rel_pos = F.interpolate(rel_pos.reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1), size=want, mode="linear")
        return rel_pos.reshape(-1, want).permute(1, 0).contiguous()
    if interp_type != "beit":
        raise NotImplementedError(f"ape_amd vit_eva: interp_type {interp_type!r} (the reference knows 'vitdet' and 'beit')")
    import numpy as np
    from scipy import interpolate
    src = rel_pos.shape[0]
    half = src // 2
    steps = BEIT_RATIO ** np.arange(half, dtype=np.float64)                 # 1, r, r^2, ...: distance between neighbouring source rows
    right = 1.0 + np.concatenate([[0.0], np.cumsum(steps[1:])]) if half else np.zeros(0)      # 1, 1 + r, 1 + r + r^2, ...
    nodes = np.concatenate([-right[::-1], [0.0], right])                    # positions of the src rows (src is odd: 2 * half + 1)
    reach = want // 2.0
    at = np.arange(-reach, reach + 0.1, 1.0)                                # the integer offsets of the resized table
    spline = interpolate.interp1d(nodes, rel_pos.cpu().numpy(), kind="cubic", axis=0, fill_value="extrapolate")
    return torch.from_numpy(np.ascontiguousarray(spline(at))).to(dtype=torch.float32, device=rel_pos.device).contiguous()


class Mlp(nn.Module):
    """timm Mlp (fc1 -> act -> fc2), vit_eva.py:264"""

    def __init__(self, in_features, hidden_features, act_layer=nn.GELU):
        super().__init__()
        assert act_layer is nn.GELU, "ape_amd vit_eva: the GEMM epilogue implements GELU (every APE config)"
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, in_features)


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=True, use_rel_pos=False, rel_pos_zero_init=True, input_size=None,
                 beit_like_qkv_bias=False, interp_type="vitdet"):
        super().__init__()
        if interp_type not in ("vitdet", "beit"):
            raise NotImplementedError(f"ape_amd vit_eva: interp_type {interp_type!r} (the reference knows 'vitdet' and 'beit')")
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = head_dim ** -0.5
        self.beit_like_qkv_bias = beit_like_qkv_bias
        if beit_like_qkv_bias:
            self.q_bias = nn.Parameter(torch.zeros(dim))
            self.v_bias = nn.Parameter(torch.zeros(dim))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.use_rel_pos = use_rel_pos
        self.interp_type = interp_type
        if use_rel_pos:
            self.rel_pos_h = nn.Parameter(torch.zeros(2 * input_size[0] - 1, head_dim))
            self.rel_pos_w = nn.Parameter(torch.zeros(2 * input_size[1] - 1, head_dim))
            if not rel_pos_zero_init:
                nn.init.trunc_normal_(self.rel_pos_h, std=0.02)
                nn.init.trunc_normal_(self.rel_pos_w, std=0.02)

    def qkv_weights(self):
        """(Wq, Wk, Wv [E, E], bq, bk, bv [E]) of either bias parameterisation (vit_eva.py:122-128)"""
        E = self.qkv.weight.shape[1]
        w = self.qkv.weight.detach().float()
        zeros = torch.zeros(E, dtype=torch.float32, device=w.device)
        if self.beit_like_qkv_bias:
            bq, bk, bv = self.q_bias.detach().float(), zeros, self.v_bias.detach().float()
        elif self.qkv.bias is not None:
            b = self.qkv.bias.detach().float()
            bq, bk, bv = b[:E], b[E:2 * E], b[2 * E:]
        else:
            bq = bk = bv = zeros
        return w[:E], w[E:2 * E], w[2 * E:], bq, bk, bv


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=True, drop_path=0.0, norm_layer=nn.LayerNorm, act_layer=nn.GELU,
                 use_rel_pos=False, rel_pos_zero_init=True, window_size=0, use_residual_block=False, input_size=None,
                 beit_like_qkv_bias=False, beit_like_gamma=False, interp_type="vitdet"):
        super().__init__()
        assert not use_residual_block, "ape_amd vit_eva: no convolutional residual blocks (residual_block_indexes=[] in every APE config)"
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, use_rel_pos=use_rel_pos, rel_pos_zero_init=rel_pos_zero_init,
                              input_size=input_size if window_size == 0 else (window_size, window_size),
                              beit_like_qkv_bias=beit_like_qkv_bias, interp_type=interp_type)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer)
        self.window_size = window_size
        self.beit_like_gamma = beit_like_gamma
        if beit_like_gamma:
            self.gamma_1 = nn.Parameter(torch.ones(dim))
            self.gamma_2 = nn.Parameter(torch.ones(dim))
        attach_cache(self)

    def packed(self, dt, group):
        """group: side of this block's attention group in tokens (the window, or the whole grid for a global block)"""
        def build(dt):
            a, m = self.attn, self.mlp
            wq, wk, wv, bq, bk, bv = a.qkv_weights()
            E, nh = wq.shape[1], a.num_heads
            hd = E // nh
            hdp = padded_head_dim(hd)
            dev = wq.device

            def pad_heads(w):                      # [nh * hd, ...] -> [nh * hdp, ...], zero rows behind every head
                if hdp == hd:
                    return w.contiguous()
                out = torch.zeros((nh, hdp) + tuple(w.shape[1:]), dtype=torch.float32, device=dev)
                out[:, :hd] = w.reshape((nh, hd) + tuple(w.shape[1:]))
                return out.reshape((nh * hdp,) + tuple(w.shape[1:]))

            wproj, bproj = a.proj.weight.detach().float(), a.proj.bias.detach().float()
            w2, b2 = m.fc2.weight.detach().float(), m.fc2.bias.detach().float()
            if self.beit_like_gamma:               # x + gamma * f(x) (vit_eva.py:296-298): the layer scale folds into f's last linear
                g1, g2 = self.gamma_1.detach().float(), self.gamma_2.detach().float()
                wproj, bproj = wproj * g1[:, None], bproj * g1
                w2, b2 = w2 * g2[:, None], b2 * g2
            if hdp != hd:                          # zero COLUMNS of the out projection where the attention output is padding
                wproj = pad_heads(wproj.t().contiguous()).t().contiguous()
            hid = m.fc1.weight.shape[0]
            hid_pad = round_up(hid, 64)
            w1 = torch.zeros((hid_pad, E), dtype=torch.float32, device=dev)     # zero rows: gelu(0) = 0 exactly in the K padding
            w1[:hid] = m.fc1.weight.detach().float()
            b1 = torch.zeros((hid_pad,), dtype=torch.float32, device=dev)
            b1[:hid] = m.fc1.bias.detach().float()
            P = dict(
                hd=hd, hdp=hdp, Ep=nh * hdp,
                wqk=pack_matrix(torch.cat([pad_heads(wq), pad_heads(wk)], 0), dt), bqk=torch.cat([pad_heads(bq), pad_heads(bk)]).contiguous(),
                wv=pack_matrix(pad_heads(wv), dt), bv=pad_heads(bv).contiguous(),
                wproj=pack_matrix(wproj, dt), bproj=bproj.contiguous(),
                n1=(f32(self.norm1.weight), f32(self.norm1.bias), self.norm1.eps),
                n2=(f32(self.norm2.weight), f32(self.norm2.bias), self.norm2.eps),
                w1=pack_matrix(w1, dt), b1=b1, hid_pad=hid_pad, w2=pack_matrix(w2, dt, kpad=64), b2=b2.contiguous())
            if a.use_rel_pos:
                rh, rw = resized_rel_pos(a.rel_pos_h, group, a.interp_type), resized_rel_pos(a.rel_pos_w, group, a.interp_type)
                nr = rh.shape[0] + rw.shape[0]
                # [Rh ; Rw] as the weight of the q . R^T GEMM: K = the padded head width (q's padding columns are zero), N padded to 64
                P.update(rcat=pack_matrix(torch.cat([rh, rw], 0), dt, kpad=hdp, rows=round_up(nr, 64)), ext=ext_width(hd, group, group))
            return P
        return self._pack.get(self, dt, build)

    def forward_tokens(self, x, dt, coords, nwin, ntok_win, vt_buf, last=False, images=1):
        """pre-norm block (vit_eva.py:283-308): x [images * N, E] fp32 residual stream (window-major per image).  coords = (ty, tx)
        int32: position of each token of an attention group inside it.  Returns the new stream."""
        group = self.window_size if self.window_size > 0 else int(round(math.sqrt(x.shape[0] // images)))
        P = self.packed(dt, group)
        a = self.attn
        nh, hd, hdp, Ep = a.num_heads, P["hd"], P["hdp"], P["Ep"]
        xn = ops.layernorm(x, P["n1"][0], P["n1"][1], P["n1"][2], out_dtype=dt)
        vjob = ops.fork(lambda: ops.gemm(xn, P["wv"], P["bv"], trans_out=True, out=vt_buf))
        qk = ops.gemm(xn, P["wqk"], P["bqk"])
        batch, n = (nwin, ntok_win) if self.window_size > 0 else (images, x.shape[0] // images)
        if a.use_rel_pos:
            # rows (token, q head | k head): the k rows of this product are unused (one GEMM over the q|k buffer as it lies in HBM)
            t = ops.gemm(qk.view(-1, hdp), P["rcat"], None)
            qe, ke = ops.relpos_extend(qk[:, :Ep], qk[:, Ep:], t, coords[0], coords[1], heads=nh, head_stride=hdp, head_dim=hd,
                                       hk=group, wk=group, ext_dim=P["ext"], scale=a.scale, t_rows_per_token=2 * nh)
            vt = vjob.join()
            o = ops.attention(qe, ke, vt, batch=batch, n=n, heads=nh, head_dim=P["ext"], v_head_dim=hdp, scale=1.0)
        else:
            vt = vjob.join()
            o = ops.attention(qk[:, :Ep], qk[:, Ep:], vt, batch=batch, n=n, heads=nh, head_dim=hdp, scale=a.scale)
        x = ops.gemm(o, P["wproj"], P["bproj"], residual=x, out_dtype=torch.float32)
        xn = ops.layernorm(x, P["n2"][0], P["n2"][1], P["n2"][2], out_dtype=dt)
        hbuf = torch.empty((xn.shape[0], P["hid_pad"]), dtype=dt, device=xn.device)
        ops.gemm(xn, P["w1"], P["b1"], act=ops.ACT_GELU, out=hbuf)
        return ops.gemm(hbuf, P["w2"], P["b2"], residual=x, out_dtype=dt if last else torch.float32)


class ViT(Backbone):
    def __init__(self, img_size=1024, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0, qkv_bias=True,
                 drop_path_rate=0.0, norm_layer=partial(nn.LayerNorm, eps=1e-6), act_layer=nn.GELU, use_abs_pos=True, use_rel_pos=False,
                 rel_pos_zero_init=True, window_size=0, window_block_indexes=(), residual_block_indexes=(), use_act_checkpoint=False,
                 pretrain_img_size=224, pretrain_use_cls_token=True, out_feature="last_feat", beit_like_qkv_bias=True,
                 beit_like_gamma=False, freeze_patch_embed=False, interp_type="vitdet", frozen_stages=-1):
        super().__init__()
        assert use_abs_pos and patch_size == 16 and len(residual_block_indexes) == 0, \
            "ape_amd vit_eva: absolute position embedding, 16 x 16 patches, no convolutional residual blocks (every APE config)"
        assert window_size > 0 and (img_size // patch_size) % window_size == 0, "token grid must be a multiple of the window size"
        self.pretrain_use_cls_token = pretrain_use_cls_token
        self.img_size, self.patch_size, self.embed_dim, self.window_size = img_size, patch_size, embed_dim, window_size
        self.patch_embed = PatchEmbed(in_chans=in_chans, embed_dim=embed_dim)
        num_patches = (pretrain_img_size // patch_size) ** 2
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + (1 if pretrain_use_cls_token else 0), embed_dim))
        if beit_like_qkv_bias:
            qkv_bias = False                                                                # vit_eva.py:399-400
        hw = img_size // patch_size
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, norm_layer=norm_layer, act_layer=act_layer,
                  use_rel_pos=use_rel_pos, rel_pos_zero_init=rel_pos_zero_init, window_size=window_size if i in window_block_indexes else 0,
                  use_residual_block=False, input_size=(hw, hw), beit_like_qkv_bias=beit_like_qkv_bias, beit_like_gamma=beit_like_gamma,
                  interp_type=interp_type) for i in range(depth)])
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
            ws = self.window_size
            dev = self.pos_embed.device
            t2r, r2t = window_major_order(hw, hw, ws)
            t2r, r2t = t2r.to(dev), r2t.to(dev)
            pos = self.pos_embed.detach().float()                      # get_abs_pos: drop cls, bicubic resize to the token grid
            if self.pretrain_use_cls_token:
                pos = pos[:, 1:]
            size = int(math.sqrt(pos.shape[1]))
            if size != hw:
                pos = F.interpolate(pos.reshape(1, size, size, -1).permute(0, 3, 1, 2), size=(hw, hw), mode="bicubic",
                                    align_corners=False).permute(0, 2, 3, 1)
            pos = pos.reshape(hw * hw, -1)[t2r.long()].contiguous()
            w = self.patch_embed.proj.weight
            j = torch.arange(ws * ws, device=dev)
            return dict(hw=hw, t2r=t2r, r2t=r2t, pos=pos, wpe=pack_matrix(w.reshape(w.shape[0], -1), dt), bpe=f32(self.patch_embed.proj.bias),
                        # a token's (row, col) inside its attention group: the window / the whole grid (window-major token order)
                        coords_win=((j // ws).to(torch.int32).contiguous(), (j % ws).to(torch.int32).contiguous()),
                        coords_glb=((t2r // hw).to(torch.int32).contiguous(), (t2r % hw).to(torch.int32).contiguous()))
        return self._pack.get(self, dt, build)

    def forward_tokens(self, image, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0), stages=None):
        """image [3,h,w] fp32 (h,w <= img_size), or a list of B such images -> last feature [B * N, E] in the compute dtype,
        WINDOW-MAJOR token order per image (same contract as vit_eva_clip.ViT.forward_tokens)"""
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
        Ep = self.blocks[0].attn.num_heads * padded_head_dim(self.embed_dim // self.blocks[0].attn.num_heads)
        vt_buf = ops.zeros((Ep, round_up(B * n, 64)), dt, x.device)
        for i, blk in enumerate(self.blocks):
            coords = P["coords_win"] if blk.window_size > 0 else P["coords_glb"]
            x = blk.forward_tokens(x, dt, coords, nwin, self.window_size ** 2, vt_buf, last=(i == len(self.blocks) - 1), images=B)
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
