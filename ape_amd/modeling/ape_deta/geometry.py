"""Per-image-size constants of the deformable transformer (computed once per (h, w), cached on the device).

Everything here depends only on the padded square size S, the un-padded image size (h, w) and the level strides --
not on pixel values -- so it is hoisted out of the per-image path:
  * padding masks per level                      (deformable_detr_segm_vl.py:382-385: nearest resize of img_masks)
  * sine position embeddings                     (detrex PositionEmbeddingSine, deformable_detr_segm_vl.py:386-388)
  * valid ratios / encoder reference points      (deformable_transformer_vl.py:402-410, 371-400)
  * two-stage anchors, their validity, level ids (deformable_transformer_vl.py:321-359)
These are small device-side torch computations (no HIP kernel: they run once per distinct image size).
"""
import math

import torch


class LevelGeometry:
    __slots__ = ("shapes", "starts", "T", "mask", "mask_u8", "pos", "valid_ratios", "enc_ref", "proposals", "invalid_u8",
                 "level_ids", "image_size", "_lvl_pos", "box_scale", "vr4", "arange_T")

    def __init__(self):
        self._lvl_pos = {}


class StaticGeometry(LevelGeometry):
    """A LevelGeometry whose tensors are FIXED buffers that `load()` overwrites with the constants of another image size:
    a captured hipGraph bakes tensor addresses, so handing the forward this object (instead of the per-size cached one)
    lets ONE graph serve every (h, w) inside the square pad.  Level shapes / starts depend on the pad only."""
    FIELDS = ("mask", "mask_u8", "valid_ratios", "enc_ref", "proposals", "invalid_u8", "box_scale", "vr4")

    def __init__(self, geo, lvl_pos_key, lvl_pos):
        super().__init__()
        self.shapes, self.starts, self.T = geo.shapes, geo.starts, geo.T
        self.level_ids, self.arange_T, self.pos = geo.level_ids, geo.arange_T, None
        self.image_size = None
        for f in self.FIELDS:
            setattr(self, f, getattr(geo, f).clone())
        self._lvl_pos = {lvl_pos_key: lvl_pos.clone()}

    def generate(self, square, image_size, pos_cfg, level_embeds):
        """overwrite the buffers with the constants of `image_size` = (h, w), computed on the device by one kernel
        (csrc/geometry.hip restates build_geometry / lvl_pos operation by operation)"""
        from ... import ops
        h, w = image_size
        lp = next(iter(self._lvl_pos.values()))
        dim_t = dim_t_table(pos_cfg["num_pos_feats"], pos_cfg["temperature"], lp.device)
        ops.geometry(square, h, w, self.shapes, dim_t, level_embeds, pos_cfg["offset"], pos_cfg["eps"], pos_cfg["scale"],
                     lvl_pos=lp, mask_u8=self.mask_u8, mask=self.mask.view(torch.uint8), invalid_u8=self.invalid_u8,
                     enc_ref=self.enc_ref, proposals=self.proposals, valid_ratios=self.valid_ratios, vr4=self.vr4,
                     box_scale=self.box_scale)
        self.image_size = (h, w)

    def load(self, geo, lvl_pos):
        for f in self.FIELDS:
            getattr(self, f).copy_(getattr(geo, f), non_blocking=True)
        next(iter(self._lvl_pos.values())).copy_(lvl_pos, non_blocking=True)
        self.image_size = geo.image_size


_DIM_T = {}


def dim_t_table(num_pos_feats, temperature, device):
    """temperature ** (2*(i//2)/n), computed on the HOST once per device: pow() may differ by an ulp between host and
    device libms, and the sine embedding of fully padded rows/columns (argument ~ -3e6) is a chaotic function of it"""
    key = (num_pos_feats, float(temperature), str(device))
    if key not in _DIM_T:
        d = torch.arange(num_pos_feats, dtype=torch.float32)
        _DIM_T[key] = (temperature ** (2 * torch.div(d, 2, rounding_mode="floor") / num_pos_feats)).to(device)
    return _DIM_T[key]


def _sine_pos(mask, num_pos_feats, temperature, normalize, offset, eps, scale):
    """mask [H,W] bool -> [H*W, 2*num_pos_feats] fp32 (token-major layout of the reference's [C,H,W] embedding)"""
    not_mask = ~mask
    y_embed = not_mask.cumsum(0, dtype=torch.float32)
    x_embed = not_mask.cumsum(1, dtype=torch.float32)
    if normalize:
        y_embed = (y_embed + offset) / (y_embed[-1:, :] + eps) * scale
        x_embed = (x_embed + offset) / (x_embed[:, -1:] + eps) * scale
    dim_t = dim_t_table(num_pos_feats, temperature, mask.device)
    pos_x = x_embed[:, :, None] / dim_t
    pos_y = y_embed[:, :, None] / dim_t
    H, W = mask.shape
    pos_x = torch.stack((pos_x[:, :, 0::2].sin(), pos_x[:, :, 1::2].cos()), dim=3).view(H, W, -1)
    pos_y = torch.stack((pos_y[:, :, 0::2].sin(), pos_y[:, :, 1::2].cos()), dim=3).view(H, W, -1)
    return torch.cat((pos_y, pos_x), dim=2).reshape(H * W, -1)


def build_geometry(square, image_size, level_shapes, device, pos_cfg):
    """square: padded size S; image_size (h, w); level_shapes [(H_l, W_l)]; pos_cfg: PositionEmbeddingSine settings"""
    h, w = image_size
    g = LevelGeometry()
    g.image_size = (h, w)
    g.shapes = [(int(a), int(b)) for a, b in level_shapes]
    g.starts = [0]
    for a, b in g.shapes[:-1]:
        g.starts.append(g.starts[-1] + a * b)
    g.T = sum(a * b for a, b in g.shapes)
    img_mask = torch.ones((square, square), dtype=torch.float32, device=device)
    img_mask[:h, :w] = 0
    masks, poss, vrs, refs, props, lids = [], [], [], [], [], []
    for lvl, (H, W) in enumerate(g.shapes):
        # F.interpolate(img_masks[None], size=(H, W)) default mode = nearest: src = floor(dst * S / H)
        iy = (torch.arange(H, device=device, dtype=torch.float32) * (square / H)).floor().long().clamp_(max=square - 1)
        ix = (torch.arange(W, device=device, dtype=torch.float32) * (square / W)).floor().long().clamp_(max=square - 1)
        m = img_mask[iy][:, ix].to(torch.bool)
        masks.append(m.reshape(-1))
        poss.append(_sine_pos(m, pos_cfg["num_pos_feats"], pos_cfg["temperature"], pos_cfg["normalize"], pos_cfg["offset"],
                              pos_cfg["eps"], pos_cfg["scale"]))
        valid_H = torch.sum(~m[:, 0]).float()
        valid_W = torch.sum(~m[0, :]).float()
        vrs.append(torch.stack([valid_W / W, valid_H / H]))
        lids.append(torch.full((H * W,), lvl, dtype=torch.long, device=device))
    g.mask = torch.cat(masks)
    g.mask_u8 = g.mask.to(torch.uint8).contiguous()
    g.pos = torch.cat(poss).contiguous()
    g.valid_ratios = torch.stack(vrs)  # [L, 2] (w, h)
    g.level_ids = torch.cat(lids)
    cur = 0
    for lvl, (H, W) in enumerate(g.shapes):
        vr = g.valid_ratios[lvl]
        ry, rx = torch.meshgrid(torch.linspace(0.5, H - 0.5, H, dtype=torch.float32, device=device),
                                torch.linspace(0.5, W - 0.5, W, dtype=torch.float32, device=device), indexing="ij")
        ref = torch.stack((rx.reshape(-1) / (vr[0] * W), ry.reshape(-1) / (vr[1] * H)), -1)
        refs.append(ref)
        m = g.mask[cur:cur + H * W].view(H, W)
        valid_H = torch.sum(~m[:, 0])
        valid_W = torch.sum(~m[0, :])
        gy, gx = torch.meshgrid(torch.linspace(0, H - 1, H, dtype=torch.float32, device=device),
                                torch.linspace(0, W - 1, W, dtype=torch.float32, device=device), indexing="ij")
        grid = torch.cat([gx.unsqueeze(-1), gy.unsqueeze(-1)], -1)
        scale = torch.stack([valid_W, valid_H]).view(1, 1, 2)
        grid = (grid + 0.5) / scale
        wh = torch.ones_like(grid) * 0.05 * (2.0 ** lvl)
        props.append(torch.cat((grid, wh), -1).view(-1, 4))
        cur += H * W
    ref = torch.cat(refs)                                            # [T, 2]
    g.enc_ref = (ref[:, None, :] * g.valid_ratios[None]).contiguous()   # [T, L, 2]
    prop = torch.cat(props)
    valid = ((prop > 0.01) & (prop < 0.99)).all(-1, keepdim=True)
    prop = torch.log(prop / (1 - prop))
    prop = prop.masked_fill(g.mask.unsqueeze(-1), float("inf")).masked_fill(~valid, float("inf"))
    g.proposals = prop.contiguous()                                  # logit-space anchors, inf where unusable
    g.invalid_u8 = (g.mask | ~valid[:, 0]).to(torch.uint8).contiguous()
    g.box_scale = torch.tensor([w, h, w, h], dtype=torch.float32).to(device)
    g.vr4 = torch.cat([g.valid_ratios, g.valid_ratios], -1).contiguous()
    g.arange_T = torch.arange(g.T, device=device)
    return g


def geometry_from_masks(level_masks, level_pos):
    """the same constants from the padding masks / position embeddings a CALLER supplies, level by level -- the inputs of the
    reference-signature `DeformableDetrTransformerVL.forward` (deformable_transformer_vl.py:422-477, 321-410).
    level_masks: [H_l, W_l] bool (True = padding); level_pos: [H_l * W_l, C] token-major position embeddings."""
    device = level_masks[0].device
    g = LevelGeometry()
    g.image_size = None
    g.shapes = [(int(m.shape[0]), int(m.shape[1])) for m in level_masks]
    g.starts = [0]
    for a, b in g.shapes[:-1]:
        g.starts.append(g.starts[-1] + a * b)
    g.T = sum(a * b for a, b in g.shapes)
    g.mask = torch.cat([m.reshape(-1).bool() for m in level_masks])
    g.mask_u8 = g.mask.to(torch.uint8).contiguous()
    g.pos = torch.cat([p.float() for p in level_pos]).contiguous()
    vrs, refs, props, lids = [], [], [], []
    for lvl, m in enumerate(level_masks):
        H, W = m.shape
        valid_H, valid_W = torch.sum(~m[:, 0]).float(), torch.sum(~m[0, :]).float()
        vrs.append(torch.stack([valid_W / W, valid_H / H]))
        lids.append(torch.full((H * W,), lvl, dtype=torch.long, device=device))
    g.valid_ratios = torch.stack(vrs)
    g.level_ids = torch.cat(lids)
    for lvl, m in enumerate(level_masks):
        H, W = m.shape
        vr = g.valid_ratios[lvl]
        ry, rx = torch.meshgrid(torch.linspace(0.5, H - 0.5, H, dtype=torch.float32, device=device),
                                torch.linspace(0.5, W - 0.5, W, dtype=torch.float32, device=device), indexing="ij")
        refs.append(torch.stack((rx.reshape(-1) / (vr[0] * W), ry.reshape(-1) / (vr[1] * H)), -1))
        valid_H, valid_W = torch.sum(~m[:, 0]), torch.sum(~m[0, :])
        gy, gx = torch.meshgrid(torch.linspace(0, H - 1, H, dtype=torch.float32, device=device),
                                torch.linspace(0, W - 1, W, dtype=torch.float32, device=device), indexing="ij")
        grid = (torch.cat([gx.unsqueeze(-1), gy.unsqueeze(-1)], -1) + 0.5) / torch.stack([valid_W, valid_H]).view(1, 1, 2)
        wh = torch.ones_like(grid) * 0.05 * (2.0 ** lvl)
        props.append(torch.cat((grid, wh), -1).view(-1, 4))
    ref = torch.cat(refs)
    g.enc_ref = (ref[:, None, :] * g.valid_ratios[None]).contiguous()
    prop = torch.cat(props)
    valid = ((prop > 0.01) & (prop < 0.99)).all(-1, keepdim=True)
    prop = torch.log(prop / (1 - prop))
    g.proposals = prop.masked_fill(g.mask.unsqueeze(-1), float("inf")).masked_fill(~valid, float("inf")).contiguous()
    g.invalid_u8 = (g.mask | ~valid[:, 0]).to(torch.uint8).contiguous()
    g.box_scale = None
    g.vr4 = torch.cat([g.valid_ratios, g.valid_ratios], -1).contiguous()
    g.arange_T = torch.arange(g.T, device=device)
    return g


def proposal_pos_embed(coords_unact, num_pos_feats=128, temperature=10000):
    """deformable_transformer_vl.py:412-420 for [Q,4] unactivated coords -> [Q, 512] fp32"""
    scale = 2 * math.pi
    dim_t = dim_t_table(num_pos_feats, temperature, coords_unact.device)
    p = coords_unact.sigmoid() * scale
    pos = p[:, :, None] / dim_t
    return torch.stack((pos[:, :, 0::2].sin(), pos[:, :, 1::2].cos()), dim=3).flatten(1)


def inverse_sigmoid(x, eps=1e-3):
    """detrex.utils.inverse_sigmoid (eps 1e-3; call sites deformable_transformer_vl.py:237, deformable_detr_segm_vl.py:490)"""
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))
