"""Golden vectors of the EVA-02-CLIP ViT in its ViT-e configuration (configs/common/backbone/vite_eva02_clip_1024.py:9-49:
postnorm=True, packed qkv, GELU Mlp, no rope, head width 112, every fourth block global) at reduced size: the reference's own
`ape/modeling/backbone/vit_eva_clip.py` ViT (executed from /root/reference through oracle/refshim) on a seeded image with seeded
weights.  Run in the build container:  python tests/golden/make_vite_golden.py  ->  ref_vite_small.pt"""
import os
import sys
from functools import partial

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import refshim, weights  # noqa: E402

# 2 heads x 112 (the ViT-e head width), mlp_ratio of the config, 16 x 16 tokens in 8 x 8 windows, blocks 0-2 windowed, 3 global
CFG = dict(img_size=256, patch_size=16, embed_dim=224, depth=4, num_heads=2, window_size=8, mlp_ratio=8.571428571428571,
           qkv_bias=True, window_block_indexes=[0, 1, 2], residual_block_indexes=[], use_rel_pos=True, out_feature="last_feat",
           use_act_checkpoint=False, xattn=True, pretrain_img_size=224, pretrain_use_cls_token=True, postnorm=True)


def main():
    refshim.install()
    V = sys.modules["ape.modeling.backbone.vit_eva_clip"]
    net = V.ViT(norm_layer=partial(nn.LayerNorm, eps=1e-6), drop_path_rate=0.0, **CFG).eval()
    spec = [(k, list(v.shape)) for k, v in net.state_dict().items()]
    sd = weights.make_state_dict(spec, seed=11)
    weights.load_into(net, sd)
    image = torch.randint(0, 256, (3, 256, 256), generator=torch.Generator().manual_seed(12)).float()
    x = (image - 120.0) / 60.0
    feats = {}
    hooks = [blk.register_forward_hook(lambda m, a, o, i=i: feats.__setitem__(i, o.detach().clone())) for i, blk in enumerate(net.blocks)]
    with torch.no_grad():
        feat = net(x[None])["last_feat"][0]                                  # [E, 16, 16]
    for h in hooks:
        h.remove()
    out = {"cfg": CFG, "spec": spec, "wseed": 11, "iseed": 12, "last_feat": feat, "blocks": {i: v[0] for i, v in feats.items()}}
    torch.save(out, os.path.join(ROOT, "tests", "golden", "ref_vite_small.pt"))
    print(feat.shape, float(feat.abs().max()), [float(v.abs().max()) for v in feats.values()])


if __name__ == "__main__":
    main()
