"""Golden vectors of the EVA-02 MIM ViT in its APE-L_A/B/C configuration (configs/common/backbone/vitl_eva02.py:10-41:
subln=True, naiveswiglu=True, windows that tile the grid, every sixth block global) at reduced size: the reference's own
`ape/modeling/backbone/vit_eva02.py` ViT (executed from /root/reference through oracle/refshim) on a seeded image with
seeded weights.  Run in the build container:  python tests/golden/make_eva02_subln_golden.py  ->  ref_eva02_subln.pt"""
import os
import sys
from functools import partial

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import refshim, weights  # noqa: E402

CFG = dict(img_size=256, patch_size=16, embed_dim=128, depth=6, num_heads=2, window_size=8, mlp_ratio=4 * 2 / 3, qkv_bias=True,
           window_block_indexes=[0, 1, 2, 3, 4], residual_block_indexes=[], use_rel_pos=True, out_feature="last_feat",
           use_act_checkpoint=False, xattn=True, subln=True, swiglu=False, naiveswiglu=True)


def main():
    refshim.install()
    V = sys.modules["ape.modeling.backbone.vit_eva02"]
    net = V.ViT(norm_layer=partial(nn.LayerNorm, eps=1e-6), drop_path_rate=0.0, **CFG).eval()
    spec = [(k, list(v.shape)) for k, v in net.state_dict().items()]
    sd = weights.make_state_dict(spec, seed=7)
    weights.load_into(net, sd)
    image = torch.randint(0, 256, (3, 256, 256), generator=torch.Generator().manual_seed(8)).float()
    x = (image - 120.0) / 60.0
    with torch.no_grad():
        feat = net(x[None])["last_feat"][0]                                  # [E, 16, 16]
    out = {"cfg": CFG, "spec": spec, "wseed": 7, "iseed": 8, "last_feat": feat}
    torch.save(out, os.path.join(ROOT, "tests", "golden", "ref_eva02_subln.pt"))
    print(feat.shape, float(feat.abs().max()))


if __name__ == "__main__":
    main()
