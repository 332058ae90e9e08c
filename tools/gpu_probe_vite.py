"""ViT-e (vite_eva02_clip_1024.py: width 1792 = 16 x 112, GELU MLP 15360, post-norm, every fourth block global) at 1024^2 on the
HIP kernels: time of an 8-block slice (6 windowed + 2 global, random weights) for 1 and 2 stacked images, extrapolated to the 64
blocks of the full backbone.     python tools/gpu_probe_vite.py"""
import os
import sys
import time
from functools import partial

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ape_amd.modeling.backbone.vit_eva_clip import ViT  # noqa: E402


def main():
    dev = "cuda"
    depth = 8
    torch.manual_seed(0)
    net = ViT(img_size=1024, patch_size=16, embed_dim=1792, depth=depth, num_heads=16, drop_path_rate=0.0, window_size=32,
              mlp_ratio=8.571428571428571, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6),
              window_block_indexes=[i for i in range(depth) if i % 4 != 3], residual_block_indexes=[], use_rel_pos=True,
              out_feature="last_feat", xattn=True, pretrain_img_size=224, pretrain_use_cls_token=True, postnorm=True).to(dev)
    for p in net.parameters():
        if p.dim() > 1:
            nn.init.normal_(p, std=0.02)
    net.compute_dtype = torch.bfloat16
    imgs = [torch.randint(0, 256, (3, 1024, 1024), generator=torch.Generator().manual_seed(i)).float().to(dev) for i in range(2)]
    mean, std = (120.0, 120.0, 120.0), (60.0, 60.0, 60.0)
    # per block and image: qkv 3 x 2 x 4096 x 1792 x 2048 (heads padded to 128), proj 2 x 4096 x 2048 x 1792, MLP 2 x 2 x 4096 x 1792 x 15360,
    # attention 4 x N_k x 4096 x 128 x 16 heads (N_k = 1024 windowed, 4096 global)
    lin = 3 * 2 * 4096 * 1792 * 2048 + 2 * 4096 * 2048 * 1792 + 2 * 2 * 4096 * 1792 * 15360
    att = lambda nk: 4 * nk * 4096 * 128 * 16
    flops = 6 * (lin + att(1024)) + 2 * (lin + att(4096))
    for B in (1, 2):
        x = imgs[:B] if B > 1 else imgs[0]
        for _ in range(2):
            out = net.forward_tokens(x, mean, std)
        torch.cuda.synchronize()
        assert torch.isfinite(out.float()).all()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            net.forward_tokens(x, mean, std)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / n
        print(f"ViT-e slice of {depth} blocks, {B} image(s): {ms:.2f} ms = {ms / depth / B:.3f} ms per block and image, "
              f"{B * flops / ms / 1e9:.0f} TF/s; 64 blocks: {64 * ms / depth / B:.1f} ms per image")
    # fp32 validation kernels vs bf16 on the same weights: the head-padded attention + post-norm path at full width
    net.compute_dtype = torch.float32
    ref = net.forward_tokens(imgs[0], mean, std).float()
    net.compute_dtype = torch.bfloat16
    got = net.forward_tokens(imgs[0], mean, std).float()
    e = float((got - ref).abs().max() / ref.abs().max())
    r = float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    print(f"bf16 vs fp32 kernels after {depth} blocks at full width: max {e:.2e} rms {r:.2e}")


if __name__ == "__main__":
    main()
