"""box AP of the bf16 pipeline vs the fp32 pipeline's detections over 16 seeded images, next to the control (the fp32 pipeline on
input noise below 8-bit quantisation) -- the `box_ap_vs_fp32_pipeline` entry of bench.py's parity object on its own."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ape_amd.modeling.build import build_ape, init_synthetic  # noqa: E402


def main():
    dev = "cuda"
    model = init_synthetic(build_ape("L_D_coco"), seed=0).to(dev).eval()      # the bench's model
    mv = model.model_vision
    text = torch.randn(80, 1024, generator=torch.Generator().manual_seed(3)).to(dev)
    imgs = bench.make_images(16, 1024, seed=7000, device=dev)
    with torch.no_grad():
        r = bench.box_ap_vs_fp32(mv, imgs, text)
    print(json.dumps(r))


if __name__ == "__main__":
    main()
