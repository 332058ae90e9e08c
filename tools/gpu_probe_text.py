"""Text tower timing on the MI355X: APE-L_D's EVA02-CLIP-bigE-14-plus text tower (32 x 1280, random weights) on COCO-sized
(80) and LVIS-sized (1203) vocabularies of synthetic 4-token names, truncated context vs all 77 positions."""
import sys
import time

import torch

sys.path.insert(0, ".")
from ape_amd.modeling.text import EVA02CLIP  # noqa: E402


def main():
    torch.manual_seed(0)
    m = EVA02CLIP("EVA02-CLIP-bigE-14-plus", dtype="float16").cuda()
    for K in (80, 1203):
        tok = torch.zeros((K, 77), dtype=torch.long)
        tok[:, 0] = 49406
        tok[:, 1:5] = torch.randint(1000, 40000, (K, 4))
        tok[:, 5] = 49407
        tok = tok.cuda()
        for allpos in (False, True):
            m.all_positions = allpos
            m.forward_tokens(tok)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                out = m.forward_tokens(tok)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            L = 77 if allpos else 8
            flops = 2 * K * L * 32 * (4 * 1280 * 1280 + 2 * 1280 * 5120)
            print(f"text tower bigE K={K} positions={L}: {dt * 1e3:.1f} ms  ({flops / dt / 1e12:.0f} TF/s linears)  finite={bool(torch.isfinite(out['last_hidden_state_eot']).all())}")


if __name__ == "__main__":
    main()
