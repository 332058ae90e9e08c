"""Golden vectors of the CLIP text tower: the reference's own tokenizer and TextTransformer (executed from /root/reference
through oracle/refshim.install_text) on fixed strings with seeded weights (oracle/text_oracle.make_state_dict).
Run in the build container:  python tests/golden/make_text_golden.py  ->  tests/golden/ref_text_tower.pt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import refshim, text_oracle as T  # noqa: E402

TEXTS = ["person", "bicycle", "traffic light", "fire hydrant", "hot dog", "teddy bear", "hair drier", "a photo of a cat",
         "the quick brown fox jumps over the lazy dog near the river bank", "don't they've I'm", "naïve café 123 &amp; more",
         "the man in the red shirt standing to the left of the woman holding an umbrella"]


def main():
    tr, tok = refshim.install_text()
    tokens = tok.tokenize(TEXTS, context_length=77)
    out = {"texts": TEXTS, "tokens": tokens}
    for name, cfg, seed in (("tiny", T.TINY, 0), ("wide", dict(T.TINY, width=256, heads=4, layers=2, embed_dim=64), 1)):
        sd = T.make_state_dict(cfg, seed)
        m = T.reference_text_tower(cfg, sd)
        with torch.no_grad():
            eot = m(tokens)
            full = m(tokens, return_all_features=True) @ m.text_projection              # clip_wrapper_eva02.py:144
        out[name] = {"cfg": cfg, "seed": seed, "eot": eot, "full": full, "keys": sorted(m.state_dict().keys())}
    torch.save(out, os.path.join(ROOT, "tests", "golden", "ref_text_tower.pt"))
    print({k: (v["eot"].shape, float(v["eot"].abs().max())) for k, v in out.items() if isinstance(v, dict)})


if __name__ == "__main__":
    main()
