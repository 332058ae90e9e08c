"""helpers shared by the oracle / parity tests (test infrastructure)"""
import json
import os

import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_spec(cfg_name):
    from oracle.configs import spec_name

    with open(os.path.join(GOLDEN, f"state_spec_{spec_name(cfg_name)}.json")) as fh:
        return [(n, tuple(s)) for n, s in json.load(fh)]


def load_golden(case):
    return torch.load(os.path.join(GOLDEN, f"ref_{case}.pt"), weights_only=False)


def case_inputs(gold):
    cfg, wseed, iseed, (h, w), K, tseed = gold["case"][:6]
    if "jpeg" in gold:                     # a real photograph: the file bytes travel in the fixture (tests/golden/make_golden.py)
        import io

        import numpy as np
        from PIL import Image
        rgb = Image.open(io.BytesIO(gold["jpeg"].numpy().tobytes())).convert("RGB")
        image = torch.from_numpy(np.asarray(rgb.resize((w, h), Image.BILINEAR)).astype("float32").transpose(2, 0, 1).copy())
    else:
        image = torch.randint(0, 256, (3, h, w), generator=torch.Generator().manual_seed(iseed)).float()
    text = torch.randn(K, 1024, generator=torch.Generator().manual_seed(tseed))
    return cfg, wseed, image, text


def case_mask_prompt(gold, hw):
    """the prompt mask of a "mask" case (tests/golden/make_golden.py case_mask_prompt): a rectangle over the middle of the image"""
    case = gold["case"]
    if len(case) <= 9 or case[9] != "mask":
        return None
    h, w = hw
    m = torch.zeros(h, w)
    m[h // 5: (3 * h) // 5, w // 4: (3 * w) // 4] = 255.0
    return m


def case_prompt(gold):
    return gold["case"][6] if len(gold["case"]) > 6 else "name"


def relerr(a, b):
    a, b = a.float(), b.float()
    fin = torch.isfinite(b)
    assert torch.equal(torch.isfinite(a), fin), "non-finite pattern differs"
    if not fin.any():
        return 0.0
    return ((a[fin] - b[fin]).abs().max() / b[fin].abs().max().clamp_min(1e-12)).item()


def check_fingerprint(t, fp, tol, name):
    assert list(t.shape) == fp["shape"], f"{name}: shape {list(t.shape)} vs {fp['shape']}"
    got = t.detach().reshape(-1)[fp["idx"]].float().cpu()
    want = fp["samples"].float()
    fin = torch.isfinite(want)
    assert torch.equal(torch.isfinite(got), fin), f"{name}: non-finite pattern differs"
    if fin.any():
        err = ((got[fin] - want[fin]).abs().max() / max(fp["absmax"], 1e-12)).item()
        assert err < tol, f"{name}: relerr {err:.3e} >= {tol}"
        return err
    return 0.0


def match_detections(boxes_a, scores_a, classes_a, boxes_b, scores_b, classes_b, box_tol=1e-2, score_tol=1e-3):
    """fraction of detections in b that have a counterpart in a (same class, close score, close box)"""
    boxes_a, boxes_b = getattr(boxes_a, "tensor", boxes_a), getattr(boxes_b, "tensor", boxes_b)      # Boxes or plain tensors
    matched = 0
    used = set()
    for j in range(len(scores_b)):
        cand = ((classes_a == classes_b[j]) & ((scores_a - scores_b[j]).abs() < score_tol)).nonzero().flatten().tolist()
        for i in cand:
            if i in used:
                continue
            if (boxes_a[i] - boxes_b[j]).abs().max() < box_tol * max(1.0, boxes_b[j].abs().max().item()):
                used.add(i)
                matched += 1
                break
    return matched / max(len(scores_b), 1)
