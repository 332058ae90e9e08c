"""helpers for the host-model tests (CPU with fake ops, GPU with the HIP ops)"""
import torch

import oracle_util as U
from ape_amd.modeling.build import build_ape
from oracle import ape_oracle, weights
from oracle.configs import CONFIGS


def build_pair(case, device="cpu", dtype=torch.float32):
    """(our model with the oracle's seeded weights, oracle, image, text, golden)"""
    gold = U.load_golden(case)
    cfg_name, wseed, image, text = U.case_inputs(gold)
    sd = weights.make_state_dict(U.load_spec(cfg_name), wseed)
    model = build_ape(cfg_name)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(m.endswith(("freqs_cos", "freqs_sin")) for m in missing), (missing, unexpected)
    model.to(device)
    model.model_vision.set_compute_dtype(dtype)
    orc = ape_oracle.ApeOracle(CONFIGS[cfg_name], sd)
    return model, orc, image, text, gold


def token_major(k, t):
    """oracle stage tensor (NCHW / batch-first) -> our token-major layout"""
    if k in ("p2", "p3", "p4", "p5", "p6", "mask_features"):
        return t[0].permute(1, 2, 0).reshape(-1, t.shape[1])
    if k == "enc_class":
        return t[0, :, 0]
    if t.dim() >= 3 and t.shape[0] == 1:
        return t[0]
    return t


def set_overlap(a, b):
    sa, sb = set(a.tolist()), set(b.tolist())
    return len(sa & sb) / max(len(sb), 1)
