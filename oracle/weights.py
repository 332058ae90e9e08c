"""Deterministic synthetic weights keyed by the reference's state-dict names.

TEST INFRASTRUCTURE (oracle).  No checkpoint can be downloaded here, so parity runs use seeded NON-degenerate
weights: the reference zero-initialises sampling_offsets.weight, attention_weights.* and bbox_embed[-1]
(multi_scale_deform_attn.py:194-209, deformable_detr.py:119-120), which would make the deformable path vacuous.
The spec (name, shape) list is the reference model's own state_dict() (tests/golden/state_spec_<cfg>.json,
written by tests/golden/make_golden.py), so it doubles as the checkpoint-key contract.
"""
import math

import torch

COMPUTED = ("freqs_cos", "freqs_sin")  # RoPE tables: persistent buffers, recomputed, never randomised


def _canonical(name):
    for sub in ("class_embed.", "bbox_embed."):
        alias = "transformer.decoder." + sub
        if alias in name:
            return name.replace(alias, sub)
    return name


def make_state_dict(spec, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd = {}
    for name, shape in sorted((n, tuple(s)) for n, s in spec):
        if name.endswith(COMPUTED):
            continue
        # class_embed / bbox_embed are the SAME module objects under two prefixes (deformable_detr.py:153-168):
        # both names must carry one tensor
        canon = _canonical(name)
        if canon != name and canon in sd:
            sd[name] = sd[canon]
            continue
        leaf = name.rsplit(".", 1)[-1]
        r = torch.randn(shape, generator=g, dtype=torch.float32)
        if name.endswith("name_prompt_fusion_feature"):
            t = torch.zeros(shape)
        elif leaf in ("gamma_v", "gamma_l"):
            t = 1.0 / 6 + 0.02 * r
        elif leaf == "log_scale":
            t = torch.zeros(shape)
        elif leaf == "bias0" or name.endswith("class_embed.6.bias") or "class_embed_ambiguous" in name and leaf == "bias":
            t = -math.log((1 - 0.01) / 0.01) + 0.1 * r
        elif leaf == "level_embeds":
            t = r
        elif leaf == "pos_embed":
            t = 0.02 * r
        elif name.endswith("sampling_offsets.bias"):
            # the reference's ring initialisation (multi_scale_deform_attn.py:195-207) plus noise
            L = shape[0] // (8 * 4 * 2)
            th = torch.arange(8, dtype=torch.float32) * (2.0 * math.pi / 8)
            grid = torch.stack([th.cos(), th.sin()], -1)
            grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(8, 1, 1, 2).repeat(1, L, 4, 1)
            for i in range(4):
                grid[:, :, i, :] *= i + 1
            t = grid.reshape(-1) + 0.1 * r
        elif len(shape) >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            gain = 0.5 if "sampling_offsets" in name else 1.0
            t = r * (gain / math.sqrt(fan_in))
        elif leaf == "weight":  # norm scales
            t = 1.0 + 0.1 * r
        else:  # biases, q_bias, v_bias, bias_lang, in_proj_bias
            t = 0.02 * r
        sd[name] = t.contiguous()
    return sd


def load_into(module, sd):
    """copy a make_state_dict() result into an nn.Module (strict on everything except the computed buffers)"""
    own = module.state_dict()
    missing = [k for k in own if k not in sd and not k.endswith(COMPUTED)]
    extra = [k for k in sd if k not in own]
    assert not missing and not extra, f"state-dict mismatch: missing={missing[:5]} extra={extra[:5]}"
    with torch.no_grad():
        for k, v in sd.items():
            own[k].copy_(v)
