"""Localise a same-rounding mismatch inside a decoder layer: the ops of DeformableDetrTransformerDecoderVL.forward_tokens one by one on the
rounded oracle's inputs, each output against the rounded oracle's intermediate (oracle/rounded.py `debug`).  GPU probe; output -> log."""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import model_util as M  # noqa: E402
import oracle_util as U  # noqa: E402
from ape_amd import ops  # noqa: E402
from ape_amd.packing import round_up  # noqa: E402
from oracle import rounded, weights  # noqa: E402
from oracle.configs import CONFIGS  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "L_D_coco80"
dt = torch.float16 if (len(sys.argv) > 2 and sys.argv[2] == "f16") else torch.bfloat16
dev = "cuda"
model, image, text, gold = M.build_model(case, dev, torch.float32)
cfg_name, wseed = gold["case"][0], gold["case"][1]
sd = weights.make_state_dict(U.load_spec(cfg_name), wseed)
orc = rounded.RoundedApeOracle(CONFIGS[cfg_name], sd, dtype=dt)
orc.debug = {}
orc.forward(image, text)
mv = model.model_vision
mv.set_compute_dtype(dt)
H = orc.hip


def rel(a, b):
    a, b = a.float().cpu().double(), b.float().cpu().double()
    return float((a - b).pow(2).sum().sqrt() / b.pow(2).sum().sqrt().clamp_min(1e-300))


def D(t):
    return t.to(dev)


dec = mv.transformer.decoder
geo = mv.geometry(tuple(image.shape[-2:]), [(image.shape[-1] // s, image.shape[-1] // s) for s in (4, 8, 16, 32, 64)]) if False else None
# geometry through a forward (fills the cache) -- cheaper to just run the model once
st = {}
out = mv.forward_single(image.to(dev), text.to(dev), forced_topk=H["topk_proposals"].to(dev), stages=st)
geo = out["geo"]
P = dec.packed(dt)
E = 256
memory = D(H["memory"]).to(dt)
from ape_amd.layers.multi_scale_deform_attn import half_value_kwargs
value_all = ops.gemm(memory, P["wval"], P["bval"], rowmask=geo.mask_u8, mask_mode=ops.MASK_ZERO_OUTPUT, **half_value_kwargs(dt, memory.shape[0]))
for i in (0, 3):
    dbg = orc.debug[i]
    layer = dec.layers[i]
    out_in = D(H["query_init"] if i == 0 else H[f"dec{i - 1}_out"]).to(dt)
    qpos = D(H["query_pos"]).to(dt)
    outp = (out_in.float() + qpos.float()).to(dt)
    print(f"layer {i}: outp {rel(outp, dbg['outp']):.2e}")
    A = layer.attentions[0]
    PA = A.packed(dt)
    Q = out_in.shape[0]
    vt_buf = ops.zeros((E, round_up(Q, 64)), dt, out_in.device)
    vt = ops.gemm(out_in, PA["wv"], PA["bv"], trans_out=True, out=vt_buf)
    qk = ops.gemm(outp, PA["wqk"], PA["bqk"])
    print(f"  q {rel(qk[:, :E], dbg['q']):.2e}  k {rel(qk[:, E:], dbg['k']):.2e}  v {rel(vt[:, :Q].t(), dbg['v']):.2e}")
    o = ops.attention(qk[:, :E], qk[:, E:], vt, batch=1, n=Q, heads=8, head_dim=32, scale=32 ** -0.5)
    print(f"  attention out {rel(o, dbg['sa']):.2e}")
    o_t = ops.attention(D(dbg['q']).to(dt), D(dbg['k']).to(dt), ops.gemm(D(dbg['v']).to(dt), torch.eye(E, device=dev, dtype=dt), None, trans_out=True, out=ops.zeros((E, round_up(Q, 64)), dt, out_in.device)),
                        batch=1, n=Q, heads=8, head_dim=32, scale=32 ** -0.5)
    print(f"  attention on the oracle's q, k, v {rel(o_t, dbg['sa']):.2e}")
    x1 = ops.gemm(D(dbg['sa']).to(dt), PA["wo"], PA["bo"], residual=out_in, out_dtype=dt)
    print(f"  x1 (out proj + residual, oracle's attention out) {rel(x1, dbg['x1']):.2e}")
    x2, x2p = ops.layernorm(D(dbg['x1']).to(dt), *layer.norm_params(0), out_dtype=dt, add=qpos)
    print(f"  x2 {rel(x2, dbg['x2']):.2e}  x2p {rel(x2p, dbg['x2p']):.2e}")
    print(f"  value (layer slice) {rel(value_all[:, i * E:(i + 1) * E], dbg['val']):.2e}")
    ref_in = D(dbg['ref_in']).contiguous()
    x3 = layer.attentions[1].forward_tokens(D(dbg['x2p']).to(dt), D(dbg['x2']).to(dt), ref_in, geo.shapes, geo.starts, dt, value=value_all[:, i * E:(i + 1) * E])
    print(f"  x3 (cross attention + identity) {rel(x3, dbg['x3']):.2e}")
    x4 = ops.layernorm(D(dbg['x3']).to(dt), *layer.norm_params(1), out_dtype=dt)
    print(f"  x4 {rel(x4, dbg['x4']):.2e}")
    x5 = layer.ffns[0].forward_tokens(D(dbg['x4']).to(dt), dt)
    o2, _ = ops.layernorm(x5, *layer.norm_params(2), out_dtype=dt, add=qpos)
    print(f"  dec{i}_out from the oracle's x4 {rel(o2, H[f'dec{i}_out']):.2e}")
