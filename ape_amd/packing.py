"""Weight repacking for the HIP kernels (done once per module and compute dtype, after load_state_dict).

Layouts: every contraction weight is kept as [N, K] row-major (nn.Linear layout) in the compute dtype, with K
zero-padded to a multiple of 8 (16-byte rows) for the bf16 MFMA kernel; biases / norm parameters stay fp32.
"""
import torch


def round_up(x, m):
    return (x + m - 1) // m * m


def pack_matrix(w, dtype, kpad=8, rows=None):
    """w [N, K] -> contiguous [rows or N, round_up(K, kpad)] in `dtype`, zero padded"""
    N, K = w.shape
    Kp = round_up(K, kpad)
    Np = rows or N
    out = torch.zeros((Np, Kp), dtype=dtype, device=w.device)
    out[:N, :K] = w.detach().to(dtype)
    return out


def ffn_w2_perm(hid):
    """column order of the pre-permuted W2 of the fused FFN kernel (csrc/ffn_fused.hip, W2P): inside every group of 32 hidden
    units position 8 g + e holds hidden 4 g + e (e < 4) / 16 + 4 g + (e - 4) (e >= 4) -- the k order in which the first MFMA's
    accumulator registers are the second MFMA's B operand"""
    assert hid % 32 == 0
    pos = torch.arange(hid)
    grp, r = pos // 32, pos % 32
    g, e = r // 8, r % 8
    return grp * 32 + torch.where(e < 4, 4 * g + e, 16 + 4 * g + (e - 4))


def permute_ffn_w2(w2):
    """w2 [N, HID] -> the same matrix with its hidden columns in the fused FFN kernel's order"""
    return w2[:, ffn_w2_perm(w2.shape[1]).to(w2.device)].contiguous()


def f32(t):
    return None if t is None else t.detach().float().contiguous()


class PackCache:
    """per-module cache of packed tensors, keyed by (compute dtype, device); dropped when weights change"""

    def __init__(self):
        self._store = {}

    def get(self, module, dtype, builder):
        p = next(module.parameters(), None)
        key = (dtype, p.device if p is not None else None)
        if key not in self._store:
            with torch.no_grad():
                self._store[key] = builder(dtype)
        return self._store[key]

    def clear(self):
        self._store.clear()


def attach_cache(module):
    module._pack = PackCache()
    module.register_load_state_dict_post_hook(lambda m, incompatible: m._pack.clear())


def _cubic_axis_weights(n_in, n_out):
    """[n_out, n_in] float64: one axis of torch's bicubic resampling (`F.interpolate(mode="bicubic", align_corners=False)`: cubic
    convolution with A = -0.75 around src = (n_in / n_out) (dst + 0.5) - 0.5, the four taps' indices clamped to the border) written as a
    matrix -- the resampling is linear in the input"""
    import math

    import numpy as np
    A = -0.75

    def near(x):          # |x| <= 1
        return ((A + 2.0) * x - (A + 3.0)) * x * x + 1.0

    def far(x):           # 1 < |x| < 2
        return ((A * x - 5.0 * A) * x + 8.0 * A) * x - 4.0 * A
    W = np.zeros((n_out, n_in), dtype=np.float64)
    scale = n_in / n_out
    for o in range(n_out):
        src = scale * (o + 0.5) - 0.5
        i0 = math.floor(src)
        t = src - i0
        for k, wk in enumerate((far(t + 1.0), near(t), near(1.0 - t), far(2.0 - t))):
            W[o, min(max(i0 - 1 + k, 0), n_in - 1)] += wk
    return W


def resize_pos_embed(pos, size, hw):
    """get_abs_pos' resize (ape/modeling/backbone/utils_eva02.py:158-187: F.interpolate(bicubic, align_corners=False) of the pretraining
    grid's position embedding to the token grid) as ONE GEMM of the library: bicubic resampling is a separable linear map, so
    out[(y, x), :] = sum_ij Wy[y, i] Wx[x, j] pos[(i, j), :] = (Wy (x) Wx) . pos with the Kronecker matrix built on the host in float64
    (4096 x 576 for APE-L_D: 9 MB).  pos [size * size, C] fp32 on the model's device -> [hw * hw, C] fp32.
    NOT used by the backbones: it agrees with torch's kernel to 1e-6 (tests/test_host_model.py), and the 1e-6 is what rules it out -- the
    reference fixtures were produced with torch's bicubic arithmetic, and at 1536^2 the fp32 pipeline sits just inside its 1e-3 bar
    against them (logits 9.6e-4, boxes 5.2e-4); with this embedding it measured logits 5.5e-4, boxes 1.04e-3 (round 5,
    profiles/r05_pos_embed_gemm_experiment.log): the random-weight model turns 1e-6 into 1e-3 either way, and only one of the two is
    the reference's arithmetic.  The
    backbones keep F.interpolate at weight-packing time (a per-model constant, off the forward path)."""
    import numpy as np

    from . import ops
    w1 = _cubic_axis_weights(size, hw)
    kron = torch.from_numpy(np.kron(w1, w1).astype(np.float32)).to(pos.device)          # square grids: the same matrix on both axes
    return ops.gemm(kron, pos.detach().float().t().contiguous(), None, out_dtype=torch.float32)

