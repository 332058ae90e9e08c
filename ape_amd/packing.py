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
