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
