"""Which tensor-library (aten) operators does ONE per-image forward still issue outside the HIP ops?

Runs the host model on the CPU with every `ape_amd.ops` entry swapped for its torch definition (tests/ref_ops.py) under a
TorchDispatchMode; aten calls made INSIDE an op's definition are the op (a HIP kernel on the GPU) and are skipped, aten calls at
depth 0 are glue the GPU would launch as tensor-library kernels.  View / metadata operators launch nothing and are not counted.
Prints the glue launches grouped by call site (file:line of the innermost ape_amd frame), in bf16 production mode.

    python tools/glue_ops.py [case]           # default tiny_padded
"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import model_util as M  # noqa: E402
import ape_amd.ops as ops  # noqa: E402
import ref_ops  # noqa: E402

NO_LAUNCH = {"view", "_unsafe_view", "reshape", "expand", "permute", "transpose", "t", "slice", "select", "unsqueeze", "squeeze", "as_strided",
             "alias", "detach", "unbind", "split", "split_with_sizes", "chunk", "narrow", "unfold", "_reshape_alias", "empty", "empty_like",
             "empty_strided", "new_empty", "new_empty_strided", "size", "stride", "sym_size", "is_pinned", "lift_fresh", "_local_scalar_dense",
             "resize_", "set_", "result_type", "item", "is_same_size", "view_as_real", "view_as_complex", "diagonal", "numpy_T", "mT"}

depth = [0]


def wrap(fn):
    def inner(*a, **k):
        depth[0] += 1
        try:
            return fn(*a, **k)
        finally:
            depth[0] -= 1
    inner.__name__ = fn.__name__
    return inner


class Glue(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = collections.Counter()
        self.kinds = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split(".")[0]
        if depth[0] == 0 and name not in NO_LAUNCH:
            site = "?"
            for fr in reversed(traceback.extract_stack()):
                if "/ape_amd/" in fr.filename and "ops.py" not in fr.filename:
                    site = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.name}"
                    break
            self.sites[(site, name)] += 1
            self.kinds[name] += 1
        return out


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else "tiny_padded"
    for n in dir(ref_ops):
        if not n.startswith("_") and callable(getattr(ref_ops, n)) and hasattr(ops, n):
            setattr(ops, n, wrap(getattr(ref_ops, n)))
    model, image, text, gold = M.build_model(case, "cpu", torch.float32)
    mv = model.model_vision
    mv.set_compute_dtype(torch.bfloat16)
    mv.forward_single(image, text)                    # packs the weights, builds the per-size caches
    with Glue() as g:
        mv.forward_single(image, text)
    total = sum(g.kinds.values())
    print(f"[{case}] {total} tensor-library launches in one forward (bf16 mode, caches warm)")
    bysite = collections.defaultdict(list)
    for (site, name), n in g.sites.items():
        bysite[site].append((name, n))
    for site, lst in sorted(bysite.items(), key=lambda kv: -sum(n for _, n in kv[1])):
        print(f"  {sum(n for _, n in lst):4d}  {site}: " + ", ".join(f"{name} x{n}" if n > 1 else name for name, n in sorted(lst, key=lambda t: -t[1])))


if __name__ == "__main__":
    main()
