"""Overlay of the alias package on a reference checkout.

The reference's LazyConfigs import training-side names next to the hot-path classes (`from ape.modeling.ape_deta import
DeformableCriterion, Stage1Assigner, ...`, `from ape.data.detection_utils import ...`: ape_deta_r50.py:9-20, the L_D config :9-17),
and `demo_lazy.py` / `train_net.py` import `ape.data`, `ape.evaluation`, `ape.engine.train_loop`.  None of that is on the inference
hot path and none of it is reimplemented here.  With a reference checkout reachable -- `$APE_REFERENCE`, or an `ape` package further
down `sys.path` -- every alias package appends the matching reference directory to its `__path__`, so sub-modules this
repository does not provide (`ape.data`, `ape.evaluation`, `ape.modeling.ape_deta.deformable_criterion`, ...) resolve to the
reference's files, and the names the reference's `__init__` files re-export from them are resolved lazily (PEP 562) on first use.
The hot-path modules keep resolving to the HIP-backed classes because the alias directory comes first."""
import importlib
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_CACHE = []


def reference_ape_dir():
    if _CACHE:
        return _CACHE[0]
    cands = []
    if os.environ.get("APE_REFERENCE"):
        cands.append(os.path.join(os.environ["APE_REFERENCE"], "ape"))
    for p in sys.path:
        d = os.path.abspath(os.path.join(p or ".", "ape"))
        if d != _HERE:
            cands.append(d)
    found = None
    for d in cands:
        if os.path.isfile(os.path.join(d, "modeling", "ape_deta", "deformable_criterion.py")):
            found = os.path.abspath(d)
            break
    _CACHE.append(found)
    return found


def extend(path_list, *rel):
    """append <reference>/ape/<rel...> to a package's __path__"""
    ref = reference_ape_dir()
    if ref:
        d = os.path.join(ref, *rel)
        if os.path.isdir(d) and d not in path_list:
            path_list.append(d)


def lazy(pkg_globals, mapping):
    """module-level __getattr__ that imports `name` from the reference sub-module mapping[name] on first access"""
    def __getattr__(name):
        if name in mapping:
            if reference_ape_dir() is None:
                raise AttributeError(f"{pkg_globals['__name__']}.{name} lives in the reference checkout (not on the inference hot path): "
                                     "put it on sys.path behind this repository or set APE_REFERENCE")
            mod = importlib.import_module(mapping[name], pkg_globals["__name__"])
            val = getattr(mod, name)
            pkg_globals[name] = val
            return val
        raise AttributeError(f"module {pkg_globals['__name__']!r} has no attribute {name!r}")
    return __getattr__
