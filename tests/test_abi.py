"""The C-ABI shared library loads on a CPU-only host and exports every symbol include/ape_hip.h declares (no compute)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "ape_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(ape_hip_\w+)\s*\(", text))


def test_every_declared_symbol_is_exported_and_bound():
    from ape_amd import _lib, build

    build.build(verbose=False)
    lib = _lib.load()
    declared = _header_symbols()
    assert declared, "no declarations found"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.ape_hip_abi_version() == 5


def test_argument_struct_layouts_match_ctypes():
    from ape_amd import _lib

    lib = _lib.load()
    assert lib.ape_hip_sizeof_args(0) == ctypes.sizeof(_lib.GemmArgs)
    assert lib.ape_hip_sizeof_args(1) == ctypes.sizeof(_lib.LayerNormArgs)
    assert lib.ape_hip_sizeof_args(2) == ctypes.sizeof(_lib.GroupNormArgs)


def test_argument_errors_are_reported_without_a_gpu():
    from ape_amd import _lib

    lib = _lib.load()
    rc = lib.ape_hip_gemm(None, None)
    assert rc != 0 and b"null" in lib.ape_hip_last_error()


def test_oracle_is_not_imported_by_the_product():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/"""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ape_amd")):
        for f in files:
            if f.endswith(".py") and re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(dirpath, f)).read(), flags=re.M):
                bad.append(os.path.join(dirpath, f))
    assert not bad, bad
