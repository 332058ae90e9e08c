import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture
def fake_ops(monkeypatch):
    """Swap the HIP ops for their torch definitions (tests/ref_ops.py) so host-side logic runs on CPU."""
    import ape_amd.ops as ops
    import ref_ops

    for name in dir(ref_ops):
        if name.startswith("_"):
            continue
        obj = getattr(ref_ops, name)
        if callable(obj) and hasattr(ops, name):
            monkeypatch.setattr(ops, name, obj)
    return ref_ops
