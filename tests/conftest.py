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
    config.addinivalue_line("markers", "slow_gpu: the widened model families at full size (ViT-e / EVA-01 ViT-g) and the IEEE-half "
                                       "repetitions of the full-size cases -- still part of -m gpu, ordered LAST (see below)")


# Inside `-m gpu` the cases of BASELINE's own configurations (APE-L_D / APE-Ti, bf16) run first; the widened families (APE on ViT-e,
# EVA-01 ViT-g: *_E_D_*, *_V_A_*, *_G_A_*) and the f16 repetitions of full-size cases are marked slow_gpu and moved to the END of the
# session: if the driver's wall-clock limit ever cuts the run, what it cuts is the breadth, not the hot path's own parity cases.
# Within each group the collection order is kept (it also keeps tests of one model configuration adjacent: tests/model_util.py shares
# built models between them).
_SLOW_KEYS = ("E_D_coco80", "V_A_coco80", "G_A_1536", "E_D_full_size")


def pytest_collection_modifyitems(config, items):
    fast, slow = [], []
    for it in items:
        nid = it.nodeid
        full_size_f16 = nid.endswith("-f16]") and ("test_L_D_bf16_pipeline" in nid or "test_L_D_bf16_teacher_forced" in nid)
        # the same-rounding (T2) cases whose oracle forward takes minutes of host time: 1536^2, and the f16 repetitions
        t2_long = "test_hip_pipeline_vs_same_rounding_oracle" in nid and ("L_D_1536" in nid or nid.endswith("-f16]"))
        if any(k in nid for k in _SLOW_KEYS) or full_size_f16 or t2_long:
            it.add_marker(pytest.mark.slow_gpu)
            slow.append(it)
        else:
            fast.append(it)
    items[:] = fast + slow


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
