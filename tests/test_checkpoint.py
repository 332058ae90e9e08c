"""DetectionCheckpointer (ape/checkpoint/detection_checkpoint.py:15-48): a checkpoint in the public format -- {"model": state
dict with the reference's parameter names (numpy arrays allowed), plus trainer entries} -- loads into the HIP-backed model and
reproduces it; unsupported entries are dropped with a warning, not raised."""
import logging
import os

import numpy as np
import torch


def test_checkpoint_round_trip_with_reference_key_names(tmp_path, caplog):
    from ape_amd.checkpoint import DetectionCheckpointer
    from ape_amd.modeling.build import build_ape, init_synthetic
    from oracle import weights
    import oracle_util as U

    src = init_synthetic(build_ape("tiny"), seed=5)
    spec = dict(U.load_spec("tiny"))                                        # the reference model's own state_dict() names
    sd = src.state_dict()
    assert set(sd) == set(spec)
    payload = {k: (v.numpy() if i % 3 == 0 else v) for i, (k, v) in enumerate(sd.items())}     # detectron2 checkpoints mix both
    payload = {"module." + k if k.startswith("model_vision.neck") else k: v for k, v in payload.items()}
    payload["model_vision.not_a_tensor"] = "left over from a converter"
    path = os.path.join(tmp_path, "model_final.pth")
    torch.save({"model": payload, "iteration": 1080000, "trainer": {"lr": 0.0}}, path)

    dst = build_ape("tiny")
    with caplog.at_level(logging.WARNING):
        ck = DetectionCheckpointer(dst).load(path)
    assert ck["iteration"] == 1080000
    assert any("Unsupported type" in r.message for r in caplog.records)
    inc = ck.get("__incompatible__")
    if inc is not None:                                                      # the local loader reports what it could not place
        assert not inc.missing_keys and not inc.unexpected_keys
    for k, v in dst.state_dict().items():
        assert torch.equal(v, sd[k]), k
    # save / resume bookkeeping
    saver = DetectionCheckpointer(dst, save_dir=str(tmp_path))
    saver.save("model_0000001", iteration=1)
    assert saver.has_checkpoint() and saver.get_checkpoint_file().endswith("model_0000001.pth")
    again = build_ape("tiny")
    DetectionCheckpointer(again).load(saver.get_checkpoint_file())
    assert all(torch.equal(v, sd[k]) for k, v in again.state_dict().items())
    _ = weights, np


def test_alias_path_of_the_checkpointer():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"import sys; sys.path.insert(0, {root!r}); from ape.checkpoint import DetectionCheckpointer as D; print(D.__module__)"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "ape_amd.checkpoint", out.stderr[-800:]
