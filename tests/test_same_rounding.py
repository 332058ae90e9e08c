"""T2 of BASELINE.md section 3: the 16-bit HIP pipeline (what bench.py times), stage by stage, against the oracle evaluated AT THE SAME
ROUNDING POINTS (oracle/rounded.py).  Every stage of the pipeline is teacher-forced with the rounded oracle's input (ape_amd/stagetap.py) and
its output compared with the rounded oracle's output: what is left is accumulation order, fp32 transcendental accuracy and the rare 16-bit
rounding flip those cause -- not a rounding point the pipeline has and the reference arithmetic has not.

CPU: the harness on the tiny model with the torch definitions of the ops (tests/ref_ops.py) -- pins the ROUNDING STRUCTURE of the host code.
GPU (-m gpu): the HIP kernels at BASELINE's configurations (APE-L_D 1024^2 square / padded, 1536^2 + semantic branch), both 16-bit flavours,
every stage <= TOL relative rms; exceptions are listed in KNOWN with their cause.  (The 1536^2 case and the f16 repetitions beyond L_D_coco80
are opt-in -- APE_TEST_SLOW=1 / APE_TEST_ALL_F16=1 -- because the oracle runs on the host and the driver's GPU suite has 1200 s.)
"""
import os
import time

import pytest
import torch

import model_util as M
import oracle_util as U
from ape_amd.stagetap import StageTap
from oracle import rounded, weights
from oracle.configs import CONFIGS

TOL = 2e-3
# (case, dtype tag) -> {stage: (allowed relative rms, cause)}
KNOWN = {}


def run_case(case, dev, dt, semantic=True):
    """-> (distances {stage: (rms, max, kind)}, extras)"""
    model, image, text, gold = M.build_model(case, dev, torch.float32)
    cfg_name, wseed = gold["case"][0], gold["case"][1]
    mv = model.model_vision
    sem = meta = None
    if semantic and "semantic_meta" in gold:
        meta = gold["semantic_meta"]
        mv.semantic_on = True
        mv.set_metadata(0, name="coco_2017_val", thing_classes=meta["thing_classes"], stuff_classes=meta["stuff_classes"])
        sem = dict(mv.metadata_list[-1], entity=mv.dataset_entities[-1])
    sd = weights.make_state_dict(U.load_spec(cfg_name), wseed)
    orc = rounded.RoundedApeOracle(CONFIGS[cfg_name], sd, dtype=dt)
    t0 = time.perf_counter()
    kw = dict(semantic=meta) if meta is not None else {}
    if "out_hw" in gold:
        kw.update(height=gold["out_hw"][0], width=gold["out_hw"][1])
    orc.forward(image, text, **kw)
    t_or = time.perf_counter() - t0
    del sd
    net = mv.backbone.net
    hw = net.img_size // net.patch_size
    teacher = rounded.teacher_stages(orc, net.token_order(hw)[0])
    mv.set_compute_dtype(dt)
    forced = StageTap(teacher=teacher)
    out = mv.forward_single(image.to(dev), text.to(dev), forced_topk=orc.hip["topk_proposals"].to(dev), stages=forced, semantic=sem)
    dist = rounded.stage_distances(forced, teacher)
    extras = {"oracle_seconds": t_or}
    if sem is not None:
        a, b = out["sem_seg"].float().cpu(), orc.hip["sem_seg"]
        extras["sem_seg_rms"] = float(((a - b).double().pow(2).sum().sqrt() / b.double().pow(2).sum().sqrt()))
        extras["sem_label_agreement"] = float((a.argmax(0) == b.argmax(0)).float().mean())
    return dist, extras


def report(tag, dist, extras):
    order = sorted(dist, key=lambda k: -dist[k][0])
    print(f"[same-rounding {tag}] {len(dist)} stages; oracle forward {extras['oracle_seconds']:.1f} s; worst: "
          + ", ".join(f"{k} {dist[k][0]:.2e}" for k in order[:6]), flush=True)
    for k in sorted(dist):
        print(f"[same-rounding {tag}]   {k:18s} rms {dist[k][0]:.2e}  max {dist[k][1]:.2e}  ({dist[k][2]})")
    for k, v in extras.items():
        if k != "oracle_seconds":
            print(f"[same-rounding {tag}]   {k}: {v:.3e}")


def check(case, tag, dist, extras, min_stages):
    assert len(dist) >= min_stages, sorted(dist)
    known = KNOWN.get((case, tag), {})
    bad = {k: v[0] for k, v in dist.items() if v[0] > known.get(k, (TOL, ""))[0]}
    assert not bad, f"{case} {tag}: stages further than {TOL} from the same-rounding oracle: {bad}"
    if "sem_seg_rms" in extras:
        assert extras["sem_seg_rms"] <= TOL and extras["sem_label_agreement"] > 0.999, extras


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("case", ["tiny_padded", "tiny_semantic"])
def test_host_pipeline_has_the_rounded_oracles_rounding_points(fake_ops, case, dt):
    """CPU: the host composition over the torch definitions of the ops, teacher-forced with the rounded oracle's stages"""
    tag = "bf16" if dt == torch.bfloat16 else "f16"
    dist, extras = run_case(case, "cpu", dt)
    report(f"{case} {tag} (torch definitions)", dist, extras)
    check(case, tag, dist, extras, 30)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("case", ["L_D_coco80", "L_D_padded", "L_D_1536_sseg"])
def test_hip_pipeline_vs_same_rounding_oracle(case, dt):
    """GPU: every stage of the 16-bit HIP pipeline <= 2e-3 relative rms from the oracle at the same rounding points"""
    tag = "bf16" if dt == torch.bfloat16 else "f16"
    # suite time: the driver gives `pytest -m gpu` 1200 s on the GPU box.  The rounded oracle's forward runs on the HOST cores: 55-65 s at
    # 1024^2, 330 s at 1536^2 + semantic branch (390 s for that one case; the whole suite took 1295 s with it, profiles/r06_gpu_suite_all_cases.log).
    # Default run: both flavours at L_D_coco80, bf16 (the timed flavour) at L_D_padded; APE_TEST_SLOW=1 adds 1536^2 + semantic (bf16),
    # APE_TEST_ALL_F16=1 the f16 repetitions.  Last full run of every case: profiles/r06_same_rounding_gpu.log.
    if case == "L_D_1536_sseg" and os.environ.get("APE_TEST_SLOW") != "1":
        pytest.skip("1536^2 + semantic under APE_TEST_SLOW=1 (the oracle forward alone takes 330 s of host time; the suite has 1200 s)")
    if case != "L_D_coco80" and dt == torch.float16 and os.environ.get("APE_TEST_ALL_F16") != "1":
        pytest.skip("f16 at this configuration under APE_TEST_ALL_F16=1 (suite time)")
    dist, extras = run_case(case, "cuda", dt)
    report(f"{case} {tag}", dist, extras)
    check(case, tag, dist, extras, 60)
    M.check_pins(f"same_rounding/{case}/{tag}", {k: v[0] for k, v in dist.items()})
