"""bench.py's launch contract, on the CPU: `python bench.py --gpus N` WITHOUT a launcher in the environment re-executes itself
under torch.distributed.run with N ranks (a driver that calls `python bench.py --gpus 8` must get 8 ranks, not a silent 1-GPU
run); `--dry --backend gloo` runs the rendezvous and the data-parallel exchanges (text-bank broadcast, per-step all-gather of
the detection records) with no device."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "gloo", "--dry", "--steps", "3", "--warmup", "0", *extra],
                       capture_output=True, text=True, timeout=600, env=e)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p, lines


def test_gpus_2_without_a_launcher_spawns_two_ranks():
    p, lines = _run("--gpus", "2")
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, p.stdout                          # rank 0 prints ONE JSON line
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["rccl_ranks"] == 2 and r["config"]["parallelism"] == "dp2"
    assert r["config"]["gathered_shape"][0] == 2 and r["config"]["gathered_rank_ids"] == [0, 1]
    assert r["config"]["text_bank_identical_on_all_ranks"] is True
    assert r["config"]["shard_of_rank0"] == [0, 499]          # contiguous shards of the 1000-image stream
    assert r["value"] is None and r["data"].startswith("dry-run")


def test_gpus_1_runs_in_process_and_a_mismatched_world_size_is_refused():
    p, lines = _run("--gpus", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 1 and r["config"]["rccl_ranks"] == 1
    p, lines = _run("--gpus", "4", env={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29591"})
    assert p.returncode != 0 and "WORLD_SIZE=2" in (p.stderr + p.stdout)


def test_uneven_shards_of_a_coco_shaped_stream_run_the_same_number_of_steps():
    """7 images over 2 ranks: contiguous blocks of 4 and 3 -- the shorter block starts one image early (dp.shard_indices), so both
    ranks run the same number of steps and every lagged all-gather finds its partner; every step's record set is collected"""
    p, lines = _run("--gpus", "2", "--stream", "coco", "--dry-images", "7", "--images-per-step", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads(lines[0])
    c = r["config"]
    assert c["shard_sizes"] == [4, 4] and c["shard_of_rank0"] == [0, 3]
    assert c["steps_run"] == 4 and c["record_sets_collected"] == 4 and c["gathered_rank_ids"] == [0, 1]


def test_gpus_8_shards_the_1000_image_stream_like_the_reference_sampler_and_names_the_same_workload_as_n1():
    """`--gpus 8 --dry --stream coco`: eight ranks start, the 1000-image COCO-shaped stream is cut into the reference's InferenceSampler
    blocks (ape/data/samplers/distributed_sampler_multi_dataset.py:160-170: contiguous, total // world each, the first total % world
    ranks one more), every rank runs the same number of steps -- and the line's workload string is the one the N = 1 line carries
    (the driver computes 1 -> N efficiency from lines it must be able to tell are the same workload)"""
    p8, lines8 = _run("--gpus", "8", "--stream", "coco")
    assert p8.returncode == 0, p8.stderr[-2000:]
    r8 = json.loads(lines8[0])
    c8 = r8["config"]
    assert r8["n_gpus"] == 8 and c8["rccl_ranks"] == 8 and c8["parallelism"] == "dp8"
    # the reference's rule, restated: shard_size = total // world, left = total % world, rank r gets shard_size + (r < left) items from
    # the running sum on
    total, world = 1000, 8
    size, left = total // world, total % world
    sizes = [size + int(r < left) for r in range(world)]
    want = [[sum(sizes[:r]), sum(sizes[:r + 1]) - 1, sizes[r]] for r in range(world)]
    assert c8["shards"] == want and c8["shard_sizes"] == sizes
    assert c8["steps_run"] == (sizes[0] + 1) // 2 and c8["record_sets_collected"] == c8["steps_run"]      # 2 images per step
    assert c8["gathered_rank_ids"] == list(range(8)) and c8["text_bank_identical_on_all_ranks"] is True
    p1, lines1 = _run("--gpus", "1", "--stream", "coco")
    assert p1.returncode == 0, p1.stderr[-2000:]
    c1 = json.loads(lines1[0])["config"]
    assert c1["workload"] == c8["workload"] and "1024x1024" in c1["workload"] and "rank" in c1["workload"]
    assert c1["shards"] == [[0, 999, 1000]]


def test_mask_format_default_is_the_same_contract_for_every_rank_count():
    """the timed step of an N-GPU run is the 1-GPU step plus the exchange: `auto` resolves to bitmasks on the host for N = 1 and to
    bitmasks on the host + all-gathered run lengths ("both") for N > 1 -- never to run lengths INSTEAD of bitmasks (round 4's bench.py:469)"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '("both" if world > 1 else "bitmask")' in src
    assert '"rle" if world > 1' not in src


def test_roofline_traffic_is_refused_unless_the_pmc_summary_was_measured_on_this_build(tmp_path, monkeypatch):
    """bench.pmc_traffic_bytes: the newest profiles/*_pmc_summary.txt counts only when its `# library_digest` header equals the digest
    of the sources bench.py runs on (ape_amd.build._digest()); a summary of another build -> (None, why).  The family's instantiations
    (dense, persistent, convolution) are launch-weighted."""
    sys.path.insert(0, ROOT)
    import bench
    from ape_amd import build
    prof = tmp_path / "profiles"
    prof.mkdir()
    rows = ("kernel | launches | FETCH_SIZE | WRITE_SIZE | HBM MB/launch\n"
            "gemm_bf16_p8_kernel<256, true, unsigned short, 0, false, tru |   100 | 1 | 1 | 200.0\n"
            "gemm_bf16_p8_kernel<256, true, unsigned short, 0, false, fal |   300 | 1 | 1 | 100.0\n"
            "gemm_bf16_p8_kernel<128, true, unsigned short, 0, false, fal |   500 | 1 | 1 | 50.0\n")
    (prof / "r09_pmc_summary.txt").write_text("# library_digest deadbeef\n" + rows)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    t, why = bench.pmc_traffic_bytes("gemm_bf16_p8_kernel<256, true>")
    assert t is None and "refused" in why and "deadbeef" in why
    (prof / "r09_pmc_summary.txt").write_text(f"# library_digest {build._digest()}\n" + rows)
    t, why = bench.pmc_traffic_bytes("gemm_bf16_p8_kernel<256, true>")
    assert abs(t - 125.0 * 1024 * 1024) < 1.0 and "400 launches" in why          # (100 x 200 + 300 x 100) / 400 MB
    t, _ = bench.pmc_traffic_bytes("gemm_bf16_p8_kernel<128, true>")
    assert abs(t - 50.0 * 1024 * 1024) < 1.0


def test_kernel_family_folds_the_tile_kernels_instantiations():
    sys.path.insert(0, ROOT)
    import bench
    fam = bench.GemmMeter.family
    assert fam("gemm_bf16_p8_kernel<256, true, conv3x3>") == fam("gemm_bf16_p8_kernel<256, true, persistent>") == "gemm_bf16_p8_kernel<256, true>"
    assert fam("gemm_bf16_p8_kernel<256, true>") == "gemm_bf16_p8_kernel<256, true>" and fam("gemm_f16_p8_kernel<128, true, conv3x3>") == "gemm_f16_p8_kernel<128, true>"
    assert fam("gemm_bf16_kres_kernel<0, false, true>") == "gemm_bf16_kres_kernel<0, false, true>"
    # round 6: the in-launch-statistics flavour of the 256 x 128 tile kernel belongs to ITS family, not to the 256 x 256 one and not to none
    assert fam("gemm_bf16_p8_kernel<128, true, rowstat>") == "gemm_bf16_p8_kernel<128, true>" != fam("gemm_bf16_p8_kernel<256, true, persistent>")
