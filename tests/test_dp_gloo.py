"""world_size-2 gloo test (CPU) of the data-parallel runner: contiguous image shards (the reference's InferenceSampler), text-bank broadcast from rank 0,
all-gather of fixed-size detection records.  The model runs on the torch definitions of the ops (fake backend)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, ret):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    import ape_amd.ops as ops
    import ref_ops
    for n in dir(ref_ops):
        if not n.startswith("_") and callable(getattr(ref_ops, n)) and hasattr(ops, n):
            setattr(ops, n, getattr(ref_ops, n))
    from ape_amd.dp import DataParallelRunner, shard_indices
    from ape_amd.modeling.build import build_ape, init_synthetic

    model = init_synthetic(build_ape("tiny"), seed=0)
    mv = model.model_vision
    mv.set_compute_dtype(torch.float32)

    def fwd(image, text):
        out = mv.forward_single(image, text, with_masks=False)
        rec = torch.cat([out["det_boxes"], out["det_scores"][:, None], out["det_classes"][:, None].float()], 1)
        return None, rec

    runner = DataParallelRunner(fwd, mv.test_topk_per_image, torch.device("cpu"))
    bank = torch.randn(6, 1024, generator=torch.Generator().manual_seed(3)) if rank == 0 else None
    text = runner.broadcast_text_bank(bank, 6, 1024)
    images = [torch.randint(0, 256, (3, 256, 256), generator=torch.Generator().manual_seed(10 + i)).float() for i in range(4)]
    mine = shard_indices(len(images), rank, world)
    gathered = []
    for i in mine:
        _, allrec = runner.step(images[i], text)
        gathered.append(allrec)

    # the pipelined form (submit image i, collect image i-1) must deliver the same records in the same order
    class Pipelined:
        def submit(self, image, text, height=None, width=None, prompt="name"):
            from types import SimpleNamespace
            rec = fwd(image, text)[1]
            # stand-in for the runtime's device RLE buffers (mask_format="rle"): [k, cap] run lengths + [k] run counts
            runs = (rec[:, :4].round().to(torch.int32).repeat(1, 2).contiguous(), rec[:, 5].to(torch.int32).contiguous())
            return SimpleNamespace(rec6=rec, runs=runs)

        def result(self, ticket):
            return None, ticket.rec6

    prunner = DataParallelRunner(Pipelined(), mv.test_topk_per_image, torch.device("cpu"), gather_masks=True)
    piped, pending, tickets = [], None, []
    for i in mine:
        t = prunner.submit(images[i], text)
        tickets.append(t)
        if pending is not None:
            piped.append(prunner.result(pending)[1])
        pending = t
    piped.append(prunner.result(pending)[1])
    same = all(torch.equal(a, b) for a, b in zip(piped, gathered)) and len(piped) == len(gathered)
    # the masks' run lengths of every rank arrive with the records (rank r's entry == what rank r produced for that step)
    for t, allrec in zip(tickets, gathered):
        counts, nruns = t.mask_runs
        same &= counts.shape[0] == world and torch.equal(counts[rank], t.runs[0]) and torch.equal(nruns[rank], t.runs[1])
        for r in range(world):
            same &= torch.equal(counts[r][:, :4], allrec[r][:, :4].round().to(torch.int32))

    # lagged exchange (lag = 1, what bench.py --gpus N > 1 runs): result(ticket_i) hands back the gathered records of ticket i-1,
    # drain() the last; the same records in the same order, each collective awaited one step late and only by the host
    lrunner = DataParallelRunner(Pipelined(), mv.test_topk_per_image, torch.device("cpu"), gather_masks=True, lag=1)
    lagged, ltickets = [], []
    for i in mine:
        t = lrunner.submit(images[i], text)
        ltickets.append(t)
        g = lrunner.result(t)[1]
        if g is not None:
            lagged.append(g)
    same &= len(lagged) == len(mine) - 1
    lagged.append(lrunner.drain())
    same &= lrunner.drain() is None and len(lagged) == len(gathered) and all(torch.equal(a, b) for a, b in zip(lagged, gathered))
    for t in ltickets:
        counts, nruns = t.mask_runs
        same &= counts.shape[0] == world and torch.equal(counts[rank], t.runs[0]) and torch.equal(nruns[rank], t.runs[1])

    # text bank from class names: only rank 0 owns a text tower
    class Tower:
        calls = 0

        def forward_text(self, names, cache=False):
            Tower.calls += 1
            g = torch.Generator().manual_seed(len(names))
            return {"last_hidden_state_eot": torch.randn(len(names), 1024, generator=g)}

    # the evaluators' end-of-run exchange: per-rank Python lists of {"image_id", "instances": [json dicts]} gathered to rank 0
    from ape_amd.dp import chain_predictions, gather_predictions
    preds = [{"image_id": i, "instances": [{"image_id": i, "category_id": int(r[5]), "bbox": r[:4].tolist(), "score": float(r[4])}
                                            for r in gathered[s][rank][:2]]} for s, i in enumerate(mine)]
    got = gather_predictions(preds, dst=0)
    if rank == 0:
        flat = chain_predictions(got)
        same &= len(got) == world and [p["image_id"] for p in flat] == sorted(p["image_id"] for p in flat) == list(range(len(images)))
        same &= all(len(p["instances"]) == 2 and p["instances"][0]["image_id"] == p["image_id"] for p in flat)
    else:
        same &= got == []
    named = runner.text_bank_from_names(Tower() if rank == 0 else None, ["cat", "dog", "traffic light"])
    same &= tuple(named.shape) == (3, 1024) and (Tower.calls == (1 if rank == 0 else 0))
    ret[f"named_sum_{rank}"] = float(named.sum())
    ret[f"same_{rank}"] = bool(same)
    if rank == 0:
        # single-process ground truth for every image
        want = [fwd(img, text)[1] for img in images]
        ok = True
        for step, allrec in enumerate(gathered):
            for r in range(world):
                ok &= torch.allclose(allrec[r], want[shard_indices(len(images), r, world)[step]], atol=1e-5)
        ret["ok"] = bool(ok) and same
        ret["text_sum"] = float(text.sum())
    else:
        ret[f"text_sum_{rank}"] = float(text.sum())
    dist.barrier()
    dist.destroy_process_group()


def test_dp_two_ranks_gloo():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret["ok"] and ret["same_0"] and ret["same_1"]
    assert abs(ret["text_sum"] - ret["text_sum_1"]) < 1e-6   # the broadcast reached rank 1
    assert abs(ret["named_sum_0"] - ret["named_sum_1"]) < 1e-6 and ret["named_sum_0"] != 0.0


def test_shard_indices():
    from ape_amd.dp import shard_indices
    # the reference's InferenceSampler._get_local_indices (distributed_sampler_multi_dataset.py:160-170), restated
    def reference(total_size, world_size, rank):
        shard_size, left = total_size // world_size, total_size % world_size
        shard_sizes = [shard_size + int(r < left) for r in range(world_size)]
        begin, end = sum(shard_sizes[:rank]), min(sum(shard_sizes[: rank + 1]), total_size)
        if end - begin < max(shard_sizes):
            begin = begin - 1
        return list(range(begin, end))

    assert shard_indices(10, 1, 4) == [3, 4, 5] and shard_indices(10, 3, 4) == [7, 8, 9]      # rank 3's short block starts one early
    for n, w in ((1000, 8), (10, 4), (5000, 8), (7, 2), (8, 8), (1203, 4)):
        shards = [shard_indices(n, r, w) for r in range(w)]
        assert all(s == reference(n, w, r) for r, s in enumerate(shards))
        assert sorted(set(sum(shards, []))) == list(range(n)) and len({len(s) for s in shards}) == 1
