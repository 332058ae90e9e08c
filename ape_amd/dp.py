"""Per-image data parallelism over the GPUs of one node.

Images are independent units (the reference evaluates batch 1 per rank, each rank a contiguous block of the dataset:
InferenceSampler, ape/data/samplers/distributed_sampler_multi_dataset.py:139-170, used by ape/data/build.py:79,127), so the
forward has NO collective on its data path.  The only exchanges are
  * one broadcast of the text-embedding bank [K, 1024] from rank 0 per vocabulary (RCCL over xGMI; 160 KB for 80
    classes, 2.4 MB for LVIS-1203) -- the reference instead recomputes / caches the text tower on every rank
    (clip_wrapper_eva02.py:88-128), and
  * an all-gather of fixed-size detection records [k, 6] = (x1, y1, x2, y2, score, class) per step -- the reference
    gathers pickled Python lists over a gloo group at the end (lvis_evaluation.py:103-104) -- and, with
    `gather_masks=True` on a runtime in `mask_format="rle"`, of the masks as COCO run lengths ([k, cap] int32 + [k] run
    counts per image: 1.6 MB at k = 100, cap = 4096, instead of 105 MB of bitmaps).
One process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_indices(n_items, rank, world):
    """the items rank `rank` processes: CONTIGUOUS blocks like the reference's InferenceSampler._get_local_indices
    (ape/data/samplers/distributed_sampler_multi_dataset.py:160-170) -- the first n % world ranks get one item more, and a
    shorter block starts one item early (that item is evaluated twice) so that every rank runs the same number of steps and
    the per-step all-gather of the detection records never waits for a rank that has run out of images."""
    size, left = n_items // world, n_items % world
    sizes = [size + int(r < left) for r in range(world)]
    begin = sum(sizes[:rank])
    end = min(sum(sizes[:rank + 1]), n_items)
    if end - begin < max(sizes) and begin > 0:
        begin -= 1
    return list(range(begin, end))


class _Gathered:
    """all ranks' tensors of one step, gathered asynchronously: `.get()` waits for THIS collective only (on the exchange stream,
    never on the compute stream) and returns the stacked [world, ...] tensor"""

    def __init__(self, outs, work, stream):
        self.outs, self.work, self.stream = outs, work, stream

    def get(self):
        if self.work is not None:
            if self.stream is not None:
                with torch.cuda.stream(self.stream):
                    self.work.wait()
                self.stream.synchronize()
            else:
                self.work.wait()
            self.work = None
        return torch.stack(self.outs)


class DataParallelRunner:
    def __init__(self, forward_fn, records_per_image, device, group=None, gather_masks=False, lag=0):
        """forward_fn(image, text) -> (host instances, device record tensor [records_per_image, 6]).
        lag = 1: result(ticket_i) returns the gathered records of ticket i-1 (None for the first; `drain()` returns the last):
        the all-gather of step i is issued asynchronously, its completion is awaited one step later and only by the host -- the
        compute stream never waits for a collective, so a slow rank delays the others' BOOK-KEEPING by a step, not their GPUs.
        lag = 0: the records of the ticket itself (the collective completes before result() returns them)."""
        self.forward_fn = forward_fn
        self.gather_masks = gather_masks
        self.lag = int(lag)
        self._late = None                 # (ticket, _Gathered records, _Gathered runs | None) awaiting collection (lag = 1)
        self._xstream = None
        self._gather_runs = None
        self.k = records_per_image
        self.device = device
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._gather = None

    def broadcast_text_bank(self, text, k, dim):
        """rank 0 owns the CLIP text features; everyone else receives them"""
        if self.rank != 0 or text is None:
            text = torch.empty((k, dim), dtype=torch.float32, device=self.device)
        text = text.to(self.device).float().contiguous()
        if self.world > 1:
            dist.broadcast(text, src=0, group=self.group)
        return text

    def text_bank_from_names(self, model_language, names, dim=1024, cache=True):
        """rank 0 runs the text tower on the class names (`model_language.forward_text`, e.g. ape_amd.modeling.text.EVA02CLIP)
        and every rank receives the [len(names), dim] bank -- the reference runs the tower on every rank
        (deformable_detr_segm_vl.py:258-260)"""
        feats = None
        if self.rank == 0:
            feats = model_language.forward_text(list(names), cache=cache)["last_hidden_state_eot"].float()
            if tuple(feats.shape) != (len(names), dim):
                raise ValueError(f"text tower returned {tuple(feats.shape)}, expected {(len(names), dim)}")
        return self.broadcast_text_bank(feats, len(names), dim)

    def _all_gather_runs(self, runs):
        """all ranks' (run lengths, run counts) of one step: ([world, ...], [world, ...])"""
        counts, nruns = runs
        if self.world == 1:
            return counts[None], nruns[None]
        if self._gather_runs is None or self._gather_runs[0][0].shape != counts.shape:
            self._gather_runs = ([torch.empty_like(counts) for _ in range(self.world)], [torch.empty_like(nruns) for _ in range(self.world)])
        dist.all_gather(self._gather_runs[0], counts.contiguous(), group=self.group)
        dist.all_gather(self._gather_runs[1], nruns.contiguous(), group=self.group)
        return torch.stack(self._gather_runs[0]), torch.stack(self._gather_runs[1])

    def _all_gather_async(self, t):
        """enqueue the all-gather of `t` behind the work already on the compute stream; nothing waits for it until `.get()`"""
        t = t.contiguous()
        if self.world == 1:
            return _Gathered([t], None, None)
        outs = [torch.empty_like(t) for _ in range(self.world)]       # per step: the previous step's may still be unread
        work = dist.all_gather(outs, t, group=self.group, async_op=True)
        if t.is_cuda and self._xstream is None:
            self._xstream = torch.cuda.Stream(device=t.device)
        return _Gathered(outs, work, self._xstream if t.is_cuda else None)

    def _all_gather(self, rec):
        if self.world == 1:
            return rec[None]
        if self._gather is None:
            self._gather = [torch.empty_like(rec.contiguous()) for _ in range(self.world)]
        dist.all_gather(self._gather, rec.contiguous(), group=self.group)
        return torch.stack(self._gather)

    def step(self, image, text):
        """one image on this rank; returns (instances, all ranks' records [world, k, 6])"""
        inst, rec = self.forward_fn(image, text)
        return inst, self._all_gather(rec)

    # pipelined form (forward_fn must offer submit/result, e.g. runtime.GraphedForward): the device->host transfer of
    # image i and the record all-gather overlap the compute of image i+1
    def submit(self, image, text, height=None, width=None, prompt="name"):
        """enqueue one image (output frame height x width, prompt mode as in GraphedForward.submit); returns a ticket"""
        ticket = self.forward_fn.submit(image, text, height, width, prompt)
        if getattr(ticket, "rec6", None) is not None:      # detections already enqueued (non-pipelined runtime): gather now,
            ticket.records = self._all_gather(ticket.rec6)  # stream-ordered behind the forward, no host wait
            if self.gather_masks and getattr(ticket, "runs", None) is not None:
                ticket.mask_runs = self._all_gather_runs(ticket.runs)
        return ticket

    def result(self, ticket):
        """(instances on the host, all ranks' records [world, k, 6]) of a submitted image.  With the software-pipelined runtime
        a ticket's detections are produced by the NEXT step's replay, so the record all-gather is enqueued here (every rank
        collects its tickets in the same order, which keeps the collective matched)."""
        inst, rec = self.forward_fn.result(ticket)
        if self.lag:
            runs = None
            if self.gather_masks and getattr(ticket, "runs", None) is not None:
                runs = (self._all_gather_async(ticket.runs[0]), self._all_gather_async(ticket.runs[1]))
            late, self._late = self._late, (ticket, self._all_gather_async(rec), runs)
            return inst, self._collect(late)
        if getattr(ticket, "records", None) is None:
            ticket.records = self._all_gather(rec)
        if self.gather_masks and getattr(ticket, "mask_runs", None) is None and getattr(ticket, "runs", None) is not None:
            ticket.mask_runs = self._all_gather_runs(ticket.runs)       # every rank's masks as run lengths: ticket.mask_runs
        return inst, ticket.records

    def _collect(self, late):
        if late is None:
            return None
        ticket, rec, runs = late
        ticket.records = rec.get()
        if runs is not None:
            ticket.mask_runs = (runs[0].get(), runs[1].get())
        return ticket.records

    def drain(self):
        """lag = 1: the gathered records of the last collected ticket (whose exchange nobody has waited for yet)"""
        late, self._late = self._late, None
        return self._collect(late)


# ------------------------------------------------------------------------------------------------------------------
# The evaluators' gather format.  Every evaluator of the reference collects `self._predictions` -- a Python list of
# {"image_id": ..., "instances": [COCO-json dicts]} per rank -- with detectron2's `comm.gather(self._predictions, dst=0)` and chains
# the per-rank lists on the main process (ape/evaluation/lvis_evaluation.py:101-107, d3_evaluation.py, refcoco_evaluation.py, ...):
# pickled Python objects over a gloo group.  `gather_predictions` is that call with the same return contract (the list of per-rank
# lists on `dst`, [] elsewhere), so an evaluator built on `ape_amd.evaluation.instances_to_coco_json` finishes exactly like the
# reference's.  The per-step tensor all-gathers above stay the fast path for serving; this is the end-of-run exchange.
# ------------------------------------------------------------------------------------------------------------------
_CPU_GROUP = [None]


def _cpu_group():
    """a gloo group over all ranks for pickled-object collectives (detectron2 comm._get_global_gloo_group): RCCL moves tensors only"""
    if dist.get_backend() == "gloo":
        return dist.group.WORLD
    if _CPU_GROUP[0] is None:
        _CPU_GROUP[0] = dist.new_group(backend="gloo")
    return _CPU_GROUP[0]


def gather_predictions(predictions, dst=0):
    """detectron2.utils.comm.gather for the evaluators' prediction lists: -> [rank 0's list, rank 1's list, ...] on rank `dst`,
    [] on the others; single process: [predictions]"""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [predictions]
    group = _cpu_group()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if rank == dst:
        out = [None] * world
        dist.gather_object(predictions, out, dst=dst, group=group)
        return out
    dist.gather_object(predictions, None, dst=dst, group=group)
    return []


def chain_predictions(gathered):
    """`list(itertools.chain(*predictions))` of the evaluators (lvis_evaluation.py:104): one flat list in rank order -- with the
    contiguous shards of shard_indices that is dataset order"""
    import itertools
    return list(itertools.chain(*gathered))
