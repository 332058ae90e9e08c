"""Time the full-size APE-L_D forward on the GPU with synthetic weights (per-stage wall clock, synchronised)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ape_amd.modeling.build import build_ape, init_synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="L_D")
    ap.add_argument("--k", type=int, default=80)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--prompt", default="name")
    ap.add_argument("--semantic", action="store_true", help="semantic branch on: the K classes are stuff classes")
    ap.add_argument("--topk", type=int, default=0)
    args = ap.parse_args()
    t0 = time.time()
    model = init_synthetic(build_ape(args.size), 0).cuda()
    mv = model.model_vision
    mv.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    S = mv.backbone.padding_constraints["square_size"]
    image = torch.randint(0, 256, (3, S, S), generator=torch.Generator().manual_seed(2)).float().cuda()
    text = torch.randn(args.k, 1024, generator=torch.Generator().manual_seed(3)).cuda()
    if args.topk:
        mv.test_topk_per_image = args.topk
    sem = dict(entity="stuff", stuff_classes=[f"s{i}" for i in range(args.k)]) if args.semantic else None
    print(f"build+init {time.time() - t0:.1f}s", flush=True)
    for it in range(args.iters):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = mv.forward_single(image, text, prompt=args.prompt, semantic=sem)
        res = mv.postprocess_instance(out, (S, S), S, S)
        torch.cuda.synchronize()
        extra = f", sem_seg {tuple(out['sem_seg'].shape)}" if sem else ""
        print(f"iter {it}: {1e3 * (time.perf_counter() - t0):.1f} ms  ({len(res.scores)} instances{extra})", flush=True)
    print("peak mem GB", torch.cuda.max_memory_allocated() / 1e9)
    if args.semantic:
        return
    from ape_amd.runtime import GraphedForward
    run = GraphedForward(mv)
    for it in range(args.iters):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        inst, _ = run(image, text, prompt=args.prompt)
        torch.cuda.synchronize()
        print(f"graphed iter {it}: {1e3 * (time.perf_counter() - t0):.1f} ms  ({len(inst.scores)} instances)", flush=True)
    print("peak mem GB", torch.cuda.max_memory_allocated() / 1e9)
    if args.prompt != "name":
        return
    # per-stage timing (synchronised between stages)
    import ape_amd.modeling.ape_deta.deformable_detr_segm_vl as mod
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    maps = mv.backbone.forward_tokens(image, mv._mean, mv._std)
    torch.cuda.synchronize()
    print(f"backbone+fpn: {1e3 * (time.perf_counter() - t0):.1f} ms")
    t0 = time.perf_counter()
    x = mv.backbone.net.forward_tokens(image, mv._mean, mv._std)
    torch.cuda.synchronize()
    print(f"  vit only: {1e3 * (time.perf_counter() - t0):.1f} ms")
    print("peak mem GB", torch.cuda.max_memory_allocated() / 1e9)


if __name__ == "__main__":
    main()
