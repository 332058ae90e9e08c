"""per-ViT-block error of the fp32 HIP path vs the reference fixture of APE-E_D at full size (free-running)"""
import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import model_util as M, oracle_util as U
case = sys.argv[1] if len(sys.argv) > 1 else "E_D_coco80"
dt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[sys.argv[2] if len(sys.argv) > 2 else "f32"]
model, image, text, gold = M.build_model(case, "cuda", dt)
mv = model.model_vision
stages = {}
mv.forward_single(image.cuda(), text.cuda(), stages=stages, prompt=U.case_prompt(gold))
P = mv.backbone.net.packed(dt)
r2t = P["r2t"].long()
for k, fp in gold["stages"].items():
    name = k.replace("vit_block", "vit_blk")
    if name not in stages:
        continue
    t = stages[name].float()
    if name.startswith("vit_blk") or name == "vit_embed":
        t = t[r2t]                                   # window-major -> raster: the reference's [1, hw, hw, E]
    else:
        t = M.ref_layout(name, t, fp["shape"])
    if t.numel() != int(torch.tensor(fp["shape"]).prod()):
        print(name, "shape", tuple(t.shape), fp["shape"]); continue
    got = t.reshape(-1)[fp["idx"]].cpu()
    want = fp["samples"].float()
    print(f"{name:16s} max err / absmax {((got - want).abs().max() / fp['absmax']).item():.3e}   rms rel {((got - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item():.3e}  absmax {fp['absmax']:.3g}")
