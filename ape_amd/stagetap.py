"""Stage taps: the `stages=` dict that the forward functions fill with intermediate tensors, plus teacher forcing.

`tap(stages, key, x)` is called at every stage boundary of the forward (ViT blocks, pyramid maps, encoder / decoder layers,
heads).  With a plain dict it records x.  With a `StageTap` that carries a `teacher` (the stage tensors of another run of
the same model -- in the parity tests the fp32-kernel run, which is pinned to the reference's fixtures), it records x and
hands the TEACHER's tensor (cast to x's dtype) to the next stage: every stage then starts from the reference's input, so
`got[key]` vs `teacher[key]` is that stage's own error, free of whatever the stages before it accumulated.
Host-side bookkeeping only; never active in the benchmarked / captured forward (stages=None).
"""
import torch


class StageTap(dict):
    def __init__(self, teacher=None, keep=None):
        super().__init__()
        self.teacher = teacher
        self.keep = keep            # optional set of keys to record (None = all); forcing applies to every teacher key

    @property
    def forcing(self):
        return self.teacher is not None

    def tap(self, key, x):
        if self.keep is None or key in self.keep:
            self[key] = x
        if self.teacher is not None and key in self.teacher:
            t = self.teacher[key]
            if tuple(t.shape) != tuple(x.shape):
                raise ValueError(f"StageTap: teacher tensor {key} has shape {tuple(t.shape)}, stage produced {tuple(x.shape)}")
            return t.to(device=x.device, dtype=x.dtype).contiguous()
        return x


def tap(stages, key, x):
    """record x under `key`; returns the tensor the forward continues with (x, or the teacher's under teacher forcing)"""
    if stages is None:
        return x
    if isinstance(stages, StageTap):
        return stages.tap(key, x)
    stages[key] = x
    return x


def forcing(stages):
    return isinstance(stages, StageTap) and stages.forcing


@torch.no_grad()
def rel_rms(got, ref):
    """||got - ref||_F / ||ref||_F in fp64 accumulation"""
    g, r = got.detach().double().flatten(), ref.detach().double().flatten().to(got.device)
    return ((g - r).pow(2).sum().sqrt() / r.pow(2).sum().sqrt().clamp_min(1e-300)).item()


@torch.no_grad()
def rel_max(got, ref):
    """max |got - ref| / max |ref|"""
    g, r = got.detach().float().flatten(), ref.detach().float().flatten().to(got.device)
    return ((g - r).abs().max() / r.abs().max().clamp_min(1e-30)).item()
