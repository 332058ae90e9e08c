"""DefaultPredictor -- mirror of ape/engine/defaults.py:157-230 (the class demo/demo_lazy.py and demo/predictor_lazy.py drive):
BGR uint8 image in, `{"instances": Instances, ...}` out.

Construction.  In a full environment (detectron2 importable) pass the LazyConfig: the model is `instantiate(cfg.model)`
(the config's `ape.*` import paths resolve to the HIP-backed classes through the `ape` alias package), the checkpoint is
loaded with `DetectionCheckpointer` and the test-time resize comes from `cfg.dataloader.test.mapper.augmentations[0]`
(`ResizeShortestEdge(short_edge_length=1024, max_size=1024)` in the APE-*_D configs).  Without detectron2 pass a built
model (`ape_amd.modeling.build.build_ape`) and the resize parameters.

Input pipeline.  The reference resizes on the host with PIL (`ResizeTransform.apply_image`, bilinear with
PIL's antialiasing filter support), converts to float32 CHW on the host, and the model normalises + pads on the device.
Here the ORIGINAL uint8 image is uploaded through a pinned staging buffer (3 bytes per source pixel instead of 12 per
resized pixel) and one kernel (csrc/imageio.hip `resize_u8_kernel`, bit exact with Pillow's two-pass fixed-point resampler)
does BGR -> RGB, the resize and the float32 CHW conversion; normalise + pad are fused into the patch-embedding gather
(csrc/spatial.hip `patchify`).  There is no host resize in this package."""
import numpy as np
import torch


def shortest_edge_size(h, w, short_edge_length, max_size):
    """detectron2 ResizeShortestEdge.get_output_shape: scale the short side to `short_edge_length`, cap the long side"""
    scale = short_edge_length * 1.0 / min(h, w)
    newh, neww = (short_edge_length, scale * w) if h < w else (scale * h, short_edge_length)
    if max(newh, neww) > max_size:
        s = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * s, neww * s
    return int(newh + 0.5), int(neww + 0.5)


class DefaultPredictor:
    def __init__(self, cfg=None, model=None, short_edge_length=1024, max_size=1024, input_format="RGB"):
        self.cfg = cfg
        self.aug = None
        if model is None:
            from detectron2.config import instantiate                    # full environment only (LazyConfig)
            from .checkpoint import DetectionCheckpointer                # ape/engine/defaults.py:9,199
            model = instantiate(cfg.model)
            model.to(cfg.train.device)
            DetectionCheckpointer(model).load(cfg.train.init_checkpoint)
            self.aug = instantiate(cfg.dataloader.test.mapper.augmentations[0])
            mv = cfg.model.model_vision if "model_vision" in cfg.model else cfg.model
            input_format = mv.input_format
        self.model = model.eval()
        self.short_edge_length, self.max_size = short_edge_length, max_size
        self.input_format = input_format
        assert self.input_format in ("RGB", "BGR"), self.input_format
        self._pinned = [None, None]       # two staging buffers in rotation: an upload may still be in flight when the next image arrives
        self._pinned_done = [None, None]  # events recorded behind each buffer's host-to-device copy
        self._pinned_next = 0

    def _upload(self, image_hwc):
        """HWC uint8 (host) -> the same bytes on the model's device through a pinned staging buffer"""
        dev = next(self.model.parameters()).device
        t = torch.from_numpy(np.ascontiguousarray(image_hwc))
        if dev.type == "cuda":
            i = self._pinned_next
            self._pinned_next = 1 - i
            if self._pinned_done[i] is not None:
                self._pinned_done[i].synchronize()          # the copy that last read this buffer has finished
            if self._pinned[i] is None or self._pinned[i].numel() < t.numel():
                self._pinned[i] = torch.empty(t.numel(), dtype=torch.uint8, pin_memory=True)
            buf = self._pinned[i][: t.numel()].view(t.shape)
            buf.copy_(t)
            t = buf.to(dev, non_blocking=True)
            self._pinned_done[i] = torch.cuda.Event()
            self._pinned_done[i].record()
        return t

    def _resize_params(self):
        """(short_edge_length, max_size) of the test augmentation (ResizeShortestEdge with a single length in the APE configs)"""
        if self.aug is not None and hasattr(self.aug, "short_edge_length"):
            sel = self.aug.short_edge_length
            return int(sel[0] if isinstance(sel, (tuple, list)) else sel), int(self.aug.max_size)
        return self.short_edge_length, self.max_size

    def preprocess(self, original_image):
        """BGR uint8 [H, W, 3] (host) -> the model's `image` input: float32 [3, newh, neww] on the device (:213-220)"""
        from . import ops
        height, width = original_image.shape[:2]
        sel, mx = self._resize_params()
        newh, neww = shortest_edge_size(height, width, sel, mx)
        raw = self._upload(original_image)
        return ops.resize_bilinear_u8(raw, newh, neww, float_chw=True, flip=self.input_format == "RGB")

    def preprocess_mask(self, mask_prompt, newh, neww):
        """(:226-228) the prompt mask goes through the SAME transform as the image: ResizeTransform.apply_image on the uint8 mask
        = Pillow's bilinear resize of a single-channel image -- here the resize kernel on three copies of the channel.
        uint8 [H, W] (host) -> float32 [newh, neww] on the device"""
        from . import ops
        m = np.ascontiguousarray(np.repeat(np.asarray(mask_prompt, dtype=np.uint8)[:, :, None], 3, axis=2))
        return ops.resize_bilinear_u8(self._upload(m), newh, neww, float_chw=True)[0].contiguous()

    @torch.no_grad()
    def __call__(self, original_image, text_prompt=None, mask_prompt=None):
        """original_image: np.ndarray [H, W, 3] uint8 in BGR order (cv2.imread) -> predictions dict of the model"""
        height, width = original_image.shape[:2]
        inputs = {"image": self.preprocess(original_image), "height": height, "width": width}
        if text_prompt is not None:
            inputs["prompt"] = "text"
            inputs["text_prompt"] = text_prompt
        if mask_prompt is not None:
            inputs["mask_prompt"] = self.preprocess_mask(mask_prompt, *inputs["image"].shape[-2:])
        return self.model([inputs])[0]
