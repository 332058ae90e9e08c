"""Golden vectors of the input pipeline: outputs of the library the reference calls (Pillow, through detectron2's
ResizeTransform.apply_image = Image.resize(BILINEAR); ape/engine/defaults.py:213-222) on seeded inputs.
Run in the build container:  python tests/golden/make_imageio_golden.py  ->  tests/golden/imageio_golden.npz"""
import os

import numpy as np
import PIL
from PIL import Image

CASES = [  # (h, w, newh, neww)
    (48, 64, 96, 128), (60, 80, 45, 60), (37, 91, 64, 157), (120, 90, 40, 30), (33, 200, 33, 100), (77, 50, 120, 50),
    (256, 200, 64, 50), (19, 23, 181, 219),
]


def main():
    rng = np.random.default_rng(11)
    out = {"pillow_version": np.array(PIL.__version__)}
    for i, (h, w, nh, nw) in enumerate(CASES):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        if i % 2:                                              # smooth content as well as noise
            yy, xx = np.mgrid[0:h, 0:w]
            img = np.stack([(yy * 255 // max(h - 1, 1)), (xx * 255 // max(w - 1, 1)), ((yy + xx) % 256)], -1).astype(np.uint8)
        out[f"in{i}"] = img
        out[f"size{i}"] = np.array([nh, nw])
        out[f"out{i}"] = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "imageio_golden.npz"), **out)


if __name__ == "__main__":
    main()
