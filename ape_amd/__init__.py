"""ape_amd -- MI355X-native (gfx950) implementation of the APE-L_D inference forward pass.

Only the hot path named in BASELINE.json lives here: HIP kernels + C-ABI (csrc/, lib/), the ctypes
binding (_lib.py, ops.py) and the host-side mirror of the reference's ape.layers / ape.modeling
operator API (layers/, modeling/).
"""
__version__ = "0.1.0"
