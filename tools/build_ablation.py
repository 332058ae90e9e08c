"""rebuild ape_amd/lib/libape_hip.so with the p8 ablation instantiations (-DAPE_P8_ABLATION); `python -m ape_amd.build --force`
restores the product library"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ape_amd import build as B  # noqa: E402

B.FILE_FLAGS["gemm_p8.hip"] = B.FLAGS + ["-DAPE_P8_ABLATION"]
print(B.build(force=True))
