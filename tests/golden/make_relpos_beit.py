"""Golden vectors of the reference's BEiT-style relative-position table resize (get_rel_pos(..., interp_type="beit"),
/root/reference/ape/modeling/backbone/utils_eva.py:92-118; scipy cubic interp1d over geometric-progression nodes).
Run in the build container (needs /root/reference):  python tests/golden/make_relpos_beit.py
Writes tests/golden/relpos_beit.pt: [(table [src, C], size, get_rel_pos(size, size, table, "beit") [size, size, C]), ...]."""
import importlib.util
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_utils_eva", "/root/reference/ape/modeling/backbone/utils_eva.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

torch.manual_seed(20260930)
cases = []
for src, size, C in ((27, 16, 8), (27, 32, 4), (9, 7, 8), (63, 24, 6), (31, 16, 8)):       # incl. one table that already has 2 size - 1 rows
    table = torch.randn(src, C)
    cases.append((table, size, ref.get_rel_pos(size, size, table, "beit").clone()))
torch.save(cases, os.path.join(HERE, "relpos_beit.pt"))
print("wrote", len(cases), "cases")
