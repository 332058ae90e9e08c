#!/usr/bin/env python
"""Per-kernel register / LDS / scratch figures of the built library, read from the gfx950 code objects' metadata notes
(works in the CPU container: nothing is executed).

    python tools/kernel_resources.py [substring ...]     # rows whose demangled name contains every substring
    python tools/kernel_resources.py --scratch           # only kernels with a private segment (spills / stack)

Columns: vgpr (arch VGPRs) / agpr / sgpr / lds bytes (static) / scratch bytes per lane / spilled vgprs / max flat workgroup size.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def rows(objdir=os.path.join(ROOT, "ape_amd", "lib", "obj")):
    out = []
    tmp = tempfile.mkdtemp(prefix="ape_res_")
    try:
        for f in sorted(os.listdir(objdir)):
            if not f.endswith(".hip.o"):
                continue
            dst = os.path.join(tmp, f)
            shutil.copy(os.path.join(objdir, f), dst)
            subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", dst], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
            co = [g for g in os.listdir(tmp) if g.startswith(f) and "gfx950" in g]
            if not co:
                continue
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", os.path.join(tmp, co[0])], capture_output=True, text=True).stdout
            for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
                blk = ".agpr_count:" + blk
                g = lambda k, d="0": (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, d])[1]
                name = g("name", "?")
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                dem = re.sub(r"\(.*\)$", "", dem).replace("void ", "")
                out.append(dict(file=f[:-6], name=dem, vgpr=int(g("vgpr_count")), agpr=int(g("agpr_count")), sgpr=int(g("sgpr_count")),
                                lds=int(g("group_segment_fixed_size")), scratch=int(g("private_segment_fixed_size")),
                                spill=int(g("vgpr_spill_count")), wg=int(g("max_flat_workgroup_size"))))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    only_scratch = "--scratch" in sys.argv
    print(f"{'file':10s} {'vgpr':>4s} {'agpr':>4s} {'sgpr':>4s} {'lds':>6s} {'scr':>4s} {'spl':>3s} {'wg':>4s}  kernel")
    for r in rows():
        if only_scratch and r["scratch"] == 0:
            continue
        if all(a in r["name"] for a in args):
            print(f"{r['file']:10s} {r['vgpr']:4d} {r['agpr']:4d} {r['sgpr']:4d} {r['lds']:6d} {r['scratch']:4d} {r['spill']:3d} {r['wg']:4d}  {r['name'][:150]}")
