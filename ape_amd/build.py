"""Build libape_hip.so (all HIP kernels + the C-ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the CPU container; the resulting
ape_amd/lib/libape_hip.so is git-ignored but travels to the GPU box with the snapshot.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libape_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-mfma-vgpr-form: MFMA accumulators live in VGPRs (gfx950 has a unified file); without it hipcc parks them in AGPRs and
# the softmax / epilogue code pays a v_accvgpr_read/write per value (176 of ~500 VALU slots per attention key tile)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]
# per-file overrides: the fused FFN keeps 160 accumulator registers per lane next to 64 operand and 64 prefetch registers -- one
# wave per SIMD owns the whole 512-entry file, so its accumulators belong in the AGPR half (with the VGPR form hipcc shuffles
# them through v_accvgpr moves between the MFMAs of a batch)
# attention: the softmax never produces a NaN (key 0 is visible to every query, masked scores are -inf and only ever meet finite
# maxima), so fmaxf needs no NaN-quieting: under the default IEEE mode hipcc canonicalises every operand of a max with a
# `v_max x, x` -- 30 of the ~160 VALU instructions a wave issues per key tile in a VALU-bound kernel
FILE_FLAGS = {"ffn_fused.hip": ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-mllvm", "-amdgpu-mfma-vgpr-form=0"],
              "attention.hip": FLAGS + ["-fno-honor-nans", "-mno-amdgpu-ieee"]}


def _sources():
    return sorted(
        os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip") or f.endswith(".cpp")
    )


def _digest():
    h = hashlib.sha256()
    # EVERY header of csrc/ (round 5: gemm_epi.h was missing here -- an edit of the shared epilogue did not rebuild the library, and one GPU
    # run validated a stale .so)
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    for f in _sources() + headers + [os.path.join(HERE, "..", "include", "ape_hip.h")]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(FILE_FLAGS.items())).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "libape_hip.sha256")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    objs = []
    procs = []
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    # per-object stamps: an object is rebuilt only when its own source, a header of csrc/ (or include/ape_hip.h) or its flags changed --
    # iterating on one kernel then costs one hipcc run, not eighteen
    hh = hashlib.sha256()
    for f in sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(HERE, "..", "include", "ape_hip.h")]:
        with open(f, "rb") as fh:
            hh.update(fh.read())
    headers_digest = hh.hexdigest()
    for src in _sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = [HIPCC] + FILE_FLAGS.get(os.path.basename(src), FLAGS) + ["-c", src, "-o", obj]
        if src.endswith(".cpp"):
            cmd = [HIPCC, "-O2", "-std=c++17", "-fPIC", "-c", src, "-o", obj]
        with open(src, "rb") as fh:
            odig = hashlib.sha256(fh.read() + headers_digest.encode() + " ".join(cmd).encode()).hexdigest()
        ostamp = obj + ".sha256"
        if not force and os.path.exists(obj) and os.path.exists(ostamp) and open(ostamp).read().strip() == odig:
            continue
        if os.path.exists(ostamp):
            os.remove(ostamp)
        if verbose:
            print("[ape_amd.build]", " ".join(cmd), flush=True)
        procs.append((src, ostamp, odig, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = None
    for src, ostamp, odig, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            sys.stderr.write(out.decode())
            failed = failed or src
            continue
        with open(ostamp, "w") as fh:
            fh.write(odig)
    if failed:
        raise RuntimeError(f"hipcc failed on {failed}")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-L/opt/rocm/lib", "-lhsa-runtime64"]
    if verbose:
        print("[ape_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, "w") as fh:
        fh.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
