"""Builds nerf_amd/csrc/libnerfhip.so (gfx950 only) with hipcc, in-tree.

The library is a plain C-ABI shared object (see include/nerf_hip.h); nothing here depends on
torch.  hipcc cross-compiles without a GPU, so this runs anywhere the ROCm toolchain is present.
Objects are rebuilt only when their source (or a shared header) is newer.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libnerfhip.so")
ARCH = "gfx950"
# -ffp-contract=off: the fp32 kernels keep the reference's separate multiply and add roundings.
FLAGS = ["-O3", f"--offload-arch={ARCH}", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden",
         "-std=c++17", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(HERE, "..", "include", "nerf_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def build(force=False, verbose=True):
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    hm = _headers_mtime()
    jobs = []
    objs = []
    for src in sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src[:-4] + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hm):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + FLAGS + os.environ.get("NERF_HIP_EXTRA_FLAGS", "").split() + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return o

    if jobs:
        if verbose:
            print(f"[nerf_amd.build] compiling {len(jobs)} source(s) for {ARCH}")
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    if jobs or not os.path.exists(LIB):
        cmd = [hipcc, "-shared", f"--offload-arch={ARCH}", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[nerf_amd.build] linked {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
