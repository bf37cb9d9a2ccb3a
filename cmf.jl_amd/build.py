"""Builds libcmf_hip.so (the gfx950 kernels + C ABI) in-tree with hipcc.

    python cmf.jl_amd/build.py          # build if stale
    python cmf.jl_amd/build.py --force

hipcc cross-compiles for gfx950 without a GPU, so this also runs in the CPU-only
build container; the resulting .so travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcmf_hip.so")
SOURCES = [os.path.join(CSRC, "cmf_api.hip")]
DEPS = SOURCES + [os.path.join(CSRC, f) for f in ("cmf_kernels.h", "cmf_group.h", "cmf_rng.h")] + [
    os.path.join(ROOT, "include", "cmf_hip.h")]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libcmf_hip.so cannot be built (there is no CPU fallback)")


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build_lib(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    cmd = [hipcc_path(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC] + SOURCES + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
