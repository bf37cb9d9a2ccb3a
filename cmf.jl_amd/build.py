"""Builds libcmf_hip.so (the gfx950 kernels + C ABI) in-tree with hipcc.

    python cmf.jl_amd/build.py          # build if stale
    python cmf.jl_amd/build.py --force

hipcc cross-compiles for gfx950 without a GPU, so this also runs in the CPU-only
build container; the resulting .so travels to the GPU box with the repo snapshot.
"""
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcmf_hip.so")
# Four translation units, compiled side by side (round 5: one 3.6 kLoC file, 54 s; now 35 s on the build container's 8 cores): the MU
# rule + C ABI, the HALS / Gram / PGD rules, the T-sharded groups, and the launchers of the few-component kernels
SOURCES = [os.path.join(CSRC, f) for f in ("cmf_api.hip", "cmf_rules.hip", "cmf_groups.hip", "cmf_small.hip")]
DEPS = SOURCES + [os.path.join(CSRC, f) for f in ("cmf_internal.h", "cmf_kernels.h", "cmf_small_k.h", "cmf_workers.h", "cmf_writeback.h", "cmf_rng.h")] + [
    os.path.join(ROOT, "include", "cmf_hip.h")]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libcmf_hip.so cannot be built (there is no CPU fallback)")


def source_digest():
    """What cmf_source_digest() (include/cmf_hip.h) must return for a library built from this tree: the first 16 hex
    characters of SHA-256 over the sources in DEPS order, each preceded by its base name and a newline."""
    h = hashlib.sha256()
    for path in DEPS:
        h.update(os.path.basename(path).encode() + b"\n")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def embedded_digest(lib_path=LIB):
    """The digest compiled into an existing libcmf_hip.so, read from the file without loading it (None if absent)."""
    try:
        with open(lib_path, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    tag = b"cmf_hip gfx950 "
    i = blob.find(tag)
    while i >= 0:
        end = blob.find(b"\0", i)
        text = blob[i:end].decode("ascii", "replace")
        if " src=" in text:
            return text.rsplit(" src=", 1)[1]
        i = blob.find(tag, i + 1)
    return None


def is_stale():
    """A binary is current when the digest compiled into it is the digest of the tree (mtimes do not survive the copy to
    the GPU box, and a prebuilt .so that no longer matches csrc/ must never be loaded silently)."""
    return embedded_digest() != source_digest()


def build_lib(force=False, verbose=False):
    """Builds the library unless the one in place already carries the tree's digest.  Safe to call from many processes at
    once (the ranks of a launcher all import the package): an exclusive lock on `libcmf_hip.so.lock` serialises them, the
    digest is looked at again under the lock so that only the first one compiles, and every build writes to a name of
    its own before the atomic rename -- no process can ever map a library another hipcc is still writing.  `force`
    rebuilds even a current binary (command line), but never twice in a row for callers queued on the lock."""
    import fcntl

    if not force and not is_stale():
        return LIB
    want = source_digest()
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if embedded_digest() == want and not (force and getattr(build_lib, "_cli", False)):
                return LIB  # another process built it while this one waited
            tmp = f"{LIB}.{os.getpid()}.tmp"
            objdir = tempfile.mkdtemp(prefix="cmf_build_")
            flags = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-fvisibility=hidden",  # (only the C ABI is exported)
                     f'-DCMF_SRC_DIGEST="{want}"', "-I", os.path.join(ROOT, "include"), "-I", CSRC]
            try:
                procs, objs = [], []
                for src in SOURCES:  # every translation unit at once
                    obj = os.path.join(objdir, os.path.basename(src) + ".o")
                    cmd = [hipcc_path()] + flags + ["-c", src, "-o", obj]
                    if verbose:
                        print(" ".join(cmd), flush=True)
                    procs.append((cmd, subprocess.Popen(cmd)))
                    objs.append(obj)
                failed = [cmd for cmd, p in procs if p.wait() != 0]
                if failed:
                    raise subprocess.CalledProcessError(1, failed[0])
                link = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp]
                if verbose:
                    print(" ".join(link), flush=True)
                subprocess.check_call(link)
                os.replace(tmp, LIB)  # never leave a half-written library where a loader could find it
            finally:
                shutil.rmtree(objdir, ignore_errors=True)
                if os.path.exists(tmp):
                    os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    build_lib._cli = True
    print(build_lib(force="--force" in sys.argv, verbose=True))
