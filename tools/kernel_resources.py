#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS figures of the gfx950 code objects the four csrc/*.hip translation units compile to (no GPU needed).

    python tools/kernel_resources.py            # all kernels; lines starting with SCRATCH use private memory
    python tools/kernel_resources.py --check    # exit 1 if any kernel has a private segment or spills VGPRs

hipcc --cuda-device-only -> clang-offload-bundler --unbundle -> llvm-readelf --notes (the AMDGPU metadata note).
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_table():
    notes = ""
    csrc = os.path.join(ROOT, "cmf.jl_amd", "csrc")
    with tempfile.TemporaryDirectory() as tmp:
        jobs = []
        for src in sorted(f for f in os.listdir(csrc) if f.endswith(".hip")):  # the translation units, side by side
            obj = os.path.join(tmp, src + ".o")
            jobs.append((src, obj, subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-c", "--cuda-device-only",
                                                     "-I", os.path.join(ROOT, "include"), "-I", csrc, os.path.join(csrc, src), "-o", obj])))
        for src, obj, proc in jobs:
            if proc.wait() != 0:
                raise RuntimeError(f"hipcc failed on {src}")
            co = obj + ".co"
            subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={obj}",
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
            notes += subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co]).decode()
    rows = []
    for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        g = lambda key: int(re.search(rf"\.{key}:\s+(\d+)", blk).group(1))  # noqa: E731
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        try:
            name = subprocess.run(["c++filt", name], stdout=subprocess.PIPE).stdout.decode().strip() or name
        except OSError:
            pass
        rows.append(dict(name=name, vgpr=g("vgpr_count"), agpr=int(blk.split("\n")[0].strip()), sgpr=g("sgpr_count"),
                         scratch=g("private_segment_fixed_size"), vgpr_spills=g("vgpr_spill_count"), sgpr_spills=g("sgpr_spill_count"),
                         lds=g("group_segment_fixed_size")))
    return rows


if __name__ == "__main__":
    rows = kernel_table()
    bad = [r for r in rows if r["scratch"] or r["vgpr_spills"]]
    for r in sorted(rows, key=lambda r: r["name"]):
        tag = "SCRATCH " if (r["scratch"] or r["vgpr_spills"]) else ""
        print(f"{tag}{r['name'][:90]:90s} vgpr {r['vgpr']:3d} agpr {r['agpr']:3d} sgpr {r['sgpr']:3d} scratch {r['scratch']:5d} "
              f"vspill {r['vgpr_spills']:3d} sspill {r['sgpr_spills']:3d} lds {r['lds']}")
    print(f"{len(rows)} kernels, {len(bad)} with scratch / VGPR spills")
    if "--check" in sys.argv and bad:
        sys.exit(1)
