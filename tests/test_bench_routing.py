"""bench.py's launch routing and build provenance (CPU)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_route_picks_the_form_for_gpus_and_world_size():
    import bench

    assert bench.route(1, None) == "single" and bench.route(1, "1") == "single"
    # a plain `python bench.py --gpus 8` (no launcher, WORLD_SIZE unset): one process drives the 8 GPUs
    assert bench.route(8, None) == "multi" and bench.route(2, "") == "multi" and bench.route(4, "1") == "multi"
    # torch.distributed.run --nproc-per-node 8: one process per GPU
    assert bench.route(8, "8") == "ranks" and bench.route(2, 2) == "ranks"
    for gpus, world in ((8, 4), (1, 8), (2, 3)):
        with pytest.raises(ValueError):
            bench.route(gpus, world)
    with pytest.raises(ValueError):
        bench.route(0, None)


def test_comm_record_from_comm_info():
    import bench

    info = "transport=rccl version=22703 lib=/opt/rocm/lib/librccl.so.1 nranks=8 local=8 ranks=0@dev0,1@dev1,2@dev2,3@dev3,4@dev4,5@dev5,6@dev6,7@dev7 overlap=0"
    rec = bench.parse_comm(info, "one-process")
    assert rec["transport"] == "rccl" and rec["nranks"] == 8 and rec["local"] == 8 and rec["version"] == 22703
    assert rec["devices"] == list(range(8)) and rec["mode"] == "one-process" and rec["lib"].endswith("librccl.so.1")
    rec = bench.parse_comm("transport=callbacks nranks=2 local=1 ranks=1@dev1 overlap=0", "one-process-per-gpu", "rank 0: boom")
    assert rec["transport"] == "callbacks" and rec["devices"] == [1] and rec["fallback_from_rccl"] == "rank 0: boom"


def test_bench_sets_the_ipc_mode_itself():
    """HSA_ENABLE_IPC_MODE_LEGACY=0 is what RCCL's dmabuf IPC needs on these hosts: bench.py must not rely on the caller."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    src = src[src.index("def main():"):]
    i_set = src.index('os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")')
    assert i_set < src.index("import numpy as np") and i_set < src.index("import torch")


def test_library_carries_the_digest_of_the_tree():
    """cmf_source_digest() of the built library equals the digest build.py computes from csrc/ + include/: a prebuilt
    .so that no longer matches the sources would be rebuilt or refused by cmf.jl_amd/_lib.load, never loaded silently."""
    import ctypes

    import cmf_jl_amd as cmf
    from cmf_jl_amd import build

    lib = cmf.load_library()
    want = build.source_digest()
    assert lib.cmf_source_digest().decode() == want and build.embedded_digest() == want
    assert f"src={want}" in lib.cmf_version().decode() and f"abi={lib.cmf_abi_version()}" in lib.cmf_version().decode()
    assert lib.cmf_abi_version() == 6
    assert not build.is_stale()
    v = ctypes.c_int64()
    assert lib.cmf_get_counter(None, b"x", ctypes.byref(v)) == 1  # CMF_ERR_ARG: needs a handle


def test_stale_library_is_detected(tmp_path):
    from cmf_jl_amd import build

    fake = tmp_path / "libcmf_hip.so"
    fake.write_bytes(b"\x7fELF....cmf_hip gfx950 0.3.0 abi=3 src=0123456789abcdef\0....")
    assert build.embedded_digest(str(fake)) == "0123456789abcdef" != build.source_digest()
    assert build.embedded_digest(str(tmp_path / "missing.so")) is None


def test_no_shipped_kernel_uses_scratch():
    """Every kernel of the gfx950 code object runs without a private segment and without VGPR spills (SGPR spills to VGPR
    lanes are allowed: transconv_kernel keeps its set-up scalars there, outside the MFMA loop)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources

    rows = kernel_resources.kernel_table()
    assert len(rows) > 60
    bad = [(r["name"], r["scratch"], r["vgpr_spills"]) for r in rows if r["scratch"] or r["vgpr_spills"]]
    assert not bad, bad
    big = {r["name"]: r for r in rows}
    assert big["void transconv_kernel<20>(TcParams)"]["vgpr"] + big["void transconv_kernel<20>(TcParams)"]["agpr"] <= 512


def test_inventory_of_the_device_gated_multi_gpu_cases():
    """The cases of tests/test_multi_gpu.py that need 2-8 distinct devices skip on a one-GPU box (31 `s` in the round-end run).
    This CPU test counts them and checks that every one names, in its id, the launch form, the transport, who enqueues, what
    runs and the device count -- so that the FIRST run on a node is attributable at a glance: a failing id reads
    "one-process.rccl.caller.overlap.n4" (VERDICT round 4, item 5a)."""
    import re
    import subprocess

    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_multi_gpu.py"), "--collect-only", "-q", "-m", "gpu"],
                       capture_output=True, text=True, cwd=ROOT, timeout=300)
    ids = [ln.strip() for ln in p.stdout.splitlines() if "::" in ln]
    assert ids, p.stdout + p.stderr
    pat = re.compile(r"\[(one-process|per-process|bench-plain-launch)\.(rccl|peer)\.(workers|caller|ranks)\.([a-z_-]+)\.n([248])\]$")
    gated = [i for i in ids if pat.search(i)]
    assert len(gated) == 31, (len(gated), ids)
    src = open(os.path.join(ROOT, "tests", "test_multi_gpu.py")).read()
    for i in gated:  # ... and every one of them really is gated on the device count it names
        fn = i.split("::")[1].split("[")[0]
        body = src[src.index(f"def {fn}("):]
        body = body[:body.index("\n\n\n") if "\n\n\n" in body else len(body)]
        assert "_need_devices(n)" in body, fn
    by = {}
    for i in gated:
        m = pat.search(i)
        by.setdefault((m.group(1), m.group(2), m.group(3)), []).append(int(m.group(5)))
    # every form x transport the library offers across devices appears at 2 devices AND at 8 (the driver's node)
    for key in (("one-process", "rccl", "workers"), ("one-process", "rccl", "caller"), ("one-process", "peer", "workers"),
                ("one-process", "peer", "caller"), ("per-process", "rccl", "ranks"), ("bench-plain-launch", "rccl", "workers")):
        assert {2, 8} <= set(by[key]), (key, by.get(key))
    # nothing that needs several devices hides behind an id that does not say so
    others = [i for i in ids if i not in gated]
    for i in others:
        fn = i.split("::")[1].split("[")[0]
        body = src[src.index(f"def {fn}("):]
        body = body[:body.index("\n\n\n") if "\n\n\n" in body else len(body)]
        assert "_need_devices(" not in body, fn
