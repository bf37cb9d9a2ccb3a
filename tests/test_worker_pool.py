"""The enqueue-worker pool of the T-sharded groups (cmf.jl_amd/csrc/cmf_workers.h: single-producer queues, sleep / wake-up, the
meeting point, the abort protocol) is free of HIP, so it is stressed HERE, on the CPU, under ThreadSanitizer
(tests/worker_pool_stress.cpp): GPU sanitizers are not available on the boxes, and a memory-ordering slip in this code would
show on a multi-GPU node once in a long while."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_worker_pool_under_sanitizers(tmp_path, sanitizer):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    exe = str(tmp_path / "stress")
    subprocess.check_call([gxx, "-std=c++17", "-O1", "-g", f"-fsanitize={sanitizer}", "-fno-omit-frame-pointer", "-pthread",
                           "-I", os.path.join(ROOT, "cmf.jl_amd", "csrc"), os.path.join(ROOT, "tests", "worker_pool_stress.cpp"), "-o", exe])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=1")
    for workers, batches in ((4, 200), (8, 120), (2, 200)):
        p = subprocess.run([exe, str(workers), str(batches)], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout + p.stderr
        assert p.stdout.startswith(f"ok {batches} "), p.stdout + p.stderr
        assert "WARNING: ThreadSanitizer" not in p.stderr and "ERROR: AddressSanitizer" not in p.stderr, p.stderr
