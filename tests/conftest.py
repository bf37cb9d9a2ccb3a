import os
import subprocess
import sys

import pytest

# the oracle's OpenMP regions are small: a thread per visible core (64+ on the GPU box, of which the
# container may use a fraction) only adds spinning
os.environ.setdefault("OMP_NUM_THREADS", str(min(8, os.cpu_count() or 1)))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (numpy + C restatement); builds the C part if needed."""
    so = os.path.join(ROOT, "oracle", "libcmf_oracle.so")
    src = os.path.join(ROOT, "oracle", "cmf_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    from oracle import cmf_oracle

    cmf_oracle.c_lib()
    return cmf_oracle
