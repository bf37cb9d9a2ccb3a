"""The header as a C compiler sees it: tests/abi_driver.c includes include/cmf_hip.h, is built with
`gcc -std=c99 -Wall -Werror`, links libcmf_hip.so and replays a golden fixture through the reference's call sequence
(MultUpdate ctor, compute_loss, update_motifs! / update_feature_maps! per iteration; alternating.jl:37,52,54).
What is proven here is the header's prototypes -- not the ctypes table of cmf.jl_amd/_lib.py."""
import os
import struct
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIBDIR = os.path.join(ROOT, "cmf.jl_amd")


def build_driver(outdir):
    import __graft_entry__

    __graft_entry__.build(quiet=True)
    exe = os.path.join(str(outdir), "abi_driver")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           os.path.join(HERE, "abi_driver.c"), "-o", exe, "-L", LIBDIR, "-lcmf_hip", "-lm",
                           "-Wl,-rpath," + LIBDIR])
    return exe


def test_header_compiles_as_c99_and_driver_links(tmp_path):
    """CPU: the header is valid C99 on its own (and as C++), and the driver builds and links against the library."""
    for comp, std in (("gcc", "-std=c99"), ("g++", "-std=c++11")):
        src = tmp_path / ("hdr_only." + ("c" if comp == "gcc" else "cpp"))
        src.write_text('#include "cmf_hip.h"\nint main(void) { return CMF_OK; }\n')
        subprocess.check_call([comp, std, "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                               "-fsyntax-only", str(src)])
    exe = build_driver(tmp_path)
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 2 and "usage" in p.stderr  # runs far enough to load the library


def _run(exe, tmp_path, fixture, ndev, rule="mult"):
    d = np.load(os.path.join(HERE, "golden", fixture))
    data, W0, H0 = (np.asfortranarray(d[k]) for k in ("data", "W0", "H0"))
    K, N, L = W0.shape
    T = data.shape[1]
    iters = int(d["max_itr"])
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("<5q4d", N, T, K, L, iters, float(d["l1W"]), float(d["l2W"]), float(d["l1H"]), float(d["l2H"])))
        for a in (data, W0, H0):
            f.write(a.tobytes(order="F"))  # Julia memory order
    p = subprocess.run([exe, fin, fout, str(ndev), rule], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    raw = np.fromfile(fout, dtype=np.float64)
    per = iters + 1 + W0.size + H0.size
    assert raw.size == 2 * per
    outs = []
    for q in range(2):
        blk = raw[q * per:(q + 1) * per]
        outs.append((blk[: iters + 1], blk[iters + 1: iters + 1 + W0.size].reshape(W0.shape, order="F"),
                     blk[iters + 1 + W0.size:].reshape(H0.shape, order="F")))
    return d, outs, p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("fixture,ndev", [("mu_small.npz", 0), ("mu_small_reg.npz", 0), ("mu_k5.npz", 0), ("mu_small.npz", 3)])
def test_c_driver_reproduces_golden_fixture(tmp_path, fixture, ndev):
    exe = build_driver(tmp_path)
    d, outs, log = _run(exe, tmp_path, fixture, ndev)
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    for loss, W, H in outs:  # call-by-call, then cmf_fit
        np.testing.assert_allclose(loss, d["loss_hist"], rtol=1e-4)
        assert rel(W, d["W"]) < 1e-4 and rel(H, d["H"]) < 1e-4
    # both passes run the same kernels in the same order
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    np.testing.assert_array_equal(outs[0][2], outs[1][2])
    if ndev:
        assert f"nranks={ndev}" in log


@pytest.mark.gpu
@pytest.mark.parametrize("fixture,ndev,rule", [("hals_small.npz", 0, "hals"), ("pgd_small.npz", 0, "pgd"), ("pgd_small.npz", 2, "pgd"),
                                               ("mu_small_reg.npz", 0, "gram"), ("mu_small.npz", 3, "gram")])
def test_c_driver_other_rules(tmp_path, fixture, ndev, rule):
    """The HALS and PGD entries and the Gram option through the header from C99, against the committed fixtures (HALS:
    hals.jl:31-42; PGD: pgd.jl:158-202 with its defaults, also as a 2-shard group; Gram form: the MU fixtures, also sharded)."""
    exe = build_driver(tmp_path)
    d, outs, log = _run(exe, tmp_path, fixture, ndev, rule)
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    for loss, W, H in outs:
        np.testing.assert_allclose(loss, d["loss_hist"], rtol=1e-4)
        assert rel(W, d["W"]) < 1e-4 and rel(H, d["H"]) < 1e-4
    np.testing.assert_array_equal(outs[0][1], outs[1][1])  # the two passes repeat the same arithmetic
    if ndev:
        assert f"nranks={ndev}" in log
    assert "abi=6" in log and "src=" in log
