"""Multi-process tests of the T-sharded path.

CPU (gloo, world_size 2 and 3): the orchestration in cmf_jl_amd.sharded against the unsharded
oracle, with a numpy stand-in for the per-rank engine.
GPU (gloo between two processes that share the one GPU of the test box): the same orchestration
on the real HIP engine, so the kernels' halo handling is checked against the oracle too.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_ranks(world, engine, out, N, T, K, L, iters, reg, timeout=300, backend="gloo", overlap=False):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0", CMF_TEST_BACKEND=backend,
                   CMF_TEST_OVERLAP="1" if overlap else "0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), engine, out,
                                       str(N), str(T), str(K), str(L), str(iters), str(int(reg))],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=timeout)
            logs.append(o.decode(errors="replace"))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{logs[r][-3000:]}"
    return np.load(out)


def frob_rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def test_partition():
    from cmf_jl_amd.sharded import partition

    assert partition(10, 1, 4) == [(0, 10)]
    assert partition(10, 2, 4) == [(0, 5), (5, 10)]
    b = partition(50000, 8, 20)
    assert b[0] == (0, 6250) and b[-1] == (43750, 50000) and all(t1 - t0 == 6250 for t0, t1 in b)
    assert partition(11, 3, 3) == [(0, 4), (4, 8), (8, 11)]
    with pytest.raises(ValueError):
        partition(10, 8, 20)


@pytest.mark.parametrize("world,reg,overlap", [(2, 0, False), (3, 1, False), (2, 1, True)])
def test_sharded_protocol_cpu_gloo(oracle, tmp_path, world, reg, overlap):
    """overlap=True is the two-step form of the W partial sums (numW started right after the H update)."""
    N, T, K, L, iters = 17, 101, 3, 6, 8
    out = str(tmp_path / "res.npz")
    got = run_ranks(world, "cpu", out, N, T, K, L, iters, reg, overlap=overlap)
    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=L, seed=1234)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=0)
    kw = dict(l1W=0.1, l2W=0.5, l1H=0.1, l2H=0.2) if reg else {}
    Wr, Hr, lr, _ = oracle.fit_mult(data, W0, H0, max_itr=iters, check_convergence=False, **kw)
    assert len(got["bounds"]) == world
    np.testing.assert_allclose(got["loss_hist"], lr, rtol=1e-10)
    np.testing.assert_allclose(got["W"], Wr, rtol=1e-8, atol=1e-13)
    np.testing.assert_allclose(got["H"], Hr, rtol=1e-8, atol=1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("world,N,T,K,L,reg", [(2, 130, 900, 32, 20, 0), (3, 40, 333, 5, 10, 1), (1, 48, 300, 4, 8, 0),
                                                   (4, 70, 517, 32, 20, 1), (2, 65, 256, 64, 33, 0)])
def test_sharded_hip_engine_gloo(oracle, tmp_path, world, N, T, K, L, reg):
    """Ranks share GPU 0; collectives go through the host (gloo).  fp32 tolerances as in test_gpu_parity."""
    iters = 6
    out = str(tmp_path / "res.npz")
    got = run_ranks(world, "hip", out, N, T, K, L, iters, reg)
    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=min(L, 20), seed=1234)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=0)
    kw = dict(l1W=0.1, l2W=0.5, l1H=0.1, l2H=0.2) if reg else {}
    Wr, Hr, lr, _ = oracle.fit_mult(data, W0, H0, max_itr=iters, check_convergence=False, **kw)
    np.testing.assert_allclose(got["loss_hist"], lr, rtol=1e-4)
    assert frob_rel(got["W"], Wr) < 1e-4
    assert frob_rel(got["H"], Hr) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", [False, True])
def test_sharded_hip_engine_rccl_single_rank(oracle, tmp_path, overlap):
    """The RCCL transport itself (backend "nccl"): a one-GPU box can only form a 1-rank group, which still sends the
    [numW | denomW] buffer and the loss scalar through RCCL's all-reduce on the library's stream."""
    N, T, K, L, iters = 130, 900, 32, 20, 6
    out = str(tmp_path / "res.npz")
    got = run_ranks(1, "hip", out, N, T, K, L, iters, 0, backend="nccl", overlap=overlap)
    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=L, seed=1234)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=0)
    Wr, Hr, lr, _ = oracle.fit_mult(data, W0, H0, max_itr=iters, check_convergence=False)
    np.testing.assert_allclose(got["loss_hist"], lr, rtol=1e-4)
    assert frob_rel(got["W"], Wr) < 1e-4
    assert frob_rel(got["H"], Hr) < 1e-4


@pytest.mark.gpu
def test_parameter_sweep_over_ranks(tmp_path):
    """independent fits spread over the ranks of a process group (replicas, one device per
    rank on a real node; here two gloo ranks share GPU 0) give the single-process sweep's results in its order."""
    import cmf_jl_amd as cmf

    out = str(tmp_path / "sweep.npz")
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_sweep_worker.py"), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=300)[0].decode(errors="replace") for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{logs[r][-3000:]}"
    got = np.load(out)
    data = cmf.gen_synthetic(N=40, T=300, seed=1234)
    ref = cmf.parameter_sweep(data, L_vals=(5, 8), K_vals=(2, 3), alg_vals=(":mult",), max_itr=6, seed=0, check_convergence=False)
    assert [tuple(k) for k in got["keys"]] == [(L, K) for (L, K, _) in ref]
    for (L, K, _), r in ref.items():
        np.testing.assert_array_equal(got[f"loss_{L}_{K}"], r.loss_hist)  # same kernels, same seeds: bitwise
        np.testing.assert_array_equal(got[f"W_{L}_{K}"], r.W)


@pytest.mark.gpu
def test_sharded_full_size_matches_single_gpu(tmp_path):
    """BASELINE config 2 (N=2000, T=50000, K=32, L=20) split over 2 ranks (25000 columns each, the shard of a
    2-GPU run; both ranks share GPU 0 here and talk over gloo) against the unsharded rule on the same inputs:
    the halo exchange, the [numW | denomW] all-reduce and the loss reduction at the sizes bench.py --gpus N uses."""
    import cmf_jl_amd as cmf

    iters = 3
    out = str(tmp_path / "full.npz")
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_fullsize_worker.py"), out, str(iters)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=600)[0].decode(errors="replace") for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{logs[r][-3000:]}"
    got = np.load(out)
    data = cmf.gen_synthetic(N=2000, T=50000, seed=1234)
    W0, H0 = cmf.init_rand(data, L=20, K=32, seed=0)
    rule = cmf.MultUpdate(data, W0, H0)
    losses = [rule.compute_loss()]
    for _ in range(iters):
        rule.update_motifs()
        losses.append(rule.update_feature_maps())
    W, H = rule.download()
    rule.close()
    # same kernels on different tilings of T: fp32 summation-order differences only
    np.testing.assert_allclose(got["loss_hist"], losses, rtol=1e-5)
    assert frob_rel(got["W"], W) < 1e-5 and frob_rel(got["H"], H) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("world,N,T,K,L,reg", [(2, 130, 900, 32, 20, 1), (3, 40, 333, 5, 10, 0)])
def test_sharded_hip_engine_overlap_form(oracle, tmp_path, world, N, T, K, L, reg):
    """The two-step W partial sums (cmf_w_partial_num / cmf_w_partial_den, numW started after the H update) on the
    real engine against the oracle."""
    iters = 6
    out = str(tmp_path / "res.npz")
    got = run_ranks(world, "hip", out, N, T, K, L, iters, reg, overlap=True)
    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=min(L, 20), seed=1234)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=0)
    kw = dict(l1W=0.1, l2W=0.5, l1H=0.1, l2H=0.2) if reg else {}
    Wr, Hr, lr, _ = oracle.fit_mult(data, W0, H0, max_itr=iters, check_convergence=False, **kw)
    np.testing.assert_allclose(got["loss_hist"], lr, rtol=1e-4)
    assert frob_rel(got["W"], Wr) < 1e-4
    assert frob_rel(got["H"], Hr) < 1e-4
