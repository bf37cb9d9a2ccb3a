"""Tests of the T-sharded path.

CPU (gloo, world_size 2 and 3): the Python mirror of the library's group protocol (tests/shard_protocol_cpu.py)
with a numpy stand-in for the per-rank engine, against the unsharded oracle.
GPU: the library's own group iteration (csrc/cmf_groups.hip) through the C ABI --
  * one process, several shards on GPU 0 (cmf_create_multi, loopback transport): middle-rank shards, config-3 shard size;
  * one process, RCCL transport with a single device (all a one-GPU box can form);
  * one process per shard (cmf_create_shard + cmf_comm_init_*): gloo ranks sharing GPU 0 through the host-callback
    transport, and a single rank over RCCL.
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


def run_ranks(world, engine, out, N, T, K, L, iters, reg, timeout=300, backend="gloo", overlap=False, mode="calls", transport="",
              fallback=False, rank_env=None, distinct_devices=False, halo_in_ar=True):
    """rank_env: {rank: {VAR: value}} extra environment of single ranks; distinct_devices: rank r works on GPU r
    (LOCAL_RANK=r) instead of all ranks sharing GPU 0."""
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r) if distinct_devices else "0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0", CMF_TEST_BACKEND=backend,
                   CMF_TEST_OVERLAP="1" if overlap else "0", CMF_TEST_MODE=mode, CMF_TEST_TRANSPORT=transport,
                   CMF_TEST_FALLBACK="1" if fallback else "0", CMF_TEST_HALO_IN_AR="1" if halo_in_ar else "0")
        env.update((rank_env or {}).get(r, {}))
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


REG = dict(l1W=0.1, l2W=0.5, l1H=0.1, l2H=0.2)


def oracle_fit(oracle, N, T, K, L, iters, reg):
    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=min(L, 20), seed=1234)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=0)
    Wr, Hr, lr, _ = oracle.fit_mult(data, W0, H0, max_itr=iters, check_convergence=False, **(REG if reg else {}))
    return data, W0, H0, Wr, Hr, lr


def test_partition():
    from cmf_jl_amd.sharded import partition

    assert partition(10, 1, 4) == [(0, 10)]
    assert partition(10, 2, 4) == [(0, 5), (5, 10)]
    b = partition(50000, 8, 20)
    assert b[0] == (0, 6250) and b[-1] == (43750, 50000) and all(t1 - t0 == 6250 for t0, t1 in b)
    b = partition(400000, 8, 20)  # BASELINE config 3
    assert all(t1 - t0 == 50000 for t0, t1 in b)
    assert partition(11, 3, 3) == [(0, 4), (4, 8), (8, 11)]
    with pytest.raises(ValueError):
        partition(10, 8, 20)


def test_loss_scalar_own_slot_encoding():
    """The (hi, lo) float pair of loss_tail_kernel reproduces the double far below the 1e-4 bar, and adding zeros is exact."""
    from shard_protocol_cpu import split_hi_lo

    for x in (0.0, 1.0, 3.141592653589793e9, 7.25e-3, 1.2345678901234567e12):
        hi, lo = split_hi_lo(x)
        assert abs((hi + lo) - x) <= 1e-13 * abs(x)
        assert np.float32(hi) == hi and np.float32(lo) == lo  # both are representable floats


@pytest.mark.parametrize("world,reg,overlap,mode,halo_in_ar", [(2, 0, False, "calls", True), (3, 1, False, "calls", True), (2, 1, True, "calls", True),
                                                                 (3, 0, False, "iterate", True), (2, 1, True, "iterate", True),
                                                                 (3, 1, False, "iterate", False), (2, 0, False, "calls", False), (4, 1, True, "iterate", True)])
def test_sharded_protocol_cpu_gloo(oracle, tmp_path, world, reg, overlap, mode, halo_in_ar):
    """overlap=True is the two-step form of the W partial sums (numW started right after the H update); mode
    "iterate" reads every loss one iteration late from the tail of the next all-reduce (cmf_iterate); halo_in_ar: the halo of H in
    the tail of the W-phase all-reduce (round 6: ONE collective per iteration; every shard with a left neighbour updates the L-1
    columns in front of its own) or, False, in an all-gather of its own behind every H update (rounds 1-5)."""
    N, T, K, L, iters = 17, 101, 3, 6, 8
    out = str(tmp_path / "res.npz")
    got = run_ranks(world, "cpu", out, N, T, K, L, iters, reg, overlap=overlap, mode=mode, halo_in_ar=halo_in_ar)
    _, _, _, Wr, Hr, lr = oracle_fit(oracle, N, T, K, L, iters, reg)
    assert len(got["bounds"]) == world
    # the collectives of the `iters` iterations: the bulk all-reduce (two in the overlap form); all-gathers = the halo exchange of the
    # old form (one per iteration) + the loss read-out (call by call: one per iteration; cmf_iterate: one flush at the end)
    assert int(got["all_reduces"]) == (2 * iters + 1 if overlap else iters)  # (overlap: numW of the NEXT iteration starts behind every H update)
    assert int(got["all_gathers"]) == (0 if halo_in_ar else iters) + (iters if mode == "calls" else 1)
    np.testing.assert_allclose(got["loss_hist"], lr, rtol=1e-10)
    np.testing.assert_allclose(got["W"], Wr, rtol=1e-8, atol=1e-13)
    np.testing.assert_allclose(got["H"], Hr, rtol=1e-8, atol=1e-13)


@pytest.mark.parametrize("world,reg,mode", [(2, 0, "calls"), (3, 1, "iterate"), (4, 1, "calls")])
def test_sharded_gram_form_cpu_gloo(oracle, tmp_path, world, reg, mode):
    """The Gram form on a T-sharded group, stated in numpy over gloo (tests/shard_rules_cpu.py): numW and every rank's
    additive share of HH = H_unfold H_unfold' (from the definition, over the columns the rank owns) in ONE all-reduce of
    [numW | HH | tail], denomW = HH W on every rank, denomH from the pairwise products of W applied to H with both halos and
    the truncation on the last rank only -- against the unsharded oracle."""
    N, T, K, L, iters = 17, 121, 3, 6, 6
    out = str(tmp_path / "res.npz")
    got = run_ranks(world, "cpu_gram", out, N, T, K, L, iters, reg, mode=mode)
    _, _, _, Wr, Hr, lr = oracle_fit(oracle, N, T, K, L, iters, reg)
    np.testing.assert_allclose(got["loss_hist"], lr, rtol=1e-9)
    np.testing.assert_allclose(got["W"], Wr, rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(got["H"], Hr, rtol=1e-7, atol=1e-12)


@pytest.mark.parametrize("world,variant", [(2, ""), (3, "_masked"), (2, "_unitnorm"), (3, "_abs_masked")])
def test_sharded_pgd_cpu_gloo(oracle, tmp_path, world, variant):
    """The PGD rule on a T-sharded group in numpy over gloo: one all-reduce of the partial gradW, gradH from the residual on
    the own columns and the right lag halo, the squared norm of gradH / the UnitNorm component norms / the loss summed over the
    ranks in rank order, a replicated step-size state machine -- against the unsharded oracle, decisions included."""
    N, T, K, L, iters = 17, 121, 3, 6, 6
    out = str(tmp_path / "res.npz")
    got = run_ranks(world, "cpu_pgd" + variant, out, N, T, K, L, iters, 0)
    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=min(L, 20), seed=1234)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=0)
    kw = {}
    if "masked" in variant:
        kw["mask"] = (np.random.default_rng(5).uniform(size=data.shape) > 0.25).astype(float)
    if "abs" in variant:
        kw["loss"] = "abs"
    if "unitnorm" in variant:
        kw.update(constrW="unitnorm", constrH="unitnorm")
    Wr, Hr, lr, sr = oracle.fit_pgd(data, W0, H0, max_itr=iters, **kw)
    np.testing.assert_allclose(got["loss_hist"], lr, rtol=1e-9)
    np.testing.assert_allclose(got["W"], Wr, rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(got["H"], Hr, rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(got["steps"], sr, rtol=1e-12)


# ---------------------------------------------------------------------------------------------------------------
# GPU: one process drives the group (cmf_create_multi)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("R,N,T,K,L,reg,overlap", [(2, 130, 900, 32, 20, 0, False), (3, 40, 333, 5, 10, 1, False),
                                                    (1, 48, 300, 4, 8, 0, False), (4, 70, 517, 32, 20, 1, True),
                                                    (2, 65, 256, 64, 33, 0, False), (3, 130, 900, 32, 20, 1, True),
                                                    (8, 96, 700, 32, 20, 0, False)])
def test_group_loopback_against_oracle(oracle, R, N, T, K, L, reg, overlap):
    """R shards of one problem on GPU 0 behind ONE handle: the reference's two calls per iteration run the whole
    sharded iteration in the library.  Checked against the unsharded fp64 oracle, call by call and as a cmf_iterate batch."""
    import cmf_jl_amd as cmf

    iters = 6
    data, W0, H0, Wr, Hr, lr = oracle_fit(oracle, N, T, K, L, iters, reg)
    kw = REG if reg else dict(l1W=0.0, l2W=0.0, l1H=0.0, l2H=0.0)
    results = []
    for mode in ("calls", "iterate"):
        rule = cmf.MultUpdate(data, W0, H0, devices=[0] * R)
        assert f"nranks={R}" in rule.comm_info() and "loopback" in rule.comm_info()
        if overlap:
            rule.set_option("allreduce_overlap", 1)
        losses = [rule.compute_loss()]
        if mode == "calls":
            for _ in range(iters):
                rule.update_motifs(l1W=kw["l1W"], l2W=kw["l2W"])
                losses.append(rule.update_feature_maps(l1H=kw["l1H"], l2H=kw["l2H"]))
        else:
            losses += list(rule.iterate(iters, **kw))
        W, H = rule.download()
        rule.close()
        np.testing.assert_allclose(losses, lr, rtol=1e-4)
        assert frob_rel(W, Wr) < 1e-4 and frob_rel(H, Hr) < 1e-4
        results.append((np.asarray(losses), W, H))
    # the pipelined batch is the same arithmetic as the call-by-call loop
    np.testing.assert_allclose(results[0][0], results[1][0], rtol=1e-12)
    np.testing.assert_array_equal(results[0][1], results[1][1])
    np.testing.assert_array_equal(results[0][2], results[1][2])


@pytest.mark.gpu
def test_group_fit_native_and_fit_cnmf_devices(oracle):
    """cmf_fit on a group handle (both its pipelined and its convergence-checking form) and fit_cnmf(devices=...)."""
    import cmf_jl_amd as cmf

    N, T, K, L, iters = 70, 517, 5, 10, 12
    data, W0, H0, Wr, Hr, lr = oracle_fit(oracle, N, T, K, L, iters, 0)
    rule = cmf.MultUpdate(data, W0, H0, devices=[0, 0, 0])
    lh, th, early = rule.fit_native(iters, np.inf, False, 3, 1e-4, False)
    W, H = rule.download()
    rule.close()
    assert not early and len(lh) == iters + 1 and th[0] == 0.0 and np.all(np.diff(th) >= 0)
    np.testing.assert_allclose(lh, lr, rtol=1e-4)
    assert frob_rel(W, Wr) < 1e-4 and frob_rel(H, Hr) < 1e-4
    # with the convergence test on, the group stops at the oracle's iteration
    _, _, lr2, _ = oracle.fit_mult(data, W0, H0, max_itr=200, check_convergence=True, patience=3, tol=2e-3)
    rule = cmf.MultUpdate(data, W0, H0, devices=[0, 0])
    lh2, _, early2 = rule.fit_native(200, np.inf, True, 3, 2e-3, False)
    rule.close()
    assert early2 and len(lh2) == len(lr2)
    np.testing.assert_allclose(lh2, lr2, rtol=1e-4)
    # the public API with devices=[...]
    res = cmf.fit_cnmf(data, L=L, K=K, alg=":mult", max_itr=iters, check_convergence=False, W_init=W0, H_init=H0, devices=[0, 0])
    np.testing.assert_allclose(res.loss_hist, lr, rtol=1e-4)
    assert frob_rel(res.W, Wr) < 1e-4 and frob_rel(res.H, Hr) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", [False, True])
def test_group_rccl_single_device(oracle, overlap):
    """The RCCL transport inside the library (ncclCommInitAll, ncclAllReduce / ncclAllGather on the handle's stream):
    a one-GPU box can form a 1-device communicator, which still sends [numW | denomW | tail] and the halo blocks
    through RCCL."""
    import cmf_jl_amd as cmf
    from cmf_jl_amd import _lib

    N, T, K, L, iters = 130, 900, 32, 20, 6
    data, W0, H0, Wr, Hr, lr = oracle_fit(oracle, N, T, K, L, iters, 0)
    rule = cmf.MultUpdate(data, W0, H0, devices=[0], transport=_lib.CMF_COMM_RCCL)
    info = rule.comm_info()
    assert "transport=rccl" in info and "librccl" in info and "nranks=1" in info
    if overlap:
        rule.set_option("allreduce_overlap", 1)
    losses = [rule.compute_loss()]
    for _ in range(2):
        rule.update_motifs()
        losses.append(rule.update_feature_maps())
    losses += list(rule.iterate(iters - 2))
    W, H = rule.download()
    rule.close()
    np.testing.assert_allclose(losses, lr, rtol=1e-4)
    assert frob_rel(W, Wr) < 1e-4 and frob_rel(H, Hr) < 1e-4


@pytest.mark.gpu
def test_group_errors():
    import cmf_jl_amd as cmf
    from cmf_jl_amd import _lib

    data = np.ones((8, 64))
    W, H = np.ones((2, 8, 4)), np.ones((2, 64))
    with pytest.raises(cmf.CMFError) as ei:  # RCCL cannot put two ranks on one device
        cmf.MultUpdate(data, W, H, devices=[0, 0], transport=_lib.CMF_COMM_RCCL)
    assert ei.value.code == _lib.CMF_ERR_ARG
    with pytest.raises(cmf.CMFError) as ei:  # every shard needs >= L-1 columns
        cmf.MultUpdate(np.ones((8, 10)), W, np.ones((2, 10)), devices=[0] * 8)
    assert ei.value.code == _lib.CMF_ERR_UNSUPPORTED
    rule = cmf.MultUpdate(data, W, H, devices=[0, 0])
    with pytest.raises(cmf.CMFError):
        rule.set_option("gram", 2)  # the Gram-sum loss is not available on groups (gram = 1 is: tests/test_group_rules.py)
    lib = cmf.load_library()
    import ctypes

    loss = ctypes.c_double()
    assert lib.cmf_hals_update_feature_maps(rule._h, 0.0, 0.0, ctypes.byref(loss)) == _lib.CMF_ERR_STATE
    rule.close()


# ---------------------------------------------------------------------------------------------------------------
# GPU: one process per shard (cmf_create_shard + cmf_comm_init_*)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("world,N,T,K,L,reg,overlap,mode", [(2, 130, 900, 32, 20, 0, False, "calls"), (3, 40, 333, 5, 10, 1, False, "iterate"),
                                                             (3, 70, 517, 32, 20, 1, True, "iterate"), (2, 65, 256, 64, 33, 0, False, "fit"),
                                                             (2, 48, 300, 4, 8, 1, False, "fit_timed")])
def test_sharded_processes_host_callbacks(oracle, tmp_path, world, N, T, K, L, reg, overlap, mode):
    """Ranks share GPU 0; the library's collectives go through its host-callback transport (gloo)."""
    iters = 6
    out = str(tmp_path / "res.npz")
    got = run_ranks(world, "hip", out, N, T, K, L, iters, reg, overlap=overlap, mode=mode)
    assert "transport=callbacks" in str(got["info"]) and f"nranks={world}" in str(got["info"])
    # every rank handed the library the L-1 columns of data in front of its shard (cmf_shard_set_left_data): where the shape allows it the
    # halo of H travels in the W-phase all-reduce (one collective per iteration), and every rank agreed on that when it joined
    assert f"halo_in_allreduce={1 if K % 32 == 0 else 0}" in str(got["info"])
    _, _, _, Wr, Hr, lr = oracle_fit(oracle, N, T, K, L, iters, reg)
    np.testing.assert_allclose(got["loss_hist"], lr, rtol=1e-4)
    assert frob_rel(got["W"], Wr) < 1e-4
    assert frob_rel(got["H"], Hr) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("backend,mode,overlap", [("nccl", "calls", False), ("gloo", "iterate", False), ("gloo", "iterate", True), ("nccl", "calls", True)])
def test_sharded_process_rccl_single_rank(oracle, tmp_path, backend, mode, overlap):
    """cmf_comm_unique_id + cmf_comm_init_rccl (ncclCommInitRank): the id travels through the torch.distributed group
    (its RCCL backend or gloo -- it is only the rendezvous), the collectives are the library's own RCCL calls.  With the overlap
    form the front end hands a SECOND id over the same way (cmf_comm_init_overlap): the communication stream's own communicator
    (lanes=2 in cmf_comm_info)."""
    N, T, K, L, iters = 130, 900, 32, 20, 6
    out = str(tmp_path / "res.npz")
    got = run_ranks(1, "hip", out, N, T, K, L, iters, 0, backend=backend, mode=mode, transport="rccl", overlap=overlap)
    assert "transport=rccl" in str(got["info"])
    assert ("lanes=2" in str(got["info"])) == overlap and (f"overlap={int(overlap)}" in str(got["info"]))
    _, _, _, Wr, Hr, lr = oracle_fit(oracle, N, T, K, L, iters, 0)
    np.testing.assert_allclose(got["loss_hist"], lr, rtol=1e-4)
    assert frob_rel(got["W"], Wr) < 1e-4
    assert frob_rel(got["H"], Hr) < 1e-4


@pytest.mark.gpu
def test_sharded_processes_fall_back_to_host_transport_together(oracle, tmp_path):
    """Two ranks on ONE GPU ask for the RCCL transport: RCCL refuses (two ranks of a communicator cannot share a device),
    every rank learns of it through the process group, and all of them rebuild their shard on the host-collective
    transport -- the agreement bench.py relies on at N > 1 so that a failed communicator costs speed, not the run."""
    N, T, K, L, iters = 65, 400, 5, 10, 4
    out = str(tmp_path / "res.npz")
    got = run_ranks(2, "hip", out, N, T, K, L, iters, 0, mode="iterate", transport="rccl", fallback=True)
    assert "transport=callbacks" in str(got["info"]) and "FALLBACK" in str(got["info"])
    _, _, _, Wr, Hr, lr = oracle_fit(oracle, N, T, K, L, iters, 0)
    np.testing.assert_allclose(got["loss_hist"], lr, rtol=1e-4)
    assert frob_rel(got["W"], Wr) < 1e-4 and frob_rel(got["H"], Hr) < 1e-4


@pytest.mark.gpu
def test_sharded_processes_asymmetric_rccl_failure_does_not_hang(oracle, tmp_path):
    """ONE rank cannot bind RCCL at all (CMF_RCCL_LIB=none makes its rccl_load fail before any collective): the two-stage
    agreement of ShardedMultUpdate._attach_rccl keeps every rank out of the blocking ncclCommInitRank -- rank 0 still
    broadcasts its id, the ready flags are exchanged, and all ranks take the host-collective transport together."""
    N, T, K, L, iters = 65, 400, 5, 10, 4
    for failing in (1, 0):  # a non-root rank, then the rank that creates the ncclUniqueId
        out = str(tmp_path / f"res{failing}.npz")
        got = run_ranks(2, "hip", out, N, T, K, L, iters, 0, mode="iterate", transport="rccl", fallback=True, timeout=240,
                        rank_env={failing: {"CMF_RCCL_LIB": "none"}})
        assert "transport=callbacks" in str(got["info"]) and "FALLBACK" in str(got["info"])
        _, _, _, Wr, Hr, lr = oracle_fit(oracle, N, T, K, L, iters, 0)
        np.testing.assert_allclose(got["loss_hist"], lr, rtol=1e-4)
        assert frob_rel(got["W"], Wr) < 1e-4 and frob_rel(got["H"], Hr) < 1e-4


@pytest.mark.gpu
def test_parameter_sweep_over_ranks(oracle, tmp_path):
    """SURVEY.md section 8f, f4 (src/model.jl:132-145): independent fits spread over the ranks of a process group
    (replicas: rank r on GPU r when the box has one per rank, otherwise the gloo ranks share GPU 0).  Every entry of the
    gathered sweep is compared with the fp64 oracle's fit of the same (L, K, alg) from the same seeds -- and with the
    single-process sweep, whose order and results it must reproduce bitwise."""
    import cmf_jl_amd as cmf

    out = str(tmp_path / "sweep.npz")
    port = _free_port()
    procs = []
    algs = (":mult", ":hals")
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_sweep_worker.py"), out, ",".join(algs)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=300)[0].decode(errors="replace") for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{logs[r][-3000:]}"
    got = np.load(out)
    data = cmf.gen_synthetic(N=40, T=300, seed=1234)
    ref = cmf.parameter_sweep(data, L_vals=(5, 8), K_vals=(2, 3), alg_vals=algs, max_itr=6, seed=0, check_convergence=False)
    tag = lambda L, K, a: f"{L}_{K}_{a.lstrip(':')}"  # noqa: E731
    assert list(got["keys"]) == [tag(L, K, a) for (L, K, a) in ref]  # the reference's insertion order (model.jl:137-142)
    for (L, K, a), r in ref.items():
        np.testing.assert_array_equal(got[f"loss_{tag(L, K, a)}"], r.loss_hist)  # same kernels, same seeds: bitwise
        np.testing.assert_array_equal(got[f"W_{tag(L, K, a)}"], r.W)
        # ... and the oracle: init_rand(seed 0) + 6 iterations of the rule, fp64
        W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=0)
        if a == ":mult":
            Wr, Hr, lr, _ = oracle.fit_mult(data, W0, H0, max_itr=6, check_convergence=False)
            tol = 1e-4
        else:
            Wr, Hr, lr, _ = oracle.c_fit_hals(data, W0, H0, max_itr=6, check_convergence=False)
            tol = 1e-4
        np.testing.assert_allclose(got[f"loss_{tag(L, K, a)}"], lr, rtol=1e-4)
        assert frob_rel(got[f"W_{tag(L, K, a)}"], Wr) < tol and frob_rel(got[f"H_{tag(L, K, a)}"], Hr) < tol


@pytest.mark.gpu
def test_sharded_full_size_matches_single_gpu(tmp_path):
    """BASELINE config 2 (N=2000, T=50000, K=32, L=20) split over 2 processes (25000 columns each, the shard of a
    2-GPU run; both share GPU 0 and the library's collectives go through gloo) against the unsharded rule."""
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


# ---------------------------------------------------------------------------------------------------------------
# GPU: BASELINE config 3 (N=2000, T=400000, K=32, L=20, 8 shards of 50000 columns)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_config3_shard_shape_middle_rank():
    """Three shards of 50000 columns on GPU 0 (T = 150000): the middle one is exactly a config-3 shard -- 50000 own
    columns, H halos on both sides, the right data halo -- compared with the unsharded rule at T = 150000."""
    import cmf_jl_amd as cmf

    N, T, K, L, iters = 2000, 150000, 32, 20, 3
    data = cmf.gen_synthetic(N=N, T=T, seed=1234)
    W0, H0 = cmf.init_rand(data, L=L, K=K, seed=0)
    rule = cmf.MultUpdate(data, W0, H0, devices=[0, 0, 0])
    assert rule.shard_bounds(1) == (50000, 100000)
    lg = [rule.compute_loss()] + list(rule.iterate(iters))
    Wg, Hg = rule.download()
    rule.close()
    rule = cmf.MultUpdate(data, W0, H0)
    ls = [rule.compute_loss()] + list(rule.iterate(iters))
    Ws, Hs = rule.download()
    rule.close()
    np.testing.assert_allclose(lg, ls, rtol=1e-5)
    assert frob_rel(Wg, Ws) < 1e-5 and frob_rel(Hg, Hs) < 1e-5
    assert all(b < a for a, b in zip(ls, ls[1:]))


@pytest.mark.gpu
def test_config3_full_T_single_gpu_properties():
    """N=2000, T=400000, K=32, L=20 on ONE GPU (what the 8 shards of config 3 add up to): takes every 32-bit offset
    guard of the kernels past T = 50000.  Size-independent properties: adjointness <conv(W,H), X> = <H, transconv(W,X)>
    through the update's own numerators (sum(H .* numH) = sum(W .* numW) = <conv(W,H), data>), monotone loss, and
    agreement of the first iterations with the 8-shard group on the same GPU."""
    import cmf_jl_amd as cmf

    N, T, K, L, iters = 2000, 400000, 32, 20, 2
    data = cmf.gen_synthetic(N=N, T=T, seed=1234)
    W0, H0 = cmf.init_rand(data, L=L, K=K, seed=0)
    rule = cmf.MultUpdate(data, W0, H0)
    ls = [rule.compute_loss()] + list(rule.iterate(iters))
    Ws, Hs = rule.download()
    rule.close()
    assert np.all(np.isfinite(ls)) and all(b < a for a, b in zip(ls, ls[1:]))
    assert Ws.min() > 0 and Hs.min() > 0
    # adjointness at full size on a thin slice of k (the stand-alone primitives go through the same kernels)
    rng = np.random.default_rng(0)
    Wt, Ht = rng.random((2, N, L)), rng.random((2, T))
    est = cmf.tensor_conv(Wt, Ht)
    lhs = float(np.vdot(est, data))
    rhs = float(np.vdot(Ht, cmf.tensor_transconv(Wt, data)))
    assert abs(lhs - rhs) <= 2e-5 * abs(lhs)
    del est
    # the same problem as 8 shards of 50000 columns (config 3's partition) on this one GPU
    rule = cmf.MultUpdate(data, W0, H0, devices=[0] * 8)
    assert rule.shard_bounds(7) == (350000, 400000)
    lg = [rule.compute_loss()] + list(rule.iterate(iters))
    Wg, Hg = rule.download()
    rule.close()
    np.testing.assert_allclose(lg, ls, rtol=1e-5)
    assert frob_rel(Wg, Ws) < 1e-5 and frob_rel(Hg, Hs) < 1e-5
