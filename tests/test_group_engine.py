"""The group engine of round 4 (csrc/cmf_groups.hip) on what a one-GPU box can run:

* enqueue workers -- one thread per shard issuing that shard's kernels and collective calls -- against the calling thread
  enqueueing every shard itself: bitwise the same results, call by call and as pipelined cmf_iterate batches, plain /
  overlap / Gram forms, on every transport that has a stream per shard (loopback-streams, peer);
* the peer transport (direct reads and writes of the other shards' buffers between event fences; the xGMI form of the
  all-reduce) rehearsed with all shards on one device: bitwise the loopback transport (both sum in rank order);
* the overlap form's second communicator on the one RCCL communicator a single device can form;
* errors raised inside a worker reach the caller, and the handle survives.

RCCL across distinct devices and peer access over xGMI are NOT covered here (tests/test_multi_gpu.py, device-count gated).
"""
import os

import numpy as np
import pytest

from test_sharded import REG, oracle_fit

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cmf():
    import cmf_jl_amd

    if cmf_jl_amd.load_library().cmf_device_count() < 1:
        pytest.skip("no HIP device")
    return cmf_jl_amd


def _mu(rule, mode, iters, kw):
    losses = [rule.compute_loss()]
    if mode == "calls":
        for _ in range(iters):
            rule.update_motifs(l1W=kw["l1W"], l2W=kw["l2W"])
            losses.append(rule.update_feature_maps(l1H=kw["l1H"], l2H=kw["l2H"]))
    else:
        losses += list(rule.iterate(iters, **kw))
    W, H = rule.download()
    return np.asarray(losses), W, H


def _info(rule):
    return dict(tok.split("=", 1) for tok in rule.comm_info().split() if "=" in tok)


@pytest.mark.parametrize("R,N,T,K,L,reg,overlap,gram", [(2, 130, 900, 32, 20, 0, False, 0), (4, 70, 517, 32, 20, 1, True, 0),
                                                         (3, 40, 333, 5, 10, 1, False, 1), (8, 96, 1100, 32, 20, 0, True, 1),
                                                         (8, 96, 700, 32, 20, 0, False, 0), (5, 33, 420, 7, 12, 1, True, 0)])
def test_enqueue_workers_and_peer_transport_are_bitwise_the_loopback_group(cmf, oracle, R, N, T, K, L, reg, overlap, gram):
    from cmf_jl_amd import _lib

    iters = 6
    data, W0, H0, Wr, Hr, lr = oracle_fit(oracle, N, T, K, L, iters, reg)
    kw = REG if reg else dict(l1W=0.0, l2W=0.0, l1H=0.0, l2H=0.0)
    ref = {}
    for mode in ("calls", "iterate"):
        rule = cmf.MultUpdate(data, W0, H0, devices=[0] * R, transport=_lib.CMF_COMM_LOOPBACK)
        assert _info(rule)["enqueue"] == "caller"  # one shared stream: nothing to hand out
        if gram:
            rule.set_option("gram", 1)
        rule.set_overlap(overlap)
        ref[mode] = _mu(rule, mode, iters, kw)
        rule.close()
        np.testing.assert_allclose(ref[mode][0], lr, rtol=1e-4)
    for tr, name in ((_lib.CMF_COMM_LOOPBACK_STREAMS, "loopback-streams"), (_lib.CMF_COMM_PEER, "peer")):
        for threads in (1, 0):
            for mode in ("calls", "iterate"):
                rule = cmf.MultUpdate(data, W0, H0, devices=[0] * R, transport=tr)
                assert _info(rule)["transport"] == name and _info(rule)["enqueue"] == "threads"  # the default
                if not threads:
                    rule.set_option("enqueue_threads", 0)
                    assert _info(rule)["enqueue"] == "caller"
                if gram:
                    rule.set_option("gram", 1)
                rule.set_overlap(overlap)
                got = _mu(rule, mode, iters, kw)
                rule.close()
                for a, b in zip(ref[mode], got):
                    np.testing.assert_array_equal(a, b, err_msg=f"{name} threads={threads} {mode}")


def test_switching_the_enqueue_form_in_the_middle_of_a_fit(cmf, oracle):
    """Workers can be stopped and started between calls; the fit goes on bit for bit."""
    from cmf_jl_amd import _lib

    data, W0, H0, Wr, Hr, lr = oracle_fit(oracle, 70, 517, 32, 20, 9, 1)
    base = cmf.MultUpdate(data, W0, H0, devices=[0] * 4, transport=_lib.CMF_COMM_LOOPBACK)
    want = _mu(base, "iterate", 9, REG)
    base.close()
    rule = cmf.MultUpdate(data, W0, H0, devices=[0] * 4, transport=_lib.CMF_COMM_PEER)
    losses = [rule.compute_loss()] + list(rule.iterate(3, **REG))
    rule.set_option("enqueue_threads", 0)
    losses += list(rule.iterate(3, **REG))
    rule.set_option("enqueue_threads", 1)
    for _ in range(3):
        rule.update_motifs(l1W=REG["l1W"], l2W=REG["l2W"])
        losses.append(rule.update_feature_maps(l1H=REG["l1H"], l2H=REG["l2H"]))
    W, H = rule.download()
    rule.close()
    np.testing.assert_array_equal(want[0], np.asarray(losses))
    np.testing.assert_array_equal(want[1], W)
    np.testing.assert_array_equal(want[2], H)
    np.testing.assert_allclose(losses, lr, rtol=1e-4)


@pytest.mark.parametrize("tr", ["streams", "peer"])
def test_pgd_on_groups_with_workers_present(cmf, oracle, tr):
    """The PGD entries enqueue from the calling thread (GroupInline) whatever the group's enqueue form is; MU phases posted
    to the workers before and after must not interleave with them."""
    from cmf_jl_amd import _lib

    data, _, _ = oracle.c_gen_synthetic(N=130, T=900, K=3, L=20, seed=1234)
    W0, H0 = oracle.c_init_rand(data, L=20, K=32, seed=0)
    rule = cmf.PGDUpdate(data, W0, H0, devices=[0] * 4, transport=_lib.CMF_COMM_PEER if tr == "peer" else _lib.CMF_COMM_LOOPBACK_STREAMS)
    assert _info(rule)["enqueue"] == "threads"
    lg = []
    for _ in range(5):
        rule.update_motifs()
        lg.append(rule.update_feature_maps())
    Wg, Hg = rule.download()
    rule.close()
    Wr, Hr, lr, _ = oracle.fit_pgd(data, W0, H0, max_itr=5)
    np.testing.assert_allclose(lg, lr[1:], rtol=1e-4)
    assert np.linalg.norm(Wg - Wr) < 1e-4 * np.linalg.norm(Wr) and np.linalg.norm(Hg - Hr) < 1e-4 * np.linalg.norm(Hr)


def test_config2_eight_shards_workers_and_peer(cmf):
    """BASELINE config 2 as the 8 shards of `bench.py --gpus 8` (all on GPU 0), 10 pipelined iterations: the peer
    transport with enqueue workers, plain and overlap form, bitwise the shared-stream loopback group."""
    from cmf_jl_amd import _lib

    data = cmf.gen_synthetic(N=2000, T=50000, seed=1234)
    W0, H0 = cmf.init_rand(data, L=20, K=32, seed=0)
    out = {}
    for tr in (_lib.CMF_COMM_LOOPBACK, _lib.CMF_COMM_PEER, _lib.CMF_COMM_LOOPBACK_STREAMS):
        for overlap in (False, True):
            rule = cmf.MultUpdate(data, W0, H0, devices=[0] * 8, transport=tr)
            rule.set_overlap(overlap)
            ls = np.asarray(rule.iterate(10))
            out[(tr, overlap)] = (ls,) + rule.download()
            rule.close()
    for overlap in (False, True):
        a = out[(_lib.CMF_COMM_LOOPBACK, overlap)]
        for tr in (_lib.CMF_COMM_PEER, _lib.CMF_COMM_LOOPBACK_STREAMS):
            for x, y in zip(a, out[(tr, overlap)]):
                np.testing.assert_array_equal(x, y)


def test_overlap_form_has_a_communicator_of_its_own_on_rccl(cmf, oracle):
    """A 1-device RCCL group (the one communicator a one-GPU box can form): switching the overlap form on creates the
    communication stream's own communicator (lanes=2 in cmf_comm_info), both are used, results equal the loopback group's."""
    from cmf_jl_amd import _lib

    data, W0, H0, Wr, Hr, lr = oracle_fit(oracle, 70, 517, 32, 20, 6, 0)
    kw = dict(l1W=0.0, l2W=0.0, l1H=0.0, l2H=0.0)
    try:
        rule = cmf.MultUpdate(data, W0, H0, devices=[0], transport=_lib.CMF_COMM_RCCL)
    except cmf.CMFError as e:
        pytest.skip(f"no RCCL communicator on this box: {e}")
    assert _info(rule)["transport"] == "rccl" and _info(rule)["lanes"] == "1"
    rule.set_overlap(True)
    assert _info(rule)["lanes"] == "2" and _info(rule)["overlap"] == "1"
    got = _mu(rule, "iterate", 6, kw)
    ms, nbytes = rule.time_kernel("allreduce_lane1", reps=3)
    assert ms > 0 and nbytes > 0
    rule.set_overlap(False)  # back to the single-stream form: the second communicator stays, unused
    more = list(rule.iterate(2, **kw))
    rule.close()
    np.testing.assert_allclose(got[0], lr, rtol=1e-4)
    assert more[-1] < got[0][-1]
    base = cmf.MultUpdate(data, W0, H0, devices=[0], transport=_lib.CMF_COMM_LOOPBACK)
    base.set_overlap(True)
    want = _mu(base, "iterate", 6, kw)
    base.close()
    for a, b in zip(want, got):  # one rank: the all-reduce is the identity
        np.testing.assert_array_equal(a, b)


def test_collective_timers_of_the_bench_line(cmf, oracle):
    """cmf_time_kernel's collective timers (bench.py's comm block) on a loopback-streams group: payload sizes as DESIGN.md
    states them, and the group goes on iterating afterwards with unchanged results."""
    from cmf_jl_amd import _lib

    data, W0, H0, Wr, Hr, lr = oracle_fit(oracle, 96, 700, 32, 20, 4, 0)
    kw = dict(l1W=0.0, l2W=0.0, l1H=0.0, l2H=0.0)
    rule = cmf.MultUpdate(data, W0, H0, devices=[0] * 4, transport=_lib.CMF_COMM_LOOPBACK_STREAMS)
    a = list(rule.iterate(2, **kw))
    Np, K32, L = 128, 32, 20
    ms, b = rule.time_kernel("allreduce", reps=2)
    assert b == 4 * (2 * L * K32 * Np + 64)
    ms, b = rule.time_kernel("allreduce_gram", reps=2)
    assert b == 4 * (L * K32 * Np + (L * K32) * 640 + 64)
    ms, b = rule.time_kernel("allgather_halo", reps=2)
    assert b == 4 * 2 * (L - 1) * K32
    a += list(rule.iterate(2, **kw))
    rule.close()
    np.testing.assert_allclose(a, lr[1:], rtol=1e-4)


def test_worker_errors_reach_the_caller(cmf, oracle):
    """A failure inside an enqueue worker's job surfaces as the error of the public call (with the worker's message), and
    later calls on the handle work: here the overlap form on an RCCL group whose second communicator was never made is
    provoked through the library's test hook."""
    from cmf_jl_amd import _lib

    data, W0, H0, Wr, Hr, lr = oracle_fit(oracle, 70, 517, 32, 20, 3, 0)
    rule = cmf.MultUpdate(data, W0, H0, devices=[0] * 3, transport=_lib.CMF_COMM_PEER)
    os.environ["CMF_TEST_HOOKS"] = "1"
    os.environ["CMF_TEST_FAIL_SHARD"] = "1"  # shard 1's next collective call fails
    try:
        with pytest.raises(cmf.CMFError) as ei:
            rule.update_motifs()
        assert "CMF_TEST_FAIL_SHARD" in str(ei.value)
    finally:
        os.environ.pop("CMF_TEST_FAIL_SHARD", None)
        os.environ.pop("CMF_TEST_HOOKS", None)
    # the group is usable again: factors are re-set (a phase that failed half-way leaves shards out of step) and the fit runs
    rule.upload(W0, H0)
    ls = [rule.compute_loss()] + list(rule.iterate(3))
    rule.close()
    np.testing.assert_allclose(ls, lr, rtol=1e-4)


def test_rccl_calls_from_an_enqueue_worker_thread(cmf, oracle):
    """What a one-GPU box can show of the worker form on RCCL: a 1-device RCCL group whose single shard is enqueued by a
    worker thread (test hook), i.e. ncclAllReduce / ncclAllGather -- on both lanes with the overlap form -- issued from a
    thread other than the one that created the communicators, through the job queue, with the pinned loss words polled by
    the caller: results bitwise those of the same group enqueued by the calling thread."""
    from cmf_jl_amd import _lib

    data, W0, H0, Wr, Hr, lr = oracle_fit(oracle, 70, 517, 32, 20, 6, 0)
    kw = dict(l1W=0.0, l2W=0.0, l1H=0.0, l2H=0.0)
    res = {}
    for forced in (0, 1):
        os.environ["CMF_TEST_HOOKS"] = "1"
        os.environ["CMF_TEST_FORCE_WORKERS"] = str(forced)
        try:
            try:
                rule = cmf.MultUpdate(data, W0, H0, devices=[0], transport=_lib.CMF_COMM_RCCL)
            except cmf.CMFError as e:
                pytest.skip(f"no RCCL communicator on this box: {e}")
            assert _info(rule)["enqueue"] == ("threads" if forced else "caller")
            for overlap in (False, True):
                rule.upload(W0, H0)
                rule.set_overlap(overlap)
                res[(forced, overlap)] = _mu(rule, "iterate", 6, kw) + (_mu(rule, "calls", 2, kw)[0],)
            rule.close()
        finally:
            os.environ.pop("CMF_TEST_HOOKS", None)
            os.environ.pop("CMF_TEST_FORCE_WORKERS", None)
    for overlap in (False, True):
        np.testing.assert_allclose(res[(1, overlap)][0], lr, rtol=1e-4)
        for a, b in zip(res[(0, overlap)], res[(1, overlap)]):
            np.testing.assert_array_equal(a, b)
