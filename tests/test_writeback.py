"""The reference's in-place semantics call by call: cmf_arm_writeback (include/cmf_hip.h).

CMF.jl's `fit` hands the SAME W and H arrays to every rule call and the rules mutate them (src/algs/alternating.jl:51-54;
mult.jl:37-38,51-52; hals.jl:110,153; pgd.jl:293), so the Julia binding writes the factors back after every
update_feature_maps!.  Since round 5 that write-back rides underneath the call's own kernels (a copy stream + helper threads);
what these tests pin: the caller's arrays are BIT FOR BIT what cmf_get_factors returns at that moment -- for every rule, for
K on the direct-copy path (a multiple of 32) and on the pack-kernel path, call after call -- and the losses do not change.
"""
import ctypes
import os

import numpy as np
import pytest

from cmf_jl_amd._lib import check  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cmf():
    import cmf_jl_amd as m

    lib = m.load_library()
    assert lib.cmf_device_count() >= 1, "no HIP device: the gpu tests need a real MI355X"
    return m


def problem(oracle, N, T, K, L, seed=3):
    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=min(L, 8), seed=seed)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=seed + 1)
    return data, W0, H0


def caller_arrays(W0, H0):
    # NaN-filled: an element the write-back missed cannot pass for a value
    return np.full(W0.shape, np.nan, order="F"), np.full(H0.shape, np.nan, order="F")


SHAPES = [(130, 700, 32, 20), (70, 257, 5, 10), (37, 150, 33, 7), (260, 600, 64, 20), (9, 40, 2, 3), (2000, 3000, 32, 20)]


@pytest.mark.parametrize("N,T,K,L", SHAPES)
def test_mu_writeback_is_bitwise_get_factors(cmf, oracle, N, T, K, L):
    data, W0, H0 = problem(oracle, N, T, K, L)
    ref = cmf.MultUpdate(data, W0, H0)
    rule = cmf.MultUpdate(data, W0, H0)
    rule.sync_every_call = True
    rule.verify_args = "none"  # (NaN-filled caller arrays: what is pinned here is the write-back, not the reading of arguments -- tests/test_dropin_contract.py)
    W, H = caller_arrays(W0, H0)
    try:
        for it in range(4):
            ref.update_motifs(l1W=0.1 * (it % 2))
            want = ref.update_feature_maps(l2H=0.05 * (it % 2))
            rule.update_motifs(data, W, H, l1W=0.1 * (it % 2))
            got = rule.update_feature_maps(data, W, H, l2H=0.05 * (it % 2))
            assert got == want  # the write-back changes nothing in the arithmetic
            Wd, Hd = rule.download()
            assert np.array_equal(W, Wd) and np.array_equal(H, Hd), f"iteration {it}"
            Wr, Hr = ref.download()
            assert np.array_equal(W, Wr) and np.array_equal(H, Hr)
        assert rule.counter("writeback_calls") == 4 and rule.counter("writeback_overlapped") == 4
    finally:
        ref.close()
        rule.close()


def test_writeback_of_one_factor_only_and_disarm(cmf, oracle):
    data, W0, H0 = problem(oracle, 60, 300, 32, 8)
    rule = cmf.MultUpdate(data, W0, H0)
    lib, h = rule._lib, rule._h
    pd = ctypes.POINTER(ctypes.c_double)
    W, H = caller_arrays(W0, H0)
    try:
        rule.update_motifs()
        check(lib.cmf_arm_writeback(h, None, H.ctypes.data_as(pd)))
        rule.update_feature_maps()
        Wd, Hd = rule.download()
        assert np.array_equal(H, Hd) and np.isnan(W).all()
        rule.update_motifs()
        check(lib.cmf_arm_writeback(h, W.ctypes.data_as(pd), None))
        rule.update_feature_maps()
        Wd, Hd2 = rule.download()
        assert np.array_equal(W, Wd) and np.array_equal(H, Hd)  # H untouched by the second call
        # armed, then disarmed: the next call must not touch the arrays
        W[...] = np.nan
        H[...] = np.nan
        rule.update_motifs()
        check(lib.cmf_arm_writeback(h, W.ctypes.data_as(pd), H.ctypes.data_as(pd)))
        check(lib.cmf_arm_writeback(h, None, None))
        rule.update_feature_maps()
        assert np.isnan(W).all() and np.isnan(H).all()
        # one arm serves one call
        check(lib.cmf_arm_writeback(h, W.ctypes.data_as(pd), H.ctypes.data_as(pd)))
        rule.update_feature_maps()
        assert not np.isnan(W).any()
        W[...] = np.nan
        rule.update_feature_maps()
        assert np.isnan(W).all()
    finally:
        rule.close()


@pytest.mark.parametrize("N,T,K,L", [(130, 700, 32, 20), (70, 257, 5, 10)])
def test_hals_and_pgd_writeback(cmf, oracle, N, T, K, L):
    data, W0, H0 = problem(oracle, N, T, K, L)
    for make, kw in ((cmf.HALSUpdate, {}), (cmf.PGDUpdate, {}), (cmf.PGDUpdate, {"constrH": cmf.UnitNormConstraint()})):
        ref = make(data, W0, H0)
        rule = make(data, W0, H0)
        rule.sync_every_call = True
        rule.verify_args = "none"  # (NaN-filled caller arrays: what is pinned here is the write-back, not the reading of arguments -- tests/test_dropin_contract.py)
        W, H = caller_arrays(W0, H0)
        try:
            for it in range(3):
                ref.update_motifs()
                want = ref.update_feature_maps(**kw)
                rule.update_motifs(data, W, H)
                got = rule.update_feature_maps(data, W, H, **kw)
                assert got == want
                Wd, Hd = rule.download()
                assert np.array_equal(W, Wd) and np.array_equal(H, Hd), (make.__name__, it)
            assert rule.counter("writeback_overlapped") == 3
        finally:
            ref.close()
            rule.close()


def test_gram_form_writeback(cmf, oracle):
    data, W0, H0 = problem(oracle, 130, 700, 32, 20)
    for gram in (1, 2):
        rule = cmf.MultUpdate(data, W0, H0)
        rule.set_option("gram", gram)
        rule.sync_every_call = True
        rule.verify_args = "none"  # (NaN-filled caller arrays: what is pinned here is the write-back, not the reading of arguments -- tests/test_dropin_contract.py)
        W, H = caller_arrays(W0, H0)
        try:
            for _ in range(2):
                rule.update_motifs(data, W, H)
                rule.update_feature_maps(data, W, H)
                Wd, Hd = rule.download()
                assert np.array_equal(W, Wd) and np.array_equal(H, Hd)
        finally:
            rule.close()


@pytest.mark.parametrize("transport,threads,gram", [(2, 1, 0), (3, 1, 0), (3, 0, 0), (4, 1, 0), (3, 1, 1)])
def test_group_handles_write_back_per_shard(cmf, oracle, transport, threads, gram):
    """A T-sharded group (the reference's `fit` driving several GPUs through one handle): every shard copies its own column
    block of H on its own copy stream behind its H update -- issued by whoever enqueues the shard: its worker thread, or the
    calling thread -- shard 0 also W, and the front handle's helpers widen block after block.  Loopback (one shared stream),
    a stream per shard with and without enqueue workers, the peer transport, and the Gram form: the caller's arrays are bit for
    bit cmf_get_factors, the losses those of an unarmed group."""
    data, W0, H0 = problem(oracle, 60, 700, 32, 8)
    ref = cmf.MultUpdate(data, W0, H0, devices=[0, 0, 0], transport=transport)
    rule = cmf.MultUpdate(data, W0, H0, devices=[0, 0, 0], transport=transport)
    for r in (ref, rule):
        if transport != 2:
            r.set_option("enqueue_threads", threads)
        r.set_option("gram", gram)
    rule.sync_every_call = True
    rule.verify_args = "none"  # (NaN-filled caller arrays: what is pinned here is the write-back, not the reading of arguments -- tests/test_dropin_contract.py)
    W, H = caller_arrays(W0, H0)
    try:
        for _ in range(3):
            ref.update_motifs()
            want = ref.update_feature_maps()
            rule.update_motifs(data, W, H)
            assert rule.update_feature_maps(data, W, H) == want
            Wd, Hd = rule.download()
            assert np.array_equal(W, Wd) and np.array_equal(H, Hd)
        assert rule.counter("writeback_calls") == 3 and rule.counter("writeback_overlapped") == 3
        # a batch in between leaves nothing armed behind, and the next armed call is right again
        rule.iterate(2)
        ref.iterate(2)
        rule.update_motifs(data, W, H)
        rule.update_feature_maps(data, W, H)
        Wd, Hd = rule.download()
        assert np.array_equal(W, Wd) and np.array_equal(H, Hd)
    finally:
        ref.close()
        rule.close()


@pytest.mark.parametrize("constr", [None, "unitnorm"])
def test_pgd_on_a_group_writes_back_per_shard(cmf, oracle, constr):
    """The PGD rule on a T-sharded group (pgd.jl:180-202 with H cut along T), with and without the UnitNorm rescaling that follows
    the step: the copies start behind the LAST kernel that writes a shard's H."""
    data, W0, H0 = problem(oracle, 60, 500, 32, 8)
    kw = {"constrH": cmf.UnitNormConstraint()} if constr else {}
    ref = cmf.PGDUpdate(data, W0, H0, devices=[0, 0, 0])
    rule = cmf.PGDUpdate(data, W0, H0, devices=[0, 0, 0])
    rule.sync_every_call = True
    rule.verify_args = "none"  # (NaN-filled caller arrays: what is pinned here is the write-back, not the reading of arguments -- tests/test_dropin_contract.py)
    W, H = caller_arrays(W0, H0)
    try:
        for _ in range(3):
            ref.update_motifs()
            want = ref.update_feature_maps(**kw)
            rule.update_motifs(data, W, H)
            assert rule.update_feature_maps(data, W, H, **kw) == want
            Wd, Hd = rule.download()
            assert np.array_equal(W, Wd) and np.array_equal(H, Hd)
        assert rule.counter("writeback_calls") == 3 and rule.counter("writeback_overlapped") == 3
    finally:
        ref.close()
        rule.close()


def test_hals_rerun_takes_h_again(cmf, oracle):
    """A persistent H pipeline whose wait ran out redoes the sweep from its snapshot after the first download of H has
    been taken (DESIGN.md HALS): the write-back must deliver the H of the redone sweep."""
    data, _, _ = oracle.c_gen_synthetic(N=40, T=600, K=3, L=8, seed=5)
    W0, H0 = oracle.c_init_rand(data, L=8, K=4, seed=2)
    rule = cmf.HALSUpdate(data, W0, H0)
    rule.sync_every_call = True
    rule.verify_args = "none"  # (NaN-filled caller arrays: what is pinned here is the write-back, not the reading of arguments -- tests/test_dropin_contract.py)
    W, H = caller_arrays(W0, H0)
    try:
        rule.update_motifs(data, W, H)
        os.environ["CMF_TEST_HOOKS"] = "1"
        try:
            rule.set_option("hals_debug", 3)
            rule.update_feature_maps(data, W, H)
            rule.set_option("hals_debug", 0)
        finally:
            os.environ.pop("CMF_TEST_HOOKS", None)
        assert rule.counter("hals_pipeline_reruns") == 1
        Wd, Hd = rule.download()
        assert np.array_equal(W, Wd) and np.array_equal(H, Hd)
    finally:
        rule.close()


def test_python_twin_refuses_arrays_it_cannot_hand_over(cmf, oracle):
    data, W0, H0 = problem(oracle, 20, 100, 4, 5)
    rule = cmf.MultUpdate(data, W0, H0)
    rule.sync_every_call = True
    rule.verify_args = "none"  # (NaN-filled caller arrays: what is pinned here is the write-back, not the reading of arguments -- tests/test_dropin_contract.py)
    try:
        rule.update_motifs()
        with pytest.raises(ValueError):
            rule.update_feature_maps(data, np.zeros(W0.shape, order="C"), np.zeros(H0.shape, order="F"))
        with pytest.raises(ValueError):
            rule.update_feature_maps(data, np.zeros(W0.shape, order="F"), np.zeros(H0.shape, dtype=np.float32, order="F"))
    finally:
        rule.close()


def test_armed_handle_can_be_destroyed_and_rearmed(cmf, oracle):
    """An arm that is never followed by its rule call must not leave anything behind: the handle is destroyed cleanly (the W
    copy in flight included), and arming again replaces the pointers."""
    data, W0, H0 = problem(oracle, 60, 300, 32, 8)
    pd = ctypes.POINTER(ctypes.c_double)
    for devices in (None, [0, 0]):
        rule = cmf.MultUpdate(data, W0, H0, devices=devices)
        W, H = caller_arrays(W0, H0)
        W2, H2 = caller_arrays(W0, H0)
        rule.update_motifs()
        check(rule._lib.cmf_arm_writeback(rule._h, W.ctypes.data_as(pd), H.ctypes.data_as(pd)))
        check(rule._lib.cmf_arm_writeback(rule._h, W2.ctypes.data_as(pd), H2.ctypes.data_as(pd)))  # the second arm wins
        rule.update_feature_maps()
        Wd, Hd = rule.download()
        assert np.array_equal(W2, Wd) and np.array_equal(H2, Hd) and np.isnan(W).all() and np.isnan(H).all()
        rule.update_motifs()
        check(rule._lib.cmf_arm_writeback(rule._h, W.ctypes.data_as(pd), H.ctypes.data_as(pd)))
        rule.close()  # armed, never served
        assert np.isnan(W).all() and np.isnan(H).all()


@pytest.mark.parametrize("devices", [None, [0, 0, 0]])
def test_an_arm_is_not_served_by_a_batch(cmf, oracle, devices):
    """cmf_arm_writeback serves the next *_update_feature_maps CALL (include/cmf_hip.h).  cmf_iterate / cmf_fit run H phases of
    their own: an arm left standing when they are called is dropped at their entry -- they write nothing into the caller's
    arrays (round 5 posted the helpers every iteration and nobody drained the last post) -- and the next armed rule call is
    right again."""
    data, W0, H0 = problem(oracle, 60, 700, 32, 8)
    rule = cmf.MultUpdate(data, W0, H0, devices=devices)
    ref = cmf.MultUpdate(data, W0, H0, devices=devices)
    pd = ctypes.POINTER(ctypes.c_double)
    W, H = caller_arrays(W0, H0)
    try:
        rule.update_motifs()
        ref.update_motifs()
        check(rule._lib.cmf_arm_writeback(rule._h, W.ctypes.data_as(pd), H.ctypes.data_as(pd)))
        got = rule.iterate(3)
        rule.synchronize()
        assert np.isnan(W).all() and np.isnan(H).all()  # untouched
        assert np.array_equal(got, ref.iterate(3))
        check(rule._lib.cmf_arm_writeback(rule._h, W.ctypes.data_as(pd), H.ctypes.data_as(pd)))
        n = 3
        lh, th, nh, ce = np.zeros(n + 1), np.zeros(n + 1), ctypes.c_int64(), ctypes.c_int()
        check(rule._lib.cmf_fit(rule._h, n, float("inf"), 1, 3, 0.0, 0, 0.0, 0.0, 0.0, 0.0, lh.ctypes.data_as(pd), th.ctypes.data_as(pd),
                                ctypes.byref(nh), ctypes.byref(ce)))  # the loop form (a stop test armed): its rule calls are not the caller's
        assert np.isnan(W).all() and np.isnan(H).all()
        rule.sync_every_call = True
        rule.verify_args = "none"  # (NaN-filled caller arrays: what is pinned here is the write-back, not the reading of arguments -- tests/test_dropin_contract.py)
        rule.update_motifs(data, W, H)
        rule.update_feature_maps(data, W, H)
        Wd, Hd = rule.download()
        assert np.array_equal(W, Wd) and np.array_equal(H, Hd)
    finally:
        rule.close()
        ref.close()
