"""The few-component kernels (csrc/cmf_small_k.h; K <= 16, L <= 64 -- the shapes the reference publishes on: README.md K = 5,
figures/fast_bcd/synthetic_comparison.jl:58-64 N = 250, K = 5, L = 20) against the fp64 oracle and against the general
kernels they replace (option small_k = 0): primitives, single iterations, whole fits, T-sharded groups, HALS and PGD on top
of them."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL_FACTORS = 1e-4
REL_LOSS = 1e-4
REL_PRIM = 2e-6


@pytest.fixture(scope="module")
def cmf():
    import cmf_jl_amd as m

    assert m.load_library().cmf_device_count() >= 1
    return m


def frob_rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


# (N, T, K, L): every k-pair count of conv_small_kernel (1, 2, 3, 4, 6, 8), J = L*K below / at / above one 128-row m group and
# above two, T shorter than a chunk / a strip / L, N crossing 32- and 128-column blocks, L at the strip limit; with small_k = 2 C3's
# reduction over n runs in 1 .. 8 pieces on these short recordings (2 slabs per piece), alone and with several row groups / VALU rows
SHAPES = [
    (48, 300, 4, 8), (7, 23, 3, 4), (5, 9, 1, 1), (1, 17, 2, 5), (6, 3, 2, 5), (4, 5, 3, 5), (70, 257, 5, 10),
    (250, 1500, 5, 20), (130, 700, 16, 20), (33, 400, 7, 19), (20, 200, 6, 40), (40, 333, 9, 15), (9, 1100, 2, 3),
    (65, 520, 11, 12), (300, 260, 13, 7), (17, 150, 16, 64), (129, 900, 15, 33), (10, 64, 8, 16), (500, 2000, 5, 10),
    # rows on the VALU (K*L = 32 m + 1 .. 4): 1, 2, 3, 4 rows behind 2, 2, 1, 1 MFMA blocks; N at / beyond what C3 keeps of Wj in LDS
    (60, 500, 5, 13), (31, 420, 2, 33), (90, 610, 5, 7), (45, 380, 3, 12), (1040, 200, 4, 25), (1100, 300, 4, 25), (50, 300, 11, 3),
    # recordings of several 64-row H strips per wave (the double-buffered LDS-DMA strips of C2): chunks of 5 and of 20 rounds
    (40, 9000, 4, 9), (36, 20000, 10, 6),
]


@pytest.mark.parametrize("N,T,K,L", SHAPES)
def test_single_iteration_small_k_vs_oracle_and_general_kernels(cmf, oracle, N, T, K, L):
    rng = np.random.default_rng(N * 7 + T)
    data = rng.random((N, T))
    W0 = np.asfortranarray(rng.random((K, N, L)))
    H0 = np.asfortranarray(rng.random((K, T)))
    out = {}
    for small in (2, 1, 0):  # 2: every few-component kernel; 1: the default (C3 on the general kernel when T is short); 0: none
        rule = cmf.MultUpdate(data, W0, H0)
        rule.set_option("small_k", small)
        l0 = rule.compute_loss()
        rule.update_motifs(l1W=0.1, l2W=0.5)
        l1 = rule.update_feature_maps(l1H=0.1, l2H=0.2)
        out[small] = (np.array([l0, l1]),) + rule.download()
        rule.close()
    W, H = W0.copy(), H0.copy()
    orule = oracle.MultUpdate(data, W, H)
    lo = [oracle.compute_loss(data, W, H)]
    oracle.update_motifs(orule, data, W, H, l1W=0.1, l2W=0.5)
    lo.append(oracle.update_feature_maps(orule, data, W, H, l1H=0.1, l2H=0.2))
    for small in (2, 1, 0):
        np.testing.assert_allclose(out[small][0], lo, rtol=1e-5, err_msg=f"small_k={small}")
        assert frob_rel(out[small][1], W) < 1e-5 and frob_rel(out[small][2], H) < 1e-5, f"small_k={small}"
    assert frob_rel(out[2][1], out[0][1]) < 5e-6 and frob_rel(out[2][2], out[0][2]) < 5e-6


@pytest.mark.parametrize("N,T,K,L,iters,reg", [(250, 4000, 5, 20, 40, 0), (500, 2000, 5, 10, 100, 1), (120, 900, 12, 25, 30, 1), (64, 700, 16, 8, 30, 0)])
def test_fit_small_k(cmf, oracle, N, T, K, L, iters, reg):
    """Whole fits (the reference's own protocol shape at a fifth of its T; BASELINE configs[0] with the README regularisers):
    W, H, loss_hist within the north star's 1e-4 of the fp64 oracle, pipelined (cmf_iterate) and call by call."""
    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=min(L, 20), seed=1234)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=0)
    kw = dict(l1W=0.1, l2W=0.5, l1H=0.1, l2H=0.2) if reg else dict(l1W=0.0, l2W=0.0, l1H=0.0, l2H=0.0)
    Wr, Hr, lr, _ = oracle.fit_mult(data, W0, H0, max_itr=iters, check_convergence=False, **kw)
    rule = cmf.MultUpdate(data, W0, H0)
    rule.set_option("small_k", 2)
    lg = [rule.compute_loss()] + list(rule.iterate(iters, **kw))
    Wg, Hg = rule.download()
    rule.upload(W0, H0)
    lc = [rule.compute_loss()]
    for _ in range(min(iters, 5)):
        rule.update_motifs(l1W=kw["l1W"], l2W=kw["l2W"])
        lc.append(rule.update_feature_maps(l1H=kw["l1H"], l2H=kw["l2H"]))
    rule.close()
    np.testing.assert_allclose(lg, lr, rtol=REL_LOSS)
    np.testing.assert_array_equal(lc, lg[: len(lc)])  # the pipelined batch is the call-by-call loop, bit for bit
    assert frob_rel(Wg, Wr) < REL_FACTORS and frob_rel(Hg, Hr) < REL_FACTORS
    assert np.all(Wg >= cmf.EPSILON) and np.all(Hg >= cmf.EPSILON)


@pytest.mark.parametrize("R,N,T,K,L", [(2, 250, 1500, 5, 20), (3, 40, 333, 5, 10), (4, 70, 600, 12, 16), (8, 33, 1100, 3, 7)])
def test_groups_on_the_small_k_kernels(cmf, oracle, R, N, T, K, L):
    """T-sharded groups (halos on both sides of the middle shards) on the few-component kernels: every transport a one-GPU box
    has, plain and overlap form, against the oracle and bitwise among themselves."""
    from cmf_jl_amd import _lib

    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=min(L, 20), seed=1234)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=0)
    Wr, Hr, lr, _ = oracle.fit_mult(data, W0, H0, max_itr=6, check_convergence=False)
    res = []
    for tr in (_lib.CMF_COMM_LOOPBACK, _lib.CMF_COMM_PEER):
        for overlap in (False, True):
            rule = cmf.MultUpdate(data, W0, H0, devices=[0] * R, transport=tr)
            rule.set_overlap(overlap)
            ls = [rule.compute_loss()] + list(rule.iterate(6))
            res.append((overlap, np.asarray(ls)) + rule.download())
            rule.close()
            np.testing.assert_allclose(ls, lr, rtol=REL_LOSS)
            assert frob_rel(res[-1][2], Wr) < REL_FACTORS and frob_rel(res[-1][3], Hr) < REL_FACTORS
    for r in res[2:]:
        base = res[0] if not r[0] else res[1]
        for a, b in zip(base[1:], r[1:]):
            np.testing.assert_array_equal(a, b)


def test_group_shards_update_h_inside_their_c3_launches(cmf):
    """Shards long enough for the fused form (the reduction over n uncut): each shard's C3 launch updates its own columns of H --
    halos on either side, the all-gather of the new halos behind it -- with the same bits as with the separate launch, on every
    transport of a one-GPU box; and the group agrees with ONE handle to rounding."""
    from cmf_jl_amd import _lib

    N, T, K, L = 120, 76000, 5, 20
    rng = np.random.default_rng(9)
    data = rng.random((N, T))
    W0 = np.asfortranarray(rng.random((K, N, L)))
    H0 = np.asfortranarray(rng.random((K, T)))
    one = cmf.MultUpdate(data, W0, H0)
    l1 = one.iterate(4, l1H=0.05)
    W1, H1 = one.download()
    one.close()
    for tr in (_lib.CMF_COMM_LOOPBACK, _lib.CMF_COMM_PEER):
        out = []
        for fuse in (1, 0):
            rule = cmf.MultUpdate(data, W0, H0, devices=[0, 0], transport=tr)
            rule.set_option("small_k_fuse", fuse)
            ls = rule.iterate(4, l1H=0.05)
            assert rule.counter("small_k_fused_h_updates") == (8 if fuse else 0)  # 2 shards x 4 iterations
            out.append((np.asarray(ls),) + rule.download())
            rule.close()
        for a, b in zip(*out):
            assert np.array_equal(a, b)
        np.testing.assert_allclose(out[0][0], l1, rtol=1e-6)
        assert frob_rel(out[0][1], W1) < 1e-6 and frob_rel(out[0][2], H1) < 1e-6


def test_primitives_and_other_rules_with_few_components(cmf, oracle):
    """tensor_conv / tensor_transconv stand-alone, and the HALS and PGD rules (which share the conv / C2 / C3 launchers) at
    K = 5: against the oracle."""
    N, T, K, L = 90, 640, 5, 12
    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=12, seed=3)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=1)
    assert frob_rel(cmf.tensor_conv(W0, H0), oracle.tensor_conv(W0, H0)) < REL_PRIM
    assert frob_rel(cmf.tensor_transconv(W0, data), oracle.tensor_transconv(W0, data)) < REL_PRIM
    for alg, fit in ((":hals", oracle.c_fit_hals), (":pgd", None)):
        res = cmf.fit_cnmf(data, L=L, K=K, alg=alg, max_itr=8, check_convergence=False, W_init=W0, H_init=H0)
        if fit is None:
            Wr, Hr, lr, _ = oracle.fit_pgd(data, W0, H0, max_itr=8)
        else:
            Wr, Hr, lr, _ = fit(data, W0, H0, max_itr=8, check_convergence=False)
        np.testing.assert_allclose(res.loss_hist, lr, rtol=REL_LOSS, err_msg=alg)
        assert frob_rel(res.W, Wr) < REL_FACTORS and frob_rel(res.H, Hr) < REL_FACTORS, alg


def test_reference_protocol_shape_full_size(cmf, oracle):
    """figures/fast_bcd/synthetic_comparison.jl:58-64 at full size (N = 250, T = 50000, K = 5, L = 20): 5 iterations against
    the oracle, and the few-component kernels against the general ones."""
    N, T, K, L = 250, 50000, 5, 20
    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=20, seed=1234)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=0)
    Wr, Hr, lr, _ = oracle.fit_mult(data, W0, H0, max_itr=5, check_convergence=False)
    out = {}
    for small in (1, 0):
        rule = cmf.MultUpdate(data, W0, H0)
        rule.set_option("small_k", small)
        ls = [rule.compute_loss()] + list(rule.iterate(5))
        out[small] = (np.asarray(ls),) + rule.download()
        rule.close()
        np.testing.assert_allclose(ls, lr, rtol=REL_LOSS)
        assert frob_rel(out[small][1], Wr) < REL_FACTORS and frob_rel(out[small][2], Hr) < REL_FACTORS
    print("small_k vs general kernels: relW", frob_rel(out[1][1], out[0][1]), "relH", frob_rel(out[1][2], out[0][2]))


def test_every_shape_with_every_buffer_an_allocation_of_its_own(cmf):
    """The small buffers of a handle (H, W, slabs, numerators, halos, partials: about twenty) live in ONE allocation at 256-byte
    granules, so an overrun of one -- the few-component kernels have slab spills, VALU rows and quarter-piece partial indices to get
    wrong -- would land in a live neighbour and could pass unnoticed.  Test hook CMF_TEST_NO_ARENA=1 gives every buffer its own
    allocation: the same iterations must give bit for bit the same factors and losses as with the arena (what ran above)."""
    import os

    for (N, T, K, L) in SHAPES:
        rng = np.random.default_rng(N * 7 + T)
        data = rng.random((N, T))
        W0 = np.asfortranarray(rng.random((K, N, L)))
        H0 = np.asfortranarray(rng.random((K, T)))
        out = []
        for no_arena in (False, True):
            if no_arena:
                os.environ.update(CMF_TEST_HOOKS="1", CMF_TEST_NO_ARENA="1")
            try:
                rule = cmf.MultUpdate(data, W0, H0)
            finally:
                os.environ.pop("CMF_TEST_HOOKS", None)
                os.environ.pop("CMF_TEST_NO_ARENA", None)
            rule.set_option("small_k", 2)
            ls = rule.iterate(2, l1W=0.1, l2W=0.5, l1H=0.1, l2H=0.2)
            out.append((ls,) + rule.download())
            rule.close()
        for a, b in zip(*out):
            assert np.array_equal(a, b), (N, T, K, L)


def test_h_update_inside_the_c3_launch_is_bitwise_the_separate_launch(cmf):
    """Option small_k_fuse (default 1): the workgroup whose ticket completes a 128-column block's slabs (its own tile's and the next
    tile's spill, every piece of the reduction over n) runs mult.jl:51-52 on that block inside g_gemm_fold_small_kernel -- sc1 slab
    stores, drained, agent-scope tickets, an acquire and sc1 loads on the updating side -- with h_update_kernel's summation order: the
    same bits as the launch of its own, on every shape (1 .. 8 pieces, T below / across 128-column blocks, the last block empty),
    iteration after iteration (the counters must be back at zero), also while another handle keeps the chip unevenly busy."""
    busy_data = np.random.default_rng(5).random((300, 30000))
    busy = cmf.MultUpdate(busy_data, *cmf.init_rand(busy_data, L=12, K=32, seed=3))
    fused_somewhere = 0
    try:
        for (N, T, K, L) in SHAPES + [(250, 50000, 5, 20), (64, 128 * 7 - 3, 5, 4), (64, 128 * 7 + 1, 8, 2)]:
            rng = np.random.default_rng(N * 7 + T)
            data = rng.random((N, T))
            W0 = np.asfortranarray(rng.random((K, N, L)))
            H0 = np.asfortranarray(rng.random((K, T)))
            out = []
            for fuse in (1, 0):
                rule = cmf.MultUpdate(data, W0, H0)
                rule.set_option("small_k", 2)
                rule.set_option("small_k_fuse", 2 * fuse)  # (2: also where the reduction over n is cut into pieces)
                ls = []
                for rep in range(3):
                    if fuse:
                        busy.update_motifs()  # (asynchronous: the other handle's launches run beside this one's)
                    ls += list(rule.iterate(2, l1W=0.1, l2W=0.5, l1H=0.1, l2H=0.2))
                    rule.update_motifs(l1W=0.1)
                    ls.append(rule.update_feature_maps(l2H=0.3))
                ls += list(rule.iterate(2, eval_mode=True))  # H updates in a row
                out.append((np.array(ls),) + rule.download())
                if fuse:
                    fused_somewhere += rule.counter("small_k_fused_h_updates") > 0
                rule.close()
            for a, b in zip(*out):
                assert np.array_equal(a, b), (N, T, K, L)
    finally:
        busy.close()
    assert fused_somewhere >= 20  # (shapes with more than one row group of components keep the separate launch)
