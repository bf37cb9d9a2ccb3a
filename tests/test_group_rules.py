"""T-sharded groups beyond the plain MU rule (SURVEY.md section 8e widened per section 8f): the Gram form of the MU
iteration ([numW | HH | tail] all-reduce) and the PGD rule (src/algs/pgd.jl:158-255; the rule the reference's long
recordings are fitted with, notebooks/test_mouse.ipynb) on group handles -- loopback groups of 2-8 shards on GPU 0 and
gloo ranks over the host-callback transport, against the fp64 oracle."""
import numpy as np
import pytest

from test_sharded import REG, frob_rel, oracle_fit, run_ranks

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


# ---- Gram form on groups ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("R,N,T,K,L,reg,overlap", [(2, 130, 900, 32, 20, 0, False), (3, 40, 333, 5, 10, 1, False), (4, 70, 517, 32, 20, 1, True),
                                                    (2, 65, 300, 64, 33, 0, False), (8, 96, 1100, 32, 20, 1, False), (3, 130, 900, 32, 20, 0, True),
                                                    (1, 48, 300, 4, 8, 1, False)])
def test_group_gram_form_against_oracle(cmf, oracle, R, N, T, K, L, reg, overlap):
    """option gram = 1 on a group: every shard contracts numW and its share of HH = H_unfold H_unfold' (the lag
    correlations of its own columns with the left H halo; the cut terms of the right end on the last shard only), the
    all-reduce carries [numW | HH | tail], denomW = HH W on every shard, denomH from the lag-Gram taps with both H halos.
    Against the oracle at the north star's bar, and against the unsharded Gram form (same arithmetic, other sum order)."""
    iters = 6
    data, W0, H0, Wr, Hr, lr = oracle_fit(oracle, N, T, K, L, iters, reg)
    kw = REG if reg else dict(l1W=0.0, l2W=0.0, l1H=0.0, l2H=0.0)
    single = cmf.MultUpdate(data, W0, H0)
    single.set_option("gram", 1)
    ls, Ws, Hs = _mu(single, "calls", iters, kw)
    single.close()
    got = {}
    for mode in ("calls", "iterate"):
        rule = cmf.MultUpdate(data, W0, H0, devices=[0] * R)
        rule.set_option("gram", 1)
        if overlap:
            rule.set_overlap(True)
        got[mode] = _mu(rule, mode, iters, kw)
        # switching the form off again on the same handle gives the reference formulation back
        rule.set_option("gram", 0)
        rule.upload(W0, H0)
        lb, Wb, Hb = _mu(rule, mode, 2, kw)
        rule.close()
        np.testing.assert_allclose(lb, lr[:3], rtol=1e-4)
        losses, W, H = got[mode]
        np.testing.assert_allclose(losses, lr, rtol=1e-4)
        assert frob_rel(W, Wr) < 1e-4 and frob_rel(H, Hr) < 1e-4
        np.testing.assert_allclose(losses, ls, rtol=2e-5)
        assert frob_rel(W, Ws) < 2e-5 and frob_rel(H, Hs) < 2e-5
    np.testing.assert_array_equal(got["calls"][1], got["iterate"][1])
    np.testing.assert_array_equal(got["calls"][2], got["iterate"][2])


def test_group_gram_refusals(cmf):
    from cmf_jl_amd import _lib

    data = np.random.default_rng(0).random((8, 64))
    W, H = np.ones((2, 8, 20)), np.ones((2, 64))
    rule = cmf.MultUpdate(data, W, H, devices=[0, 0])
    with pytest.raises(cmf.CMFError) as ei:  # T < 4 L: the cut terms of HH would cancel most of the correlations
        rule.set_option("gram", 1)
    assert ei.value.code == _lib.CMF_ERR_UNSUPPORTED
    with pytest.raises(cmf.CMFError):
        rule.set_option("gram", 2)  # the Gram-sum loss is outside the 1e-4 bar: unsharded handles only
    rule.close()


# ---- PGD on groups ---------------------------------------------------------------------------------------------------
def _pgd_problem(oracle, N, T, K, L, seed=1234):
    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=min(L, 20), seed=seed)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=0)
    return data, W0, H0


@pytest.mark.parametrize("R,N,T,K,L", [(2, 48, 300, 4, 8), (3, 130, 900, 32, 20), (4, 37, 250, 33, 7), (8, 96, 1100, 32, 20), (1, 60, 200, 5, 10)])
def test_group_pgd_default_rule(cmf, oracle, R, N, T, K, L):
    """Default PGDUpdate (SquareLoss, NonnegConstraint, penaltiesW=[SquarePenalty(1)]) on R shards: losses, factors and the
    accept / reject decisions of the step-size state machine equal the unsharded oracle's."""
    data, W0, H0 = _pgd_problem(oracle, N, T, K, L)
    iters = 8
    rule = cmf.PGDUpdate(data, W0, H0, devices=[0] * R)
    lg = [rule.compute_loss()]
    for _ in range(iters):
        rule.update_motifs()
        lg.append(rule.update_feature_maps())
    Wg, Hg = rule.download()
    sg = rule.steps
    rule.close()
    Wr, Hr, lr, sr = oracle.fit_pgd(data, W0, H0, max_itr=iters)
    np.testing.assert_allclose(lg, lr, rtol=1e-4)
    assert frob_rel(Wg, Wr) < 1e-4 and frob_rel(Hg, Hr) < 1e-4
    np.testing.assert_allclose(sg, sr, rtol=1e-12)
    # the public entry: fit_cnmf(alg=:pgd, devices=[...])
    res = cmf.fit_cnmf(data, L=L, K=K, alg=":pgd", max_itr=iters, check_convergence=False, W_init=W0, H_init=H0, devices=[0] * R)
    np.testing.assert_allclose(res.loss_hist, lr, rtol=1e-4)
    assert frob_rel(res.W, Wr) < 1e-4 and frob_rel(res.H, Hr) < 1e-4


@pytest.mark.parametrize("R,constrW,constrH", [(3, "unitnorm", "nonneg"), (2, "nonneg", "unitnorm"), (4, "unitnorm", "unitnorm")])
def test_group_pgd_unit_norm_and_penalties(cmf, oracle, R, constrW, constrH):
    """UnitNormConstraint on a sharded H needs the component norms over ALL of T (summed over the shards in rank order);
    penalties and the unconstrained step are replicated arithmetic."""
    N, T, K, L = 60, 500, 5, 10
    data, W0, H0 = _pgd_problem(oracle, N, T, K, L, seed=7)
    cls = {"unitnorm": cmf.UnitNormConstraint(), "nonneg": cmf.NonnegConstraint()}
    rule = cmf.PGDUpdate(data, W0, H0, devices=[0] * R)
    lg = []
    for _ in range(6):
        rule.update_motifs(constrW=cls[constrW])
        lg.append(rule.update_feature_maps(constrH=cls[constrH]))
    Wg, Hg = rule.download()
    sg = rule.steps
    rule.close()
    Wr, Hr, lr, sr = oracle.fit_pgd(data, W0, H0, max_itr=6, constrW=constrW, constrH=constrH)
    np.testing.assert_allclose(lg, lr[1:], rtol=1e-4)
    assert frob_rel(Wg, Wr) < 1e-4 and frob_rel(Hg, Hr) < 1e-4
    np.testing.assert_allclose(sg, sr, rtol=1e-12)
    # penalties + no constraint on W
    rule = cmf.PGDUpdate(data, W0, H0, devices=[0] * R)
    lg = []
    for _ in range(5):
        rule.update_motifs(penaltiesW=[cmf.SquarePenalty(0.5), cmf.AbsolutePenalty(0.2)], constrW=None)
        lg.append(rule.update_feature_maps(penaltiesH=[cmf.AbsolutePenalty(0.1)]))
    Wg, Hg = rule.download()
    rule.close()
    W, H = W0.copy(), H0.copy()
    orule = oracle.PGDUpdate(data, W, H)
    lo = []
    for _ in range(5):
        oracle.pgd_update_motifs(orule, data, W, H, penaltiesW_sq=(0.5,), penaltiesW_abs=(0.2,), nonneg=False)
        lo.append(oracle.pgd_update_feature_maps(orule, data, W, H, penaltiesH_abs=(0.1,), nonneg=True))
    np.testing.assert_allclose(lg, lo, rtol=1e-4)
    assert frob_rel(Wg, W) < 1e-4 and frob_rel(Hg, H) < 1e-4


@pytest.mark.parametrize("R,N,T,K,L,loss", [(2, 100, 100, 10, 5, "square"), (3, 130, 700, 32, 20, "square"), (4, 48, 300, 4, 8, "abs"), (2, 37, 150, 33, 7, "abs")])
def test_group_pgd_masked_loss(cmf, oracle, R, N, T, K, L, loss):
    """MaskedLoss on a group: the mask is cut along T like data (own columns + the right lag halo in the transposed layout).
    (2, 100, 100, 10, 5) is the reference's own test/test.jl:41-47 configuration, mask[1:20, :] = 1."""
    data, W0, H0 = _pgd_problem(oracle, N, T, K, L)
    if (N, T) == (100, 100):
        mask = np.zeros(data.shape)
        mask[:20, :] = 1
    else:
        mask = np.random.default_rng(3).uniform(0, 1, size=data.shape) * (np.random.default_rng(4).uniform(size=data.shape) > 0.3)
    base = cmf.AbsoluteLoss() if loss == "abs" else cmf.SquareLoss()
    lf = cmf.MaskedLoss(base, mask)
    iters = 6
    rule = cmf.PGDUpdate(data, W0, H0, devices=[0] * R)
    lg = []
    for _ in range(iters):
        rule.update_motifs(loss_func=lf)
        lg.append(rule.update_feature_maps(loss_func=lf))
    Wg, Hg = rule.download()
    sg = rule.steps
    # ... and without the mask again on the same handle
    rule.update_motifs(loss_func=base)
    l_plain = rule.update_feature_maps(loss_func=base)
    rule.close()
    W, H = W0.copy(), H0.copy()
    orule = oracle.PGDUpdate(data, W, H)
    lo = []
    for _ in range(iters):
        oracle.pgd_update_motifs(orule, data, W, H, mask=mask, loss=loss)
        lo.append(oracle.pgd_update_feature_maps(orule, data, W, H, mask=mask, loss=loss))
    tol_f = 3e-4 if loss == "abs" else 1e-4  # (a residual within fp32 rounding of zero may take the other sign)
    np.testing.assert_allclose(lg, lo, rtol=1e-4)
    assert frob_rel(Wg, W) < tol_f and frob_rel(Hg, H) < tol_f
    np.testing.assert_allclose(sg, (orule.stepW, orule.stepH), rtol=1e-12)
    oracle.pgd_update_motifs(orule, data, W, H, loss=loss)
    lo_plain = oracle.pgd_update_feature_maps(orule, data, W, H, loss=loss)
    assert abs(l_plain - lo_plain) <= 1e-4 * lo_plain


def test_group_pgd_refuses_mu_only_options(cmf):
    data = np.random.default_rng(0).random((8, 200))
    W, H = np.ones((2, 8, 4)), np.ones((2, 200))
    rule = cmf.PGDUpdate(data, W, H, devices=[0, 0])
    rule.set_option("gram", 1)
    with pytest.raises(cmf.CMFError):
        rule.update_motifs()
    rule.set_option("gram", 0)
    rule.update_motifs()
    assert np.isfinite(rule.update_feature_maps())
    rule.close()


@pytest.mark.parametrize("world,masked,unitnorm", [(2, False, False), (3, True, False), (2, False, True)])
def test_sharded_processes_pgd(oracle, tmp_path, world, masked, unitnorm):
    """ShardedPGDUpdate: one process per shard (gloo ranks sharing GPU 0, the library's collectives through the host
    callbacks), optionally with a MaskedLoss whose GLOBAL mask every rank cuts to its block."""
    N, T, K, L, iters = 65, 400, 5, 10, 5
    out = str(tmp_path / "res.npz")
    got = run_ranks(world, "hip_pgd_masked" if masked else ("hip_pgd_unitnorm" if unitnorm else "hip_pgd"), out, N, T, K, L, iters, 0)
    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=min(L, 20), seed=1234)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=0)
    mask = (np.random.default_rng(5).uniform(size=data.shape) > 0.25).astype(float) if masked else None
    cons = dict(constrW="unitnorm", constrH="unitnorm") if unitnorm else {}
    Wr, Hr, lr, sr = oracle.fit_pgd(data, W0, H0, max_itr=iters, mask=mask, **cons)
    np.testing.assert_allclose(got["loss_hist"], lr, rtol=1e-4)
    assert frob_rel(got["W"], Wr) < 1e-4 and frob_rel(got["H"], Hr) < 1e-4
    np.testing.assert_allclose(got["steps"], sr, rtol=1e-12)


# ---- a stream per shard: RCCL's stream semantics on one GPU ----------------------------------------------------------
@pytest.mark.parametrize("R,N,T,K,L,reg,overlap,gram", [(2, 130, 900, 32, 20, 0, False, 0), (4, 70, 517, 32, 20, 1, True, 0), (3, 40, 333, 5, 10, 1, False, 1),
                                                         (8, 96, 1100, 32, 20, 0, True, 1), (8, 96, 700, 32, 20, 0, False, 0)])
def test_group_with_a_stream_per_shard(cmf, oracle, R, N, T, K, L, reg, overlap, gram):
    """The plain loopback transport puts all shards of a one-GPU group on ONE stream, which would hide a missing
    dependency between shards.  CMF_COMM_LOOPBACK_STREAMS gives every shard its own stream and keeps RCCL's semantics
    for the collectives (start when every shard's stream has arrived, every stream continues when done) through events:
    the group iteration -- call by call and as a pipelined cmf_iterate batch, overlap and Gram forms included -- must give
    the shared-stream group's results bitwise."""
    from cmf_jl_amd import _lib

    iters = 6
    data, W0, H0, Wr, Hr, lr = oracle_fit(oracle, N, T, K, L, iters, reg)
    kw = REG if reg else dict(l1W=0.0, l2W=0.0, l1H=0.0, l2H=0.0)
    res = {}
    for tr in (_lib.CMF_COMM_LOOPBACK, _lib.CMF_COMM_LOOPBACK_STREAMS):
        for mode in ("calls", "iterate"):
            rule = cmf.MultUpdate(data, W0, H0, devices=[0] * R, transport=tr)
            assert ("loopback-streams" in rule.comm_info()) == (tr == _lib.CMF_COMM_LOOPBACK_STREAMS)
            if gram:
                rule.set_option("gram", 1)
            if overlap:
                rule.set_overlap(True)
            res[(tr, mode)] = _mu(rule, mode, iters, kw)
            rule.close()
    for mode in ("calls", "iterate"):
        a, b = res[(_lib.CMF_COMM_LOOPBACK, mode)], res[(_lib.CMF_COMM_LOOPBACK_STREAMS, mode)]
        np.testing.assert_allclose(b[0], lr, rtol=1e-4)
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
        np.testing.assert_array_equal(a[2], b[2])


def test_group_pgd_with_a_stream_per_shard(cmf, oracle):
    from cmf_jl_amd import _lib

    data, W0, H0 = _pgd_problem(oracle, 130, 900, 32, 20)
    mask = (np.random.default_rng(5).uniform(size=data.shape) > 0.25).astype(float)
    lf = cmf.MaskedLoss(cmf.SquareLoss(), mask)
    rule = cmf.PGDUpdate(data, W0, H0, devices=[0] * 4, transport=_lib.CMF_COMM_LOOPBACK_STREAMS)
    lg = []
    for _ in range(5):
        rule.update_motifs(loss_func=lf, constrW=cmf.UnitNormConstraint())
        lg.append(rule.update_feature_maps(loss_func=lf, constrH=cmf.UnitNormConstraint()))
    Wg, Hg = rule.download()
    rule.close()
    Wr, Hr, lr, _ = oracle.fit_pgd(data, W0, H0, max_itr=5, mask=mask, constrW="unitnorm", constrH="unitnorm")
    np.testing.assert_allclose(lg, lr[1:], rtol=1e-4)
    assert frob_rel(Wg, Wr) < 1e-4 and frob_rel(Hg, Hr) < 1e-4


def test_config2_eight_shards_with_a_stream_per_shard(cmf):
    """BASELINE config 2 cut into the 8 shards of `bench.py --gpus 8`, every shard on its own stream of GPU 0, 10 pipelined
    iterations: bitwise the shared-stream group (whose agreement with the oracle test_gpu_parity.py checks at this size)."""
    from cmf_jl_amd import _lib

    data = cmf.gen_synthetic(N=2000, T=50000, seed=1234)
    W0, H0 = cmf.init_rand(data, L=20, K=32, seed=0)
    out = []
    for tr in (_lib.CMF_COMM_LOOPBACK, _lib.CMF_COMM_LOOPBACK_STREAMS):
        for overlap in (False, True):
            rule = cmf.MultUpdate(data, W0, H0, devices=[0] * 8, transport=tr)
            rule.set_overlap(overlap)
            ls = list(rule.iterate(10))
            rule.synchronize()
            out.append((tr, overlap, np.asarray(ls)) + rule.download())
            rule.close()
    for r in out[1:]:
        if r[1] == out[0][1]:
            np.testing.assert_array_equal(r[2], out[0][2])
            np.testing.assert_array_equal(r[3], out[0][3])
            np.testing.assert_array_equal(r[4], out[0][4])
    np.testing.assert_array_equal(out[1][2], out[3][2])  # overlap form: streams vs shared stream
    np.testing.assert_array_equal(out[1][3], out[3][3])


# ---- seeded ragged shapes through every group rule ---------------------------------------------------------------------
def _ragged_group_shapes(n, seed):
    rng = np.random.default_rng(seed)
    shapes = []
    while len(shapes) < n:
        K = int(rng.choice([1, 2, 5, 17, 31, 32, 33, 64]))
        L = int(rng.choice([1, 2, 3, 7, 19, 20, 31, 32, 33, 40]))
        N = int(rng.choice([1, 2, 31, 63, 64, 65, 127, 128, 129, 200]))
        R = int(rng.choice([2, 3, 4, 5, 8]))
        T = int(rng.choice([64, 127, 128, 129, 300, 511, 513, 700, 1100]))
        if -(-T // R) >= max(L - 1, 1) and T - (R - 1) * -(-T // R) >= max(L - 1, 1):  # every shard holds >= L-1 columns
            shapes.append((R, N, T, K, L))
    return shapes


@pytest.mark.parametrize("R,N,T,K,L", _ragged_group_shapes(14, 2024))
def test_group_rules_on_ragged_shapes(cmf, oracle, R, N, T, K, L):
    """Every group rule -- MU in the reference formulation, its Gram form (where T >= 4 L), PGD with a mask -- on seeded
    ragged shapes (K and L around the 32-wide blocks, N and T around the tile edges, uneven last shards), stream per shard,
    against the oracle."""
    from cmf_jl_amd import _lib

    data, W0, H0 = _pgd_problem(oracle, N, T, K, L, seed=77)
    iters = 3
    Wr, Hr, lr, _ = oracle.fit_mult(data, W0, H0, max_itr=iters, check_convergence=False, **REG)
    for gram in (0, 1):
        if gram and T < 4 * L:
            continue
        rule = cmf.MultUpdate(data, W0, H0, devices=[0] * R, transport=_lib.CMF_COMM_LOOPBACK_STREAMS)
        if gram:
            rule.set_option("gram", 1)
        ls, W, H = _mu(rule, "iterate", iters, REG)
        rule.close()
        np.testing.assert_allclose(ls, lr, rtol=1e-4, err_msg=f"gram={gram}")
        assert frob_rel(W, Wr) < 1e-4 and frob_rel(H, Hr) < 1e-4, (gram, frob_rel(W, Wr), frob_rel(H, Hr))
    mask = (np.random.default_rng(5).uniform(size=data.shape) > 0.25).astype(float)
    lf = cmf.MaskedLoss(cmf.SquareLoss(), mask)
    rule = cmf.PGDUpdate(data, W0, H0, devices=[0] * R, transport=_lib.CMF_COMM_LOOPBACK_STREAMS)
    lg = []
    for _ in range(iters):
        rule.update_motifs(loss_func=lf)
        lg.append(rule.update_feature_maps(loss_func=lf))
    Wg, Hg = rule.download()
    sg = rule.steps
    rule.close()
    Wo, Ho, lo, so = oracle.fit_pgd(data, W0, H0, max_itr=iters, mask=mask)
    np.testing.assert_allclose(lg, lo[1:], rtol=1e-4)
    assert frob_rel(Wg, Wo) < 1e-4 and frob_rel(Hg, Ho) < 1e-4
    np.testing.assert_allclose(sg, so, rtol=1e-12)


def test_long_recording_needs_a_group_and_runs_on_one(cmf, oracle):
    """The reference's long-recording use (notebooks/test_mouse.ipynb cell 5: a 3 x 19 980 000 matrix, fitted with PGDUpdate): ONE
    handle refuses that many columns (a 64-row block of est', 64 * Tpad * 4 bytes, passes the 2 GiB its kernels' 32-bit offsets
    address: 8.3 M columns per handle), a T-sharded group takes them -- here as three shards on this GPU -- and its PGD and MU
    iterations match the fp64 oracle."""
    from cmf_jl_amd import _lib

    N, T, K, L = 3, 19_980_000, 3, 20
    rng = np.random.default_rng(7)
    data = np.asfortranarray(rng.random((N, T)))
    W0 = np.asfortranarray(rng.random((K, N, L)) * 0.2)
    H0 = np.asfortranarray(rng.random((K, T)) * 0.2)
    import ctypes

    lib, hs = cmf.load_library(), ctypes.c_void_p()
    dl = np.asfortranarray(data[:, :T // 2 + L - 1])  # one shard of HALF the columns is still too long for a handle
    assert lib.cmf_create_shard(ctypes.byref(hs), 0, N, T // 2, K, L, dl.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 0, T) == _lib.CMF_ERR_UNSUPPORTED
    del dl
    # the plain rule constructor (cmf_create) cuts such a recording into shards on the same device by itself
    rule = cmf.PGDUpdate(data, W0, H0)
    assert "transport=loopback" in rule.comm_info() and "nranks=3" in rule.comm_info()
    assert rule.shard_bounds(2) == (2 * (T // 3), T)
    with pytest.raises(cmf.CMFError):
        cmf.HALSUpdate(data, W0, H0)  # the H sweep of HALS is one chain along T: no sharding, and it says so
    lg = [rule.compute_loss()]
    for _ in range(2):
        rule.update_motifs()
        lg.append(rule.update_feature_maps())
    Wg, Hg = rule.download()
    sg = rule.steps
    rule.close()
    Wo, Ho, lo, so = oracle.fit_pgd(data, W0, H0, max_itr=2)
    np.testing.assert_allclose(lg, lo, rtol=1e-4)
    assert frob_rel(Wg, Wo) < 1e-4 and frob_rel(Hg, Ho) < 1e-4
    np.testing.assert_allclose(sg, so, rtol=1e-12)
    del Wo, Ho
    res = cmf.fit_cnmf(data, L=L, K=K, alg=":mult", max_itr=2, check_convergence=False, W_init=W0, H_init=H0)  # the public entry, no devices=
    lm, Wm, Hm = res.loss_hist, res.W, res.H
    Wr, Hr, lr, _ = oracle.fit_mult(data, W0, H0, max_itr=2, check_convergence=False)
    np.testing.assert_allclose(lm, lr, rtol=1e-4)
    assert frob_rel(Wm, Wr) < 1e-4 and frob_rel(Hm, Hr) < 1e-4


@pytest.mark.parametrize("N,T,K,L,cap", [(40, 1000, 5, 10, 300), (130, 900, 32, 20, 250), (17, 333, 3, 7, 64)])
def test_long_recording_paths_at_small_sizes(cmf, oracle, N, T, K, L, cap):
    """The paths a recording longer than one handle takes -- cmf_create cutting it into shards on the same device, init_rand
    going through it in column blocks with the L-1 columns in front of each block -- forced at small sizes (CMF_MAX_COLUMNS), so
    that they can be checked against the oracle quickly: init_rand's least-squares scale, then whole fits by fit_cnmf."""
    import os

    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=min(L, 20), seed=1234)
    Wo, Ho = oracle.c_init_rand(data, L=L, K=K, seed=3)
    os.environ["CMF_MAX_COLUMNS"] = str(cap)
    os.environ["CMF_TEST_HOOKS"] = "1"
    try:
        Wg, Hg = cmf.init_rand(data, L=L, K=K, seed=3)
        est_b = cmf.tensor_conv(Wo, Ho)              # the stand-alone primitives in column blocks
        tc_b = cmf.tensor_transconv(Wo, data)
        syn_b = cmf.gen_synthetic(N=N, T=T, seed=5)  # (its tensor_conv goes through the same blocks)
        res = cmf.fit_cnmf(data, L=L, K=K, alg=":mult", max_itr=5, check_convergence=False, W_init=Wo, H_init=Ho, l1_H=0.1, l2_W=0.5)
        rule = cmf.MultUpdate(data, Wo, Ho)
        info = rule.comm_info()
        rule.close()
        resp = cmf.fit_cnmf(data, L=L, K=K, alg=":pgd", max_itr=4, check_convergence=False, W_init=Wo, H_init=Ho)
    finally:
        os.environ.pop("CMF_MAX_COLUMNS", None)
        os.environ.pop("CMF_TEST_HOOKS", None)
    assert frob_rel(Wg, Wo) < 1e-6 and frob_rel(Hg, Ho) < 1e-6  # same uniforms, the scale from the blockwise sums
    assert frob_rel(est_b, oracle.tensor_conv(Wo, Ho)) < 2e-6 and frob_rel(tc_b, oracle.tensor_transconv(Wo, data)) < 2e-6
    np.testing.assert_array_equal(syn_b, cmf.gen_synthetic(N=N, T=T, seed=5))  # blocks or not: the same kernels on the same windows
    R = -(-T // (cap - L))
    assert "transport=loopback" in info and f"nranks={R}" in info
    Wr, Hr, lr, _ = oracle.fit_mult(data, Wo, Ho, max_itr=5, check_convergence=False, l1H=0.1, l2W=0.5)
    np.testing.assert_allclose(res.loss_hist, lr, rtol=1e-4)
    assert frob_rel(res.W, Wr) < 1e-4 and frob_rel(res.H, Hr) < 1e-4
    Wp, Hp, lp, _ = oracle.fit_pgd(data, Wo, Ho, max_itr=4)
    np.testing.assert_allclose(resp.loss_hist, lp, rtol=1e-4)
    assert frob_rel(resp.W, Wp) < 1e-4 and frob_rel(resp.H, Hp) < 1e-4
