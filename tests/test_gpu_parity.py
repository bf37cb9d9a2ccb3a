"""GPU parity tests: the HIP path (through the C ABI of libcmf_hip.so) against the CPU oracle.

Tolerances (north_star: W, H, loss_hist within 1e-4 relative of the fp64 CPU reference,
fp32 on the device; defined norm-wise per BASELINE.md section 4):
    REL_FACTORS = 1e-4   Frobenius-relative error of W and of H
    REL_LOSS    = 1e-4   per-entry relative error of loss_hist
    REL_PRIM    = 2e-6   Frobenius-relative error of a single conv / transconv (one fp32 contraction)
"""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL_FACTORS = 1e-4
REL_LOSS = 1e-4
REL_PRIM = 2e-6
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def cmf():
    import cmf_jl_amd as m

    lib = m.load_library()
    assert lib.cmf_device_count() >= 1, "no HIP device: the gpu tests need a real MI355X"
    return m


def frob_rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


def rand_problem(seed, N, T, K, L):
    rng = np.random.default_rng(seed)
    return rng.random((K, N, L)), rng.random((K, T)), rng.random((N, T))


# (N, T, K, L): odd sizes, K/L edge cases, multi k-block (K>32), multi lag-block (L>32), T<L
PRIM_SHAPES = [
    (48, 300, 4, 8),
    (7, 23, 3, 4),
    (5, 9, 1, 1),
    (1, 17, 2, 5),
    (6, 3, 2, 5),       # T < L
    (4, 5, 3, 5),       # T == L
    (70, 257, 5, 10),   # config-1 K, L
    (130, 700, 32, 20), # config-2 K, L; N crosses a 128 tile
    (37, 150, 33, 7),   # two k blocks
    (20, 200, 6, 40),   # two lag blocks
    (9, 1100, 2, 3),    # several 512-wide t tiles
    (50, 400, 32, 7),   # K multiple of 32 (barrier-free conv path), odd L
    (40, 300, 32, 33),  # ... two lag blocks, odd remainder
    (260, 600, 64, 20), # ... two k blocks, three n tiles
]


@pytest.mark.parametrize("N,T,K,L", PRIM_SHAPES)
def test_tensor_conv(cmf, oracle, N, T, K, L):
    W, H, _ = rand_problem(1, N, T, K, L)
    got = cmf.tensor_conv(W, H)
    ref = oracle.tensor_conv(W, H)
    assert got.shape == (N, T)
    assert frob_rel(got, ref) < REL_PRIM
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max())


@pytest.mark.parametrize("N,T,K,L", PRIM_SHAPES)
def test_tensor_transconv(cmf, oracle, N, T, K, L):
    W, _, X = rand_problem(2, N, T, K, L)
    got = cmf.tensor_transconv(W, X)
    ref = oracle.tensor_transconv(W, X)
    assert got.shape == (K, T)
    assert frob_rel(got, ref) < REL_PRIM
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max())


def test_conv_truncation_at_edges(cmf):
    """common.jl:29-31: terms with t-l < 1 are dropped; impulses in H make that visible exactly."""
    K, N, L, T = 2, 3, 4, 10
    W = np.arange(1, K * N * L + 1, dtype=np.float64).reshape(K, N, L)
    H = np.zeros((K, T))
    H[1, 0] = 1.0      # est[:, l] = W[1, :, l] for l < L
    H[0, T - 2] = 2.0  # only lags 0 and 1 fit before the right edge
    expect = np.zeros((N, T))
    expect[:, :L] = W[1]
    expect[:, T - 2] = 2 * W[0, :, 0]
    expect[:, T - 1] = 2 * W[0, :, 1]
    np.testing.assert_array_equal(cmf.tensor_conv(W, H), expect)
    # transconv drops t+l > T (common.jl:76-78): an impulse in X at the last column reaches
    # out[:, T-1-l] through lag l only
    X = np.zeros((N, T))
    X[2, T - 1] = 1.0
    out = cmf.tensor_transconv(W, X)
    expect_t = np.zeros((K, T))
    for l in range(L):
        expect_t[:, T - 1 - l] = W[:, 2, l]
    np.testing.assert_array_equal(out, expect_t)


@pytest.mark.parametrize("N,T,K,L", [(48, 300, 4, 8), (130, 700, 32, 20), (37, 150, 33, 7), (20, 200, 6, 40), (6, 3, 2, 5)])
@pytest.mark.parametrize("reg", [dict(), dict(l1W=0.1, l2W=0.5, l1H=0.1, l2H=0.2)])
def test_single_iteration(cmf, oracle, N, T, K, L, reg):
    """One update_motifs! + update_feature_maps! (mult.jl:23-58) against the oracle."""
    W0, H0, data = rand_problem(3, N, T, K, L)
    rule = cmf.MultUpdate(data, W0, H0)
    rule.update_motifs(l1W=reg.get("l1W", 0), l2W=reg.get("l2W", 0))
    Wg, _ = rule.download()
    loss = rule.update_feature_maps(l1H=reg.get("l1H", 0), l2H=reg.get("l2H", 0))
    Wg2, Hg = rule.download()
    rule.close()
    Wr, Hr = W0.copy(), H0.copy()
    orule = oracle.MultUpdate(data, Wr, Hr)
    oracle.update_motifs(orule, data, Wr, Hr, l1W=reg.get("l1W", 0), l2W=reg.get("l2W", 0))
    assert frob_rel(Wg, Wr) < 1e-5
    lr = oracle.update_feature_maps(orule, data, Wr, Hr, l1H=reg.get("l1H", 0), l2H=reg.get("l2H", 0))
    np.testing.assert_array_equal(Wg, Wg2)  # update_feature_maps! does not touch W
    assert frob_rel(Hg, Hr) < 1e-5
    assert abs(loss - lr) <= 1e-5 * lr


@pytest.mark.parametrize("name", ["mu_small", "mu_small_reg", "mu_k5"])
def test_golden_fit(cmf, name):
    """Full fits against the committed fixtures (tests/golden/make_golden.py)."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    data, W0, H0 = g["data"], g["W0"], g["H0"]
    K, N, L = W0.shape
    res = cmf.fit_cnmf(data, L=L, K=K, alg=":mult", max_itr=int(g["max_itr"]), check_convergence=False,
                       W_init=W0, H_init=H0, l1_W=float(g["l1W"]), l2_W=float(g["l2W"]),
                       l1_H=float(g["l1H"]), l2_H=float(g["l2H"]))
    assert len(res.loss_hist) == int(g["max_itr"]) + 1 and res.time_hist[0] == 0.0
    np.testing.assert_allclose(res.loss_hist, g["loss_hist"], rtol=REL_LOSS)
    assert frob_rel(res.W, g["W"]) < REL_FACTORS
    assert frob_rel(res.H, g["H"]) < REL_FACTORS
    assert frob_rel(cmf.tensor_conv(W0, H0), g["conv0"]) < REL_PRIM
    assert frob_rel(cmf.tensor_transconv(W0, data), g["transconv0"]) < REL_PRIM


@pytest.mark.parametrize("name,alg", [("hals_small", ":hals"), ("pgd_small", ":pgd")])
def test_golden_fit_hals_pgd(cmf, name, alg):
    """fit_cnmf(alg=:hals) / (alg=:pgd) against the committed fixtures of those rules (the GPU box needs no oracle for this)."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    data, W0, H0 = g["data"], g["W0"], g["H0"]
    K, N, L = W0.shape
    res = cmf.fit_cnmf(data, L=L, K=K, alg=alg, max_itr=int(g["max_itr"]), check_convergence=False, W_init=W0, H_init=H0)
    np.testing.assert_allclose(res.loss_hist, g["loss_hist"], rtol=REL_LOSS)
    assert frob_rel(res.W, g["W"]) < REL_FACTORS and frob_rel(res.H, g["H"]) < REL_FACTORS


def test_fit_k32_l20_against_oracle(cmf, oracle):
    """Config-2 K and L at a size the oracle finishes in seconds; gen_synthetic inputs, init seed 0."""
    data, _, _ = oracle.c_gen_synthetic(N=200, T=3000, K=3, L=20, seed=1234)
    W0, H0 = oracle.c_init_rand(data, L=20, K=32, seed=0)
    res = cmf.fit_cnmf(data, L=20, K=32, alg=cmf.MultUpdate, max_itr=30, check_convergence=False, W_init=W0, H_init=H0)
    Wr, Hr, lr, _ = oracle.fit_mult(data, W0, H0, max_itr=30, check_convergence=False)
    np.testing.assert_allclose(res.loss_hist, lr, rtol=REL_LOSS)
    assert frob_rel(res.W, Wr) < REL_FACTORS
    assert frob_rel(res.H, Hr) < REL_FACTORS
    assert np.all(np.diff(res.loss_hist) <= 1e-6)  # un-regularised MU is monotone


def test_fit_regularised_k32(cmf, oracle):
    """Config-4 regularisers (README.md:52) on the same problem."""
    data, _, _ = oracle.c_gen_synthetic(N=150, T=2000, K=3, L=20, seed=1234)
    W0, H0 = oracle.c_init_rand(data, L=20, K=32, seed=0)
    reg = dict(l1_H=0.1, l2_H=0.2, l1_W=0.1, l2_W=0.5)
    res = cmf.fit_cnmf(data, L=20, K=32, alg=":mult", max_itr=20, check_convergence=False, W_init=W0, H_init=H0, **reg)
    Wr, Hr, lr, _ = oracle.fit_mult(data, W0, H0, max_itr=20, check_convergence=False, l1H=0.1, l2H=0.2, l1W=0.1, l2W=0.5)
    np.testing.assert_allclose(res.loss_hist, lr, rtol=REL_LOSS)
    assert frob_rel(res.W, Wr) < REL_FACTORS
    assert frob_rel(res.H, Hr) < REL_FACTORS


def test_early_stop_eval_mode_and_native_loop(cmf, oracle):
    data, _, _ = oracle.c_gen_synthetic(N=40, T=400, K=3, L=8, seed=9)
    W0, H0 = oracle.c_init_rand(data, L=8, K=4, seed=0)
    # early stop (alternating.jl:63-66): same stopping iteration as the oracle
    res = cmf.fit_cnmf(data, L=8, K=4, max_itr=500, tol=1e-3, patience=2, W_init=W0, H_init=H0)
    _, _, lr, _ = oracle.fit_mult(data, W0, H0, max_itr=500, tol=1e-3, patience=2)
    assert len(res.loss_hist) == len(lr) < 501
    np.testing.assert_allclose(res.loss_hist, lr, rtol=REL_LOSS)
    # eval_mode (alternating.jl:51-53): W untouched
    res2 = cmf.fit_cnmf(data, L=8, K=4, max_itr=5, eval_mode=True, check_convergence=False, W_init=W0, H_init=H0)
    assert frob_rel(res2.W, W0) < 1e-7  # fp32 round trip only
    assert res2.loss_hist[-1] < res2.loss_hist[0]
    # the in-library loop (cmf_fit) reproduces the host loop bit for bit
    rule = cmf.MultUpdate(data, W0, H0)
    lh, th, early = rule.fit_native(25, np.inf, False, 3, 1e-4, False)
    rule.close()
    res3 = cmf.fit_cnmf(data, L=8, K=4, max_itr=25, check_convergence=False, W_init=W0, H_init=H0)
    np.testing.assert_array_equal(lh, res3.loss_hist)
    assert th[0] == 0.0 and not early and np.all(np.diff(th) > 0)


@pytest.mark.parametrize("devices", [None, [0, 0, 0]])
def test_iterate_is_the_call_by_call_loop(cmf, oracle, devices):
    """cmf_iterate (losses read one iteration late, loss reduction riding on the next slab sum / all-reduce) is bit for
    bit n x (update_motifs!; update_feature_maps!) -- with and without eval_mode, for n = 0, 1 and several, on one GPU
    and on a 3-shard group -- and the options that change the launch sequence keep it so."""
    data, _, _ = oracle.c_gen_synthetic(N=70, T=517, K=3, L=10, seed=21)
    W0, H0 = oracle.c_init_rand(data, L=10, K=5, seed=3)
    kw = dict(l1W=0.1, l2W=0.5, l1H=0.1, l2H=0.2)
    mk = (lambda: cmf.MultUpdate(data, W0, H0, devices=devices)) if devices else (lambda: cmf.MultUpdate(data, W0, H0))
    for eval_mode in (False, True):
        for reuse in (1, 0):
            a, b = mk(), mk()
            a.set_option("reuse_est", reuse)
            b.set_option("reuse_est", reuse)
            assert len(a.iterate(0, eval_mode=eval_mode, **kw)) == 0
            la = list(a.iterate(1, eval_mode=eval_mode, **kw)) + list(a.iterate(5, eval_mode=eval_mode, **kw))
            lb = []
            for _ in range(6):
                if not eval_mode:
                    b.update_motifs(l1W=kw["l1W"], l2W=kw["l2W"])
                lb.append(b.update_feature_maps(l1H=kw["l1H"], l2H=kw["l2H"]))
            np.testing.assert_array_equal(la, lb)
            (Wa, Ha), (Wb, Hb) = a.download(), b.download()
            np.testing.assert_array_equal(Wa, Wb)
            np.testing.assert_array_equal(Ha, Hb)
            # a call-by-call step right after a batch sees consistent state (nothing left pending)
            a.update_motifs()
            b.update_motifs()
            assert a.update_feature_maps() == b.update_feature_maps()
            a.close()
            b.close()


def test_deterministic(cmf, oracle):
    """Slab reductions instead of atomics: two runs agree bit for bit."""
    data, _, _ = oracle.c_gen_synthetic(N=130, T=1500, K=3, L=20, seed=4)
    W0, H0 = oracle.c_init_rand(data, L=20, K=32, seed=0)
    a = cmf.fit_cnmf(data, L=20, K=32, max_itr=5, check_convergence=False, W_init=W0, H_init=H0)
    b = cmf.fit_cnmf(data, L=20, K=32, max_itr=5, check_convergence=False, W_init=W0, H_init=H0)
    np.testing.assert_array_equal(a.loss_hist, b.loss_hist)
    np.testing.assert_array_equal(a.W, b.W)
    np.testing.assert_array_equal(a.H, b.H)


def test_est_reuse_is_bitwise_neutral(cmf, oracle):
    """Option "reuse_est": skipping the redundant conv of update_motifs! must not change a single bit."""
    data, _, _ = oracle.c_gen_synthetic(N=140, T=1300, K=3, L=20, seed=6)
    W0, H0 = oracle.c_init_rand(data, L=20, K=32, seed=0)
    outs = []
    for reuse in (1, 0):
        rule = cmf.MultUpdate(data, W0, H0)
        rule.set_option("reuse_est", reuse)
        lh = [rule.compute_loss()]
        for _ in range(4):
            rule.update_motifs(l1W=0.1, l2W=0.5)
            lh.append(rule.update_feature_maps(l1H=0.1, l2H=0.2))
        W, H = rule.download()
        rule.close()
        outs.append((np.asarray(lh), W, H))
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    np.testing.assert_array_equal(outs[0][2], outs[1][2])
    # and a caller that re-uploads factors in between invalidates the kept est
    rule = cmf.MultUpdate(data, W0, H0)
    rule.compute_loss()
    rule.upload(outs[0][1], outs[0][2])
    rule.update_motifs()
    Wa, _ = rule.download()
    rule.close()
    rule = cmf.MultUpdate(data, outs[0][1], outs[0][2])
    rule.update_motifs()
    Wb, _ = rule.download()
    rule.close()
    np.testing.assert_array_equal(Wa, Wb)


def test_init_rand_and_gen_synthetic_match_oracle(cmf, oracle):
    """Same RNG spec on both sides; the product's conv runs in fp32 on the device."""
    data, W, H = cmf.gen_synthetic(N=60, T=500, K=3, L=20, seed=1234, return_factors=True)
    d2, W2, H2 = oracle.c_gen_synthetic(N=60, T=500, K=3, L=20, seed=1234)
    np.testing.assert_allclose(W, W2, rtol=1e-13)
    np.testing.assert_allclose(H, H2, rtol=1e-13)
    assert frob_rel(data, d2) < 1e-6
    Wi, Hi = cmf.init_rand(d2, L=10, K=5, seed=0)
    Wo, Ho = oracle.c_init_rand(d2, L=10, K=5, seed=0)
    assert frob_rel(Wi, Wo) < 1e-6 and frob_rel(Hi, Ho) < 1e-6
    # seeded fit_cnmf is reproducible and equals explicit init (model.jl:64-73)
    r1 = cmf.fit_cnmf(d2, L=10, K=5, max_itr=3, seed=0, check_convergence=False)
    r2 = cmf.fit_cnmf(d2, L=10, K=5, max_itr=3, check_convergence=False, W_init=Wi, H_init=Hi)
    np.testing.assert_array_equal(r1.loss_hist, r2.loss_hist)


def test_errors(cmf):
    lib = cmf.load_library()
    h = ctypes.c_void_p()
    d = np.zeros((4, 4), order="F")
    rc = lib.cmf_create(ctypes.byref(h), 0, 0, 4, 2, 2, d.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    assert rc == 1 and b">= 1" in lib.cmf_last_error()
    rc = lib.cmf_create(ctypes.byref(h), 0, 4, 4, 2, 2, None)
    assert rc == 1
    rc = lib.cmf_create(ctypes.byref(h), 9999, 4, 4, 2, 2, d.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    assert rc == 1
    with pytest.raises(ValueError):
        cmf.MultUpdate(np.zeros((4, 6)), np.zeros((2, 5, 3)), np.zeros((2, 6)))  # N mismatch
    with pytest.raises(NotImplementedError):
        cmf.fit_cnmf(np.ones((4, 6)), alg=":anls")
    rule = cmf.MultUpdate(np.ones((4, 6)), np.ones((2, 4, 3)), np.ones((2, 6)))
    with pytest.raises(cmf.CMFError):
        rule.fit_native(3, np.inf, True, 0, 1e-4, False)  # patience >= 1 (alternating.jl:30)
    rule.close()


# ---- full-size (BASELINE.json config 2) size-independent properties --------------------------
@pytest.fixture(scope="module")
def config2(cmf):
    N, T, K, L = 2000, 50000, 32, 20
    data = cmf.gen_synthetic(N=N, T=T, seed=1234)
    W0, H0 = cmf.init_rand(data, L=L, K=K, seed=0)
    return data, W0, H0


def test_config2_adjointness(cmf, config2):
    """<conv(W,H), X> == <H, transconv(W,X)> at N=2000, T=50000, K=32, L=20."""
    data, W0, H0 = config2
    est = cmf.tensor_conv(W0, H0)
    lhs = float(np.vdot(est, data))
    del est
    tc = cmf.tensor_transconv(W0, data)
    rhs = float(np.vdot(H0, tc))
    assert abs(lhs - rhs) <= 2e-6 * abs(lhs)


def test_config2_iterations(cmf, config2):
    """Monotone loss, bitwise repeatability and linearity of conv in H at full size."""
    data, W0, H0 = config2
    a = cmf.fit_cnmf(data, L=20, K=32, max_itr=4, check_convergence=False, W_init=W0, H_init=H0)
    assert len(a.loss_hist) == 5 and np.all(np.diff(a.loss_hist) < 0)
    assert np.all(a.W >= cmf.EPSILON) and np.all(a.H >= cmf.EPSILON)  # clamp floor (mult.jl:38,52)
    b = cmf.fit_cnmf(data, L=20, K=32, max_itr=4, check_convergence=False, W_init=W0, H_init=H0)
    np.testing.assert_array_equal(a.loss_hist, b.loss_hist)
    np.testing.assert_array_equal(a.H, b.H)
    # init_rand scaling is the least-squares one: <data - est, est> ~ 0 (model.jl:120)
    est = cmf.tensor_conv(W0, H0)
    assert abs(np.vdot(data - est, est)) <= 1e-5 * np.vdot(est, est)


# ---- HALS (BASELINE config 5; src/algs/hals.jl) -----------------------------------------------
HALS_SHAPES = [
    (12, 40, 3, 6),      # tiny, edge columns matter
    (48, 300, 4, 8),
    (30, 70, 2, 1),      # L = 1: no lag coupling, no edge columns
    (9, 5, 2, 8),        # T < L: every column is an edge column
    (130, 700, 32, 20),  # config-5 K, L
    (37, 200, 33, 7),    # two k blocks
    (200, 1500, 5, 10),
    (150, 900, 64, 20),  # L * Kpad = 1280: the W sweep's long-state form (20 register slots per lane)
    (64, 400, 40, 30),   # L * Kpad = 1920 (Kpad = 64): 32 slots
    (2000, 1200, 32, 20),  # config 5's N, K, L on as many columns as the CPU restatement sweeps in a few seconds
]


@pytest.mark.parametrize("N,T,K,L", HALS_SHAPES)
@pytest.mark.parametrize("reg", [dict(), dict(l1W=0.1, l2W=0.5, l1H=0.1, l2H=0.2)])
@pytest.mark.parametrize("form", ["gram", "gram_h", "resid"])
def test_hals_single_iteration(cmf, oracle, N, T, K, L, reg, form):
    """One HALS update_motifs! + update_feature_maps! (hals.jl:31-42) against the oracle, in the three forms of the
    sweeps' projections: P of the H phase as denomH - numH of the MU quantities and G of the W phase contracted from the
    stored residual (default, hals_gram = 2), both contracted from the residual (0; same bars), both as differences (1:
    G = denomW - numW carries ~20x the rounding error through the W sweep's 640 dependent updates: opt-in, looser bars)."""
    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=min(L, 20) if L > 1 else 2, seed=1234)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=0)
    rule = cmf.HALSUpdate(data, W0, H0)
    rule.set_option("hals_gram", {"gram": 1, "gram_h": 2, "resid": 0}[form])
    rule.update_motifs(l1W=reg.get("l1W", 0), l2W=reg.get("l2W", 0))
    Wg, _ = rule.download()
    loss = rule.update_feature_maps(l1H=reg.get("l1H", 0), l2H=reg.get("l2H", 0))
    _, Hg = rule.download()
    rule.close()
    Wr, Hr, lh, _ = oracle.c_fit_hals(data, W0, H0, max_itr=1, check_convergence=False, **reg)
    tolW, tolH, tolL = (3e-4, 3e-4, 5e-5) if form == "gram" else (2e-5, 5e-5, 2e-5)
    print(form, (N, T, K, L), frob_rel(Wg, Wr), frob_rel(Hg, Hr), abs(loss - lh[-1]) / lh[-1])
    assert frob_rel(Wg, Wr) < tolW
    assert frob_rel(Hg, Hr) < tolH
    assert abs(loss - lh[-1]) <= tolL * lh[-1]
    assert Wg.min() >= 0.0 and Hg.min() >= 0.0   # clamp at 0, not eps (hals.jl:110,153)


@pytest.mark.parametrize("N,T,K,L", [(150, 1000, 15, 100),   # the reference's own micro-benchmark shape (notebooks/benchmarks.ipynb cell 2): L > 64
                                     (96, 700, 64, 40),      # L * Kpad = 2560 > 2048: general W sweep, on-chip H sweep
                                     (40, 300, 3, 70),       # L just past the wave-wide window
                                     (30, 50, 4, 80)])       # T < L: every column of H is an edge column
@pytest.mark.parametrize("reg", [dict(), dict(l1W=0.1, l2W=0.5, l1H=0.1, l2H=0.2)])
def test_hals_shapes_beyond_the_on_chip_sweeps(cmf, oracle, N, T, K, L, reg):
    """hals.jl:90-154 has no shape limits; the fast sweeps do (L <= 64 for H, L * Kpad <= 2048 for W).  Beyond them the
    general sweeps run -- same recurrences, same Gauss-Seidel order, state in LDS: two iterations against the oracle."""
    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=20, seed=1234)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=0)
    rule = cmf.HALSUpdate(data, W0, H0)
    losses = []
    for _ in range(2):
        rule.update_motifs(l1W=reg.get("l1W", 0), l2W=reg.get("l2W", 0))
        losses.append(rule.update_feature_maps(l1H=reg.get("l1H", 0), l2H=reg.get("l2H", 0)))
    Wg, Hg = rule.download()
    rule.close()
    Wr, Hr, lh, _ = oracle.c_fit_hals(data, W0, H0, max_itr=2, check_convergence=False, **reg)
    print((N, T, K, L), frob_rel(Wg, Wr), frob_rel(Hg, Hr), np.abs(np.asarray(losses) - lh[1:]) / lh[1:])
    np.testing.assert_allclose(losses, lh[1:], rtol=REL_LOSS)
    assert frob_rel(Wg, Wr) < REL_FACTORS and frob_rel(Hg, Hr) < REL_FACTORS
    assert Wg.min() >= 0.0 and Hg.min() >= 0.0


@pytest.mark.parametrize("N,T,K,L", [(96, 700, 5, 10), (130, 900, 32, 20), (37, 150, 33, 7), (64, 100, 4, 33)])
def test_hals_general_sweeps_agree_with_the_on_chip_sweeps(cmf, oracle, N, T, K, L):
    """The general sweeps forced (option "hals_general") at shapes the on-chip sweeps cover: same order of updates, so the two
    agree to rounding; both against the oracle."""
    import os

    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=min(L, 20), seed=3)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=1)
    reg = dict(l1W=0.05, l2W=0.1, l1H=0.05, l2H=0.1)
    out = {}
    for general in ("0", "3"):
        rule = cmf.HALSUpdate(data, W0, H0)
        rule.set_option("hals_general", int(general))
        ls = []
        for _ in range(3):
            rule.update_motifs(l1W=reg["l1W"], l2W=reg["l2W"])
            ls.append(rule.update_feature_maps(l1H=reg["l1H"], l2H=reg["l2H"]))
        out[general] = (np.asarray(ls),) + rule.download()
        rule.close()
    Wr, Hr, lh, _ = oracle.c_fit_hals(data, W0, H0, max_itr=3, check_convergence=False, **reg)
    for general in ("0", "3"):
        ls, W, H = out[general]
        np.testing.assert_allclose(ls, lh[1:], rtol=REL_LOSS)
        assert frob_rel(W, Wr) < REL_FACTORS and frob_rel(H, Hr) < REL_FACTORS
    np.testing.assert_allclose(out["0"][0], out["3"][0], rtol=2e-5)
    assert frob_rel(out["3"][1], out["0"][1]) < 5e-5 and frob_rel(out["3"][2], out["0"][2]) < 5e-5


@pytest.mark.parametrize("N,T,K,L", [(1, 64, 1, 1), (2, 3, 2, 5), (31, 65, 5, 2), (65, 129, 17, 19), (129, 200, 33, 33), (64, 40, 2, 40),
                                     (200, 513, 31, 7), (63, 127, 64, 3)])
def test_hals_general_sweeps_on_ragged_shapes(cmf, oracle, N, T, K, L):
    """The general sweeps (forced) on ragged shapes -- single units and components, L = 1, T < L, K and L around the 32-wide
    blocks -- two iterations against the oracle."""
    import os

    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=min(L, 20) if L > 1 else 2, seed=9)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=4)
    rule = cmf.HALSUpdate(data, W0, H0)
    rule.set_option("hals_general", 3)
    ls = []
    for _ in range(2):
        rule.update_motifs(l1W=0.05, l2W=0.1)
        ls.append(rule.update_feature_maps(l1H=0.05, l2H=0.1))
    Wg, Hg = rule.download()
    rule.close()
    Wr, Hr, lh, _ = oracle.c_fit_hals(data, W0, H0, max_itr=2, check_convergence=False, l1W=0.05, l2W=0.1, l1H=0.05, l2H=0.1)
    np.testing.assert_allclose(ls, lh[1:], rtol=REL_LOSS)
    assert frob_rel(Wg, Wr) < REL_FACTORS and frob_rel(Hg, Hr) < REL_FACTORS


def test_hals_fit_against_oracle(cmf, oracle):
    data, _, _ = oracle.c_gen_synthetic(N=120, T=1200, K=3, L=20, seed=1234)
    W0, H0 = oracle.c_init_rand(data, L=20, K=8, seed=0)
    res = cmf.fit_cnmf(data, L=20, K=8, alg=":hals", max_itr=12, check_convergence=False, W_init=W0, H_init=H0)
    Wr, Hr, lr, _ = oracle.c_fit_hals(data, W0, H0, max_itr=12, check_convergence=False)
    np.testing.assert_allclose(res.loss_hist, lr, rtol=REL_LOSS)
    assert frob_rel(res.W, Wr) < REL_FACTORS   # the north star's bar (measured: 3e-6 / 4e-6 after these 12 iterations, and the
    assert frob_rel(res.H, Hr) < REL_FACTORS   # pattern of exact zeros -- hals.jl:110,153 clamp at 0 -- equals the oracle's)
    assert np.array_equal(res.W == 0, Wr == 0) and np.array_equal(res.H == 0, Hr == 0)
    assert np.all(np.diff(res.loss_hist) <= 1e-6)
    # HALS beats MU per iteration on this problem (README.md:16-23 uses :hals for that reason)
    mu = cmf.fit_cnmf(data, L=20, K=8, alg=":mult", max_itr=12, check_convergence=False, W_init=W0, H_init=H0)
    assert res.loss_hist[-1] < mu.loss_hist[-1]


def test_hals_config5_full_size(cmf, config2):
    """BASELINE config 5 (N=2000, T=50000, K=32, L=20, alg=:hals): monotone loss, exact zeros, bitwise
    repeatability, and agreement of the H sweep's pipelines: the persistent one (default; 4 and 2 puller workgroups per
    row) and the stage pipeline across segment sizes and schedules."""
    import os

    data, W0, H0 = config2
    runs = {}
    for name, opts in (("persist", {}), ("persist2", {"hals_persist": 2}), ("nochase", {"hals_chase": 0}),
                       ("stage256", {"hals_persist": 0, "hals_seg": 256}),
                       ("stage1024", {"hals_persist": 0, "hals_seg": 1024}),
                       ("stage384lag3", {"hals_persist": 0, "hals_seg": 384, "hals_lag": 3}),
                       ("resid", {"hals_gram": 0}),  # both projections contracted from the stored residual
                       ("gram", {"hals_gram": 1})):  # both as differences of the MU quantities (opt-in)
        r = cmf.fit_cnmf(data, L=20, K=32, alg=":hals", max_itr=3, check_convergence=False, W_init=W0, H_init=H0, options=opts)
        if name in ("persist", "stage256"):  # same configuration twice: bit for bit
            r2 = cmf.fit_cnmf(data, L=20, K=32, alg=":hals", max_itr=3, check_convergence=False, W_init=W0, H_init=H0, options=opts)
            np.testing.assert_array_equal(r.loss_hist, r2.loss_hist)
            np.testing.assert_array_equal(r.H, r2.H)
        runs[name] = r
    a = runs["persist"]
    assert np.all(np.diff(a.loss_hist) < 0) and a.loss_hist[-1] < 0.25
    assert a.W.min() == 0.0 and a.H.min() == 0.0               # clamp at 0 (hals.jl:110,153)
    # no pipeline changes the order of the updates; they differ in how the cross-row sums are associated (rounding level)
    for name, r in runs.items():
        # (the Gram form differs from the default in how G and P are formed, not only in summation order)
        np.testing.assert_allclose(a.loss_hist, r.loss_hist, rtol=1e-5 if name in ("gram", "resid") else 1e-6, err_msg=name)
        assert frob_rel(a.H, r.H) < (1e-3 if name in ("gram", "resid") else 1e-4), name
        print(name, float(np.max(np.abs(a.loss_hist / r.loss_hist - 1))), frob_rel(a.H, r.H))


def test_evaluate_test_and_sweep(cmf, oracle):
    """SURVEY.md section 8f callers: held-out evaluation (evaluate.jl:8-25) and parameter_sweep (model.jl:132-145)."""
    data, _, _ = oracle.c_gen_synthetic(N=60, T=800, K=3, L=10, seed=11)
    train, test = data[:, :500], data[:, 500:]
    sweep = cmf.parameter_sweep(train, L_vals=[10], K_vals=[2, 4], alg_vals=[":mult", ":hals"], max_itr=15,
                                seed=0, check_convergence=False)
    assert set(sweep) == {(10, 2, ":mult"), (10, 2, ":hals"), (10, 4, ":mult"), (10, 4, ":hals")}
    r = sweep[(10, 4, ":hals")]
    assert abs(cmf.evaluate_mse(r) - r.loss_hist[-1]) < 1e-5
    te = cmf.evaluate_test(r, test)
    # oracle: 30 HALS H-sweeps from H = 0 with W fixed
    Wf = np.asarray(r.W, dtype=np.float64)
    _, Ho, lo, _ = oracle.c_fit_hals(test, Wf, np.zeros((4, test.shape[1])), max_itr=30, eval_mode=True, check_convergence=False)
    assert abs(te - lo[-1]) <= 1e-4 * lo[-1]
    assert te > r.loss_hist[-1] * 0.5  # sanity: held-out loss is of the same order


# ---- PGD (src/algs/pgd.jl) ---------------------------------------
@pytest.mark.parametrize("N,T,K,L", [(48, 300, 4, 8), (130, 700, 32, 20), (37, 150, 33, 7), (9, 5, 2, 8)])
def test_pgd_iterations(cmf, oracle, N, T, K, L):
    """Default PGDUpdate (SquareLoss, NonnegConstraint, penaltiesW=[SquarePenalty(1)], penaltiesH=[])."""
    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=min(L, 20), seed=1234)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=0)
    iters = 8
    res = cmf.fit_cnmf(data, L=L, K=K, alg=cmf.PGDUpdate, max_itr=iters, check_convergence=False, W_init=W0, H_init=H0)
    Wr, Hr, lr, steps = oracle.fit_pgd(data, W0, H0, max_itr=iters)
    np.testing.assert_allclose(res.loss_hist, lr, rtol=REL_LOSS)
    assert frob_rel(res.W, Wr) < REL_FACTORS and frob_rel(res.H, Hr) < REL_FACTORS


def test_pgd_penalties_and_unconstrained(cmf, oracle):
    data, _, _ = oracle.c_gen_synthetic(N=60, T=500, K=3, L=10, seed=5)
    W0, H0 = oracle.c_init_rand(data, L=10, K=5, seed=1)
    rule = cmf.PGDUpdate(data, W0, H0)
    kwW = dict(penaltiesW=[cmf.SquarePenalty(0.5), cmf.AbsolutePenalty(0.2)], constrW=None)
    kwH = dict(penaltiesH=[cmf.AbsolutePenalty(0.1)], constrH=cmf.NonnegConstraint())
    lg = []
    for _ in range(6):
        rule.update_motifs(**kwW)
        lg.append(rule.update_feature_maps(**kwH))
    Wg, Hg = rule.download()
    sg = rule.steps
    rule.close()
    W, H = W0.copy(), H0.copy()
    orule = oracle.PGDUpdate(data, W, H)
    lo = []
    for _ in range(6):
        oracle.pgd_update_motifs(orule, data, W, H, penaltiesW_sq=(0.5,), penaltiesW_abs=(0.2,), nonneg=False)
        lo.append(oracle.pgd_update_feature_maps(orule, data, W, H, penaltiesH_abs=(0.1,), nonneg=True))
    np.testing.assert_allclose(lg, lo, rtol=REL_LOSS)
    assert frob_rel(Wg, W) < REL_FACTORS and frob_rel(Hg, H) < REL_FACTORS
    np.testing.assert_allclose(sg, (orule.stepW, orule.stepH), rtol=1e-12)  # same accept/reject decisions

@pytest.mark.parametrize("N,T,K,L,constrW,constrH", [(60, 500, 5, 10, "unitnorm", "nonneg"), (130, 700, 32, 20, "unitnorm", "nonneg"),
                                                      (37, 150, 33, 7, "nonneg", "unitnorm"), (48, 300, 4, 8, "unitnorm", "unitnorm")])
def test_pgd_unit_norm_constraint(cmf, oracle, N, T, K, L, constrW, constrH):
    """UnitNormConstraint (pgd.jl:100-110; constrW=UnitNormConstraint() in figures/thesis/exp_reconstruct_synth.jl:69)
    against both oracle restatements: the per-component norms never exceed 1 and the step decisions agree."""
    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=min(L, 20), seed=7)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=2)
    cls = {"unitnorm": cmf.UnitNormConstraint(), "nonneg": cmf.NonnegConstraint()}
    rule = cmf.PGDUpdate(data, W0, H0)
    lg = []
    for _ in range(6):
        rule.update_motifs(constrW=cls[constrW])                       # penaltiesW = [SquarePenalty(1)] (pgd.jl:162)
        lg.append(rule.update_feature_maps(constrH=cls[constrH]))
    Wg, Hg = rule.download()
    sg = rule.steps
    rule.close()
    Wr, Hr, lr, sr = oracle.fit_pgd(data, W0, H0, max_itr=6, constrW=constrW, constrH=constrH)
    Wc, Hc, lc, sc = oracle.c_fit_pgd(data, W0, H0, max_itr=6, constrW=constrW, constrH=constrH)
    np.testing.assert_allclose(lr[1:], lc[1:], rtol=1e-9)
    np.testing.assert_allclose(lg, lr[1:], rtol=REL_LOSS)
    assert frob_rel(Wg, Wr) < REL_FACTORS and frob_rel(Hg, Hr) < REL_FACTORS
    np.testing.assert_allclose(sg, sr, rtol=1e-12)
    if constrW == "unitnorm":
        assert max(np.linalg.norm(Wg[k]) for k in range(K)) <= 1 + 1e-5  # (no clamp under this constraint: negative entries are legal)
    if constrH == "unitnorm":
        assert max(np.linalg.norm(Hg[k]) for k in range(K)) <= 1 + 1e-5


@pytest.mark.parametrize("N,T,K,L,masked", [(60, 500, 5, 10, False), (130, 700, 32, 20, False), (48, 300, 4, 8, True), (37, 150, 33, 7, True)])
def test_pgd_absolute_loss(cmf, oracle, N, T, K, L, masked):
    """AbsoluteLoss (pgd.jl:41-47): gradient sign(est - data) formed in the conv epilogue, loss norm(data - est, 1);
    alone and under MaskedLoss (pgd.jl:58-70).  The sign of a residual within fp32 rounding of zero can differ from
    the fp64 oracle's, so the factors are compared a little looser than the smooth losses."""
    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=min(L, 20), seed=9)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=4)
    rng = np.random.default_rng(1)
    mask = (rng.random((N, T)) < 0.7).astype(float) if masked else None
    lf = cmf.MaskedLoss(cmf.AbsoluteLoss(), mask) if masked else cmf.AbsoluteLoss()
    rule = cmf.PGDUpdate(data, W0, H0)
    lg = []
    for _ in range(5):
        rule.update_motifs(loss_func=lf)
        lg.append(rule.update_feature_maps(loss_func=lf))
    Wg, Hg = rule.download()
    sg = rule.steps
    rule.close()
    Wr, Hr, lr, sr = oracle.fit_pgd(data, W0, H0, max_itr=5, loss="abs", mask=mask)
    Wc, Hc, lc, sc = oracle.c_fit_pgd(data, W0, H0, max_itr=5, loss="abs", mask=mask)
    np.testing.assert_allclose(lr[1:], lc[1:], rtol=1e-9)
    np.testing.assert_allclose(lg, lr[1:], rtol=REL_LOSS)
    assert frob_rel(Wg, Wr) < 3e-4 and frob_rel(Hg, Hr) < 3e-4
    np.testing.assert_allclose(sg, sr, rtol=1e-12)
    # switching back to SquareLoss on the same handle recomputes the stored residual
    data2 = data
    rule = cmf.PGDUpdate(data2, W0, H0)
    rule.update_motifs(loss_func=cmf.AbsoluteLoss())
    rule.update_feature_maps(loss_func=cmf.AbsoluteLoss())
    rule.update_motifs()
    l_sq = rule.update_feature_maps()
    Wm, Hm = rule.download()
    rule.close()
    W, H = W0.copy(), H0.copy()
    orule = oracle.PGDUpdate(data, W, H)
    oracle.pgd_update_motifs(orule, data, W, H, loss="abs")
    oracle.pgd_update_feature_maps(orule, data, W, H, loss="abs")
    oracle.pgd_update_motifs(orule, data, W, H)
    l_sq_o = oracle.pgd_update_feature_maps(orule, data, W, H)
    assert abs(l_sq - l_sq_o) <= REL_LOSS * l_sq_o and frob_rel(Wm, W) < 3e-4


def test_pgd_masked_loss_reference_test_case(cmf, oracle):
    """The one runnable entry of the reference's own test/test.jl (:15-21, :41-47): N, T, K, L = 100, 100, 10, 5,
    data from synthetic_sequences(N, T, K, L) with seed 1234, init_rand, PGDUpdate with
    loss_func=MaskedLoss(SquareLoss(), mask), mask[1:20, :] = 1 -- run for a fixed iteration count
    instead of max_time=5 so that both sides do the same work."""
    N, T, K, L = 100, 100, 10, 5
    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=K, L=L, seed=1234)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=0)
    mask = np.zeros(data.shape)
    mask[:20, :] = 1
    iters = 25
    res = cmf.fit_cnmf(data, L=L, K=K, alg=cmf.PGDUpdate, max_itr=iters, check_convergence=False, W_init=W0, H_init=H0,
                       loss_func=cmf.MaskedLoss(cmf.SquareLoss(), mask))
    Wr, Hr, lr, _ = oracle.fit_pgd(data, W0, H0, max_itr=iters, mask=mask)
    assert len(res.loss_hist) == iters + 1
    np.testing.assert_allclose(res.loss_hist, lr, rtol=REL_LOSS)
    assert frob_rel(res.W, Wr) < REL_FACTORS and frob_rel(res.H, Hr) < REL_FACTORS
    # the masked-out rows never enter a gradient: their motifs only shrink under SquarePenalty(1) (pgd.jl:162)
    assert np.all(res.W[:, 20:, :] <= W0[:, 20:, :] + 1e-12)


@pytest.mark.parametrize("N,T,K,L", [(130, 700, 32, 20), (37, 150, 33, 7)])
def test_pgd_masked_loss_general_mask(cmf, oracle, N, T, K, L):
    """A real-valued random mask (K % 32 == 0 and the general-K kernel), switching the mask off again in the same rule."""
    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=min(L, 20), seed=7)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=2)
    mask = np.random.default_rng(3).uniform(0, 1, size=data.shape) * (np.random.default_rng(4).uniform(size=data.shape) > 0.3)
    lf = cmf.MaskedLoss(cmf.SquareLoss(), mask)
    rule = cmf.PGDUpdate(data, W0, H0)
    W, H = W0.copy(), H0.copy()
    orule = oracle.PGDUpdate(data, W, H)
    lg, lo = [], []
    for it in range(6):
        m = mask if it < 4 else None  # iterations 5 and 6 run with the plain SquareLoss again
        rule.update_motifs(loss_func=lf if it < 4 else cmf.SquareLoss())
        lg.append(rule.update_feature_maps(loss_func=lf if it < 4 else cmf.SquareLoss()))
        oracle.pgd_update_motifs(orule, data, W, H, mask=m)
        lo.append(oracle.pgd_update_feature_maps(orule, data, W, H, mask=m))
    Wg, Hg = rule.download()
    rule.close()
    np.testing.assert_allclose(lg, lo, rtol=REL_LOSS)
    assert frob_rel(Wg, W) < REL_FACTORS and frob_rel(Hg, H) < REL_FACTORS


def test_config1_full_fit(cmf, oracle):
    """BASELINE.json configs[0]: gen_synthetic N=500 T=2000, fit_cnmf alg=:mult K=5 L=10, the reference's own
    CPU-runnable case, 100 iterations (README.md:34-37 defaults) against the fp64 oracle."""
    data, _, _ = oracle.c_gen_synthetic(N=500, T=2000, K=3, L=20, seed=1234)
    W0, H0 = oracle.c_init_rand(data, L=10, K=5, seed=0)
    res = cmf.fit_cnmf(data, L=10, K=5, alg=":mult", max_itr=100, check_convergence=False, W_init=W0, H_init=H0)
    Wr, Hr, lr, _ = oracle.fit_mult(data, W0, H0, max_itr=100, check_convergence=False)
    assert len(res.loss_hist) == 101
    np.testing.assert_allclose(res.loss_hist, lr, rtol=REL_LOSS)
    assert frob_rel(res.W, Wr) < REL_FACTORS and frob_rel(res.H, Hr) < REL_FACTORS
    # and with the default early stop the iteration count agrees too (alternating.jl:63-66)
    res2 = cmf.fit_cnmf(data, L=10, K=5, alg=":mult", max_itr=100, W_init=W0, H_init=H0)
    _, _, lr2, _ = oracle.fit_mult(data, W0, H0, max_itr=100)
    assert len(res2.loss_hist) == len(lr2)


@pytest.mark.parametrize("reg,iters", [(dict(), 2), (dict(l1W=0.1, l2W=0.5, l1H=0.1, l2H=0.2), 1)])
def test_config2_full_size_against_oracle(cmf, oracle, config2, reg, iters):
    """BASELINE.json configs[1] itself (N=2000, T=50000, K=32, L=20, gen_synthetic seed 1234, init_rand seed 0) and
    configs[3] (the same with README.md:52's regularisers) against the fp64 oracle COMPUTED HERE, on the inputs the product's
    own generator made: the north star's 1e-4 bar on W, H and loss_hist at the size the metric is quoted on, for the few
    iterations the CPU restatement can afford inside a test (about 10 s each).  The whole default fit (100 iterations) at this
    size is compared with committed oracle fixtures in tests/test_gpu_full_fits.py."""
    data, W0, H0 = config2
    try:
        from threadpoolctl import threadpool_limits

        ctx = threadpool_limits(limits=16, user_api="blas")
    except Exception:  # pragma: no cover - threadpoolctl is in the image
        import contextlib

        ctx = contextlib.nullcontext()
    with ctx:
        Wr, Hr, lr, _ = oracle.fit_mult(data, W0, H0, max_itr=iters, check_convergence=False, **reg)
    rule = cmf.MultUpdate(data, W0, H0)
    lg = [rule.compute_loss()] + list(rule.iterate(iters, **reg))
    Wg, Hg = rule.download()
    rule.close()
    np.testing.assert_allclose(lg, lr, rtol=REL_LOSS)
    assert frob_rel(Wg, Wr) < REL_FACTORS and frob_rel(Hg, Hr) < REL_FACTORS
    print("config 2 vs oracle after", iters, "iterations: relW", frob_rel(Wg, Wr), "relH", frob_rel(Hg, Hr),
          "max rel loss", float(np.max(np.abs(np.asarray(lg) - lr) / lr)))
    # the same problem as the 8 shards of 6250 columns bench.py --gpus 8 runs (all on this one GPU, loopback transport):
    # the T-sharded iteration at the metric's size against the same oracle fit
    rule = cmf.MultUpdate(data, W0, H0, devices=[0] * 8)
    l8 = [rule.compute_loss()] + list(rule.iterate(iters, **reg))
    W8, H8 = rule.download()
    rule.close()
    np.testing.assert_allclose(l8, lr, rtol=REL_LOSS)
    assert frob_rel(W8, Wr) < REL_FACTORS and frob_rel(H8, Hr) < REL_FACTORS
    # the optional Gram form (option gram = 1: denomW = (H_unfold H_unfold') W, denomH from the lag-Gram taps of W) at the
    # same size against the same oracle fit -- unsharded and as the 8-shard group whose all-reduce carries [numW | HH]
    for devices, gram in ((None, 1), ([0] * 8, 1), (None, 2)):
        rule = cmf.MultUpdate(data, W0, H0, devices=devices)
        rule.set_option("gram", gram)
        lgm = [rule.compute_loss()] + list(rule.iterate(iters, **reg))
        Wm, Hm = rule.download()
        rule.close()
        np.testing.assert_allclose(lgm, lr, rtol=REL_LOSS)
        assert frob_rel(Wm, Wr) < REL_FACTORS and frob_rel(Hm, Hr) < REL_FACTORS
        print("  gram =", gram, "8 shards" if devices else "unsharded", "relW", frob_rel(Wm, Wr), "relH", frob_rel(Hm, Hr),
              "max rel loss", float(np.max(np.abs(np.asarray(lgm) - lr) / lr)))


def test_config4_regularised_full_size(cmf, config2):
    """BASELINE.json configs[3] (README.md:52 regularisers) at full size: runs, stays positive, and differs from
    the unregularised path in the expected direction (smaller W)."""
    data, W0, H0 = config2
    reg = cmf.fit_cnmf(data, L=20, K=32, max_itr=3, check_convergence=False, W_init=W0, H_init=H0,
                       l1_H=0.1, l2_H=0.2, l1_W=0.1, l2_W=0.5)
    plain = cmf.fit_cnmf(data, L=20, K=32, max_itr=3, check_convergence=False, W_init=W0, H_init=H0)
    assert np.all(np.isfinite(reg.loss_hist)) and reg.W.min() >= cmf.EPSILON and reg.H.min() >= cmf.EPSILON
    # the W penalties shrink W (H may compensate); the two paths must actually differ
    assert np.linalg.norm(reg.W) < np.linalg.norm(plain.W)
    assert reg.loss_hist[-1] != plain.loss_hist[-1] and frob_rel(reg.H, plain.H) > 1e-7


# ---- optional Gram form of the MU denominators ---------------------------
@pytest.mark.parametrize("N,T,K,L", [(48, 300, 4, 8), (130, 700, 32, 20), (37, 150, 33, 7), (6, 3, 2, 5), (20, 200, 6, 40)])
@pytest.mark.parametrize("gram", [1, 2])
def test_gram_form_matches_oracle(cmf, oracle, N, T, K, L, gram):
    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=min(L, 20), seed=1234)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=0)
    reg = dict(l1W=0.1, l2W=0.5, l1H=0.1, l2H=0.2)
    rule = cmf.MultUpdate(data, W0, H0)
    rule.set_option("gram", gram)
    lg = []
    for _ in range(10):
        rule.update_motifs(l1W=reg["l1W"], l2W=reg["l2W"])
        lg.append(rule.update_feature_maps(l1H=reg["l1H"], l2H=reg["l2H"]))
    Wg, Hg = rule.download()
    rule.close()
    Wr, Hr, lr, _ = oracle.fit_mult(data, W0, H0, max_itr=10, check_convergence=False, **reg)
    np.testing.assert_allclose(lg, lr[1:], rtol=REL_LOSS)  # gram = 2 too: its Gram-sum loss is measured at <= 5e-7 on these shapes
    assert frob_rel(Wg, Wr) < REL_FACTORS and frob_rel(Hg, Hr) < REL_FACTORS


def test_gram_form_100_iterations_config1(cmf, oracle):
    """BASELINE configs[0] (N=500, T=2000, K=5, L=10) for 100 iterations: the Gram form's rounding-level differences do
    not drift away from the reference formulation -- loss_hist, W and H stay inside the 1e-4 bar to the end."""
    data, _, _ = oracle.c_gen_synthetic(N=500, T=2000, K=3, L=20, seed=1234)
    W0, H0 = oracle.c_init_rand(data, L=10, K=5, seed=0)
    Wr, Hr, lr, _ = oracle.fit_mult(data, W0, H0, max_itr=100, check_convergence=False)
    for gram in (0, 1):
        rule = cmf.MultUpdate(data, W0, H0)
        rule.set_option("gram", gram)
        lg = [rule.compute_loss()] + list(rule.iterate(100))
        Wg, Hg = rule.download()
        rule.close()
        np.testing.assert_allclose(lg, lr, rtol=REL_LOSS)
        assert frob_rel(Wg, Wr) < REL_FACTORS and frob_rel(Hg, Hr) < REL_FACTORS
        print("gram", gram, "after 100 iterations: relW", frob_rel(Wg, Wr), "relH", frob_rel(Hg, Hr),
              "max rel loss", float(np.max(np.abs(np.asarray(lg) - lr) / lr)))


def test_in_loop_kernel_timing(cmf, oracle):
    """Option "profile": HIP event pairs around the contraction launches of the rule entries (bench.py's roofline
    source).  Counts follow the iteration structure (est reuse: one conv_t, one conv_loss_store, one hxt and one
    transconv per iteration, plus the first iteration's plain conv); results are unchanged.  (Option "speculate" off on the
    counted handle: it moves every hxt launch behind the loss conv of the call before and adds one nobody reads at the end --
    the unprofiled reference handle runs with it.)"""
    W0, H0, data = rand_problem(11, 130, 900, 32, 20)
    ref = cmf.MultUpdate(data, W0, H0)
    rule = cmf.MultUpdate(data, W0, H0)
    rule.set_option("speculate", 0)
    rule.set_option("profile", 1)
    for _ in range(3):
        ref.update_motifs(); ref.update_feature_maps()
        rule.update_motifs(); rule.update_feature_maps()
    counts = {name: rule.kernel_times(name) for name in ("conv", "conv_t", "conv_loss", "conv_loss_store", "hxt", "transconv")}
    assert [counts[k][1] for k in ("conv", "conv_t", "conv_loss", "conv_loss_store", "hxt", "transconv")] == [1, 3, 0, 3, 3, 3]
    assert all(ms > 0 for ms, n in counts.values() if n)
    rule.set_option("profile", 1)  # restart drops the records
    assert rule.kernel_times("hxt") == (0.0, 0)
    rule.set_option("profile", 2)  # every second launch of each class
    for _ in range(4):
        ref.update_motifs(); ref.update_feature_maps()
        rule.update_motifs(); rule.update_feature_maps()
    assert rule.kernel_times("hxt")[1] == 2 and rule.kernel_times("transconv")[1] == 2
    rule.set_option("profile", 0)
    Wa, Ha = ref.download()
    Wb, Hb = rule.download()
    assert np.array_equal(Wa, Wb) and np.array_equal(Ha, Hb)
    with pytest.raises(cmf.CMFError):
        rule.kernel_times("nope")
    ref.close(); rule.close()


@pytest.mark.parametrize("variant", [2, 3])
@pytest.mark.parametrize("N,T,K,L", [(130, 700, 32, 20), (260, 600, 64, 20), (40, 300, 32, 33), (70, 130, 32, 5)])
def test_conv_kernel_variants(cmf, oracle, N, T, K, L, variant):
    """Both K % 32 == 0 conv kernels (128 x 128 workgroup tiles / one-wave 64 x 64 workgroups) in every role of an MU
    iteration and of the residual-based rules, forced through the "conv_kernel" option."""
    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=min(L, 20), seed=21)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=3)
    rule = cmf.MultUpdate(data, W0, H0)
    rule.set_option("conv_kernel", variant)
    kw = dict(l1W=0.1, l2W=0.5, l1H=0.1, l2H=0.2)
    lg = []
    for _ in range(4):
        rule.update_motifs(**kw)
        lg.append(rule.update_feature_maps(**kw))
    Wg, Hg = rule.download()
    rule.close()
    Wr, Hr, lr, _ = oracle.fit_mult(data, W0, H0, max_itr=4, check_convergence=False, **kw)
    np.testing.assert_allclose(lg, lr[1:], rtol=REL_LOSS)
    assert frob_rel(Wg, Wr) < REL_FACTORS and frob_rel(Hg, Hr) < REL_FACTORS
    pg = cmf.PGDUpdate(data, W0, H0)
    pg.set_option("conv_kernel", variant)
    lp = []
    for _ in range(3):
        pg.update_motifs()
        lp.append(pg.update_feature_maps())
    Wp, Hp = pg.download()
    pg.close()
    Wo, Ho, lo, _ = oracle.fit_pgd(data, W0, H0, max_itr=3)
    np.testing.assert_allclose(lp, lo[1:], rtol=REL_LOSS)
    assert frob_rel(Wp, Wo) < REL_FACTORS and frob_rel(Hp, Ho) < REL_FACTORS


def _random_shapes(n, seed):
    """Seeded ragged shapes: every kernel-selection boundary gets hit (K % 32 == 0 or not, one or several k blocks and
    lag blocks, N and T around the 64 / 128 / 512 tile edges, T shorter than L)."""
    rng = np.random.default_rng(seed)
    shapes = []
    for _ in range(n):
        K = int(rng.choice([1, 2, 3, 5, 17, 31, 32, 33, 64]))
        L = int(rng.choice([1, 2, 3, 7, 19, 20, 31, 32, 33, 40]))
        N = int(rng.choice([1, 2, 31, 63, 64, 65, 127, 128, 129, 200, 257]))
        T = int(rng.choice([1, 2, L, L + 1, 63, 64, 65, 127, 128, 129, 511, 512, 513, 700, 1100]))
        shapes.append((N, max(T, 1), K, L))
    return shapes


@pytest.mark.parametrize("N,T,K,L", _random_shapes(28, 2024))
def test_random_shapes_primitives_and_iteration(cmf, oracle, N, T, K, L):
    """tensor_conv, tensor_transconv and one regularised MU iteration on seeded ragged shapes (both conv kernels where
    K % 32 == 0), against the fp64 restatement."""
    W, H, X = rand_problem(N * 7919 + T * 31 + K * 7 + L, N, T, K, L)
    ref_c, ref_t = oracle.tensor_conv(W, H), oracle.tensor_transconv(W, X)
    got_c, got_t = cmf.tensor_conv(W, H), cmf.tensor_transconv(W, X)
    np.testing.assert_allclose(got_c, ref_c, rtol=2e-5, atol=2e-5 * max(np.abs(ref_c).max(), 1e-30))
    np.testing.assert_allclose(got_t, ref_t, rtol=2e-5, atol=2e-5 * max(np.abs(ref_t).max(), 1e-30))
    kw = dict(l1W=0.05, l2W=0.3, l1H=0.1, l2H=0.2)
    Wr, Hr = W.copy(), H.copy()
    orule = oracle.MultUpdate(X, Wr, Hr)
    oracle.update_motifs(orule, X, Wr, Hr, l1W=kw["l1W"], l2W=kw["l2W"])
    lr = oracle.update_feature_maps(orule, X, Wr, Hr, l1H=kw["l1H"], l2H=kw["l2H"])
    for variant in ((2, 3) if K % 32 == 0 else (0,)):
        rule = cmf.MultUpdate(X, W, H)
        rule.set_option("conv_kernel", variant)
        rule.update_motifs(l1W=kw["l1W"], l2W=kw["l2W"])
        loss = rule.update_feature_maps(l1H=kw["l1H"], l2H=kw["l2H"])
        Wg, Hg = rule.download()
        rule.close()
        assert frob_rel(Wg, Wr) < 2e-5 and frob_rel(Hg, Hr) < 2e-5
        assert abs(loss - lr) <= 2e-5 * max(lr, 1e-30)


def _random_rule_shapes(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        K = int(rng.choice([1, 3, 5, 16, 32]))
        L = int(rng.choice([1, 2, 5, 12, 20, 31]))
        N = int(rng.choice([2, 31, 64, 65, 130]))
        T = int(rng.choice([L + 2, 63, 64, 129, 300, 513, 777]))
        out.append((N, T, K, L))
    return out


@pytest.mark.parametrize("N,T,K,L", _random_rule_shapes(12, 77))
def test_random_shapes_hals_and_pgd(cmf, oracle, N, T, K, L):
    """One HALS iteration and two PGD iterations (plain and masked loss) on seeded ragged shapes against the oracle."""
    rng = np.random.default_rng(N + 13 * T + 101 * K + 1009 * L)
    data = np.asfortranarray(rng.random((N, T)))
    W0 = np.asfortranarray(rng.random((K, N, L)))
    H0 = np.asfortranarray(rng.random((K, T)))
    est = oracle.tensor_conv(W0, H0)
    s = np.sqrt(abs(np.vdot(data, est) / np.vdot(est, est)))  # the scaling of init_rand (model.jl:118-122): est ~ data,
    W0 *= s                                                     # otherwise the first residual is two orders above the
    H0 *= s                                                     # data and fp32 keeps 3 digits of what survives the clamp
    reg = dict(l1W=0.05, l2W=0.2, l1H=0.1, l2H=0.3)
    rule = cmf.HALSUpdate(data, W0, H0)
    rule.update_motifs(l1W=reg["l1W"], l2W=reg["l2W"])
    loss = rule.update_feature_maps(l1H=reg["l1H"], l2H=reg["l2H"])
    Wg, Hg = rule.download()
    rule.close()
    Wr, Hr, lh, _ = oracle.c_fit_hals(data, W0, H0, max_itr=1, check_convergence=False, **reg)
    assert frob_rel(Wg, Wr) < 5e-5 and frob_rel(Hg, Hr) < 1e-4
    assert abs(loss - lh[-1]) <= 5e-5 * lh[-1]
    mask = (rng.random((N, T)) > 0.3).astype(np.float64)
    for m in (None, mask):
        pg = cmf.PGDUpdate(data, W0, H0)
        lf = cmf.SquareLoss() if m is None else cmf.MaskedLoss(cmf.SquareLoss(), m)
        lg = []
        for _ in range(2):
            pg.update_motifs(loss_func=lf)
            lg.append(pg.update_feature_maps(loss_func=lf))
        Wp, Hp = pg.download()
        pg.close()
        Wo, Ho, lo, _ = oracle.fit_pgd(data, W0, H0, max_itr=2, mask=m)
        np.testing.assert_allclose(lg, lo[1:], rtol=REL_LOSS)
        assert frob_rel(Wp, Wo) < REL_FACTORS and frob_rel(Hp, Ho) < REL_FACTORS


@pytest.mark.parametrize("N,T,K,L", [(777, 5000, 48, 25), (300, 3000, 64, 40), (1001, 2500, 32, 20)])
def test_medium_sizes_two_iterations(cmf, oracle, N, T, K, L):
    """Mid-size shapes that exercise several k blocks / lag blocks / many tiles at once (general-K conv kernel, two k
    blocks with two lag blocks, N just past a tile edge): two regularised MU iterations against the oracle."""
    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=20, seed=99)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=5)
    kw = dict(l1W=0.1, l2W=0.5, l1H=0.1, l2H=0.2)
    res = cmf.fit_cnmf(data, L=L, K=K, alg=":mult", max_itr=2, check_convergence=False, W_init=W0, H_init=H0,
                       l1_W=kw["l1W"], l2_W=kw["l2W"], l1_H=kw["l1H"], l2_H=kw["l2H"])
    Wr, Hr, lr, _ = oracle.fit_mult(data, W0, H0, max_itr=2, check_convergence=False, **kw)
    np.testing.assert_allclose(res.loss_hist, lr, rtol=REL_LOSS)
    assert frob_rel(res.W, Wr) < REL_FACTORS and frob_rel(res.H, Hr) < REL_FACTORS


def test_hals_persistent_pipeline_waits_are_bounded(cmf, oracle):
    """The persistent H pipeline's workgroups wait for each other through flags in memory; a wait that is never satisfied
    (here: the puller workgroups leave without doing their work, option "hals_debug" = 3 under CMF_TEST_HOOKS=1) must run out and drain the grid --
    not hang the device -- and the call must still deliver the sweep: H is restored from the snapshot taken at its start,
    the sweep is redone on the stage pipeline, the event is counted, the result is the oracle's."""
    import os

    data, _, _ = oracle.c_gen_synthetic(N=40, T=600, K=3, L=8, seed=5)
    W0, H0 = oracle.c_init_rand(data, L=8, K=4, seed=2)
    ref = cmf.HALSUpdate(data, W0, H0)
    ref.update_motifs()
    loss_ref = ref.update_feature_maps()
    Wref, Href = ref.download()
    ref.close()
    rule = cmf.HALSUpdate(data, W0, H0)
    assert rule.counter("hals_pipeline_reruns") == 0
    rule.update_motifs()
    with pytest.raises(Exception, match="CMF_TEST_HOOKS"):
        rule.set_option("hals_debug", 3)  # (a wrong-results switch does not exist without the hooks)
    os.environ["CMF_TEST_HOOKS"] = "1"
    try:
        rule.set_option("hals_debug", 3)
        loss = rule.update_feature_maps()
        rule.set_option("hals_debug", 0)
    finally:
        os.environ.pop("CMF_TEST_HOOKS", None)
    assert rule.counter("hals_pipeline_reruns") == 1
    Wg, Hg = rule.download()
    _, _, lh, _ = oracle.c_fit_hals(data, W0, H0, max_itr=2, check_convergence=False)
    assert abs(loss - lh[1]) <= 1e-4 * lh[1]
    # the stage pipeline performs the same updates in the same order as the persistent one
    assert abs(loss - loss_ref) <= 1e-6 * loss_ref and frob_rel(Hg, Href) < 1e-6 and frob_rel(Wg, Wref) == 0.0
    # the handle stays usable (and keeps to the stage pipeline): a second iteration matches the oracle too
    rule.update_motifs()
    loss2 = rule.update_feature_maps()
    assert rule.counter("hals_pipeline_reruns") == 1
    assert abs(loss2 - lh[2]) <= 1e-4 * lh[2]
    rule.close()


def test_hals_residual_conv_chasing_the_row_pipeline(cmf, oracle):
    """Option "hals_chase": the first tile rows of the residual conv run on the CUs the persistent H pipeline leaves free, each tile
    waiting for the LAST row's progress flag and reading H with agent-scope loads while the pipeline is still sweeping.  Same
    arithmetic per tile as the one-launch conv: W, H and the losses are those of chase = 0 to rounding, run to run bit for bit, and
    the oracle's."""
    N, T, K, L = 130, 9000, 32, 20
    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=20, seed=3)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=1)
    out = {}
    for name, chase in (("off", 0), ("on", 65), ("on2", 65), ("most", 95)):
        rule = cmf.HALSUpdate(data, W0, H0)
        rule.set_option("hals_chase", chase)
        ls = []
        for _ in range(3):
            rule.update_motifs(l1W=0.05, l2W=0.1)
            ls.append(rule.update_feature_maps(l1H=0.05, l2H=0.1))
        assert rule.counter("hals_pipeline_reruns") == 0
        out[name] = (np.array(ls),) + rule.download()
        rule.close()
    for a, b in zip(out["on"], out["on2"]):
        assert np.array_equal(a, b)  # reproducible
    for name in ("on", "most"):
        # (the chased form runs three pullers per row instead of four: the cross-row sums are associated differently -- rounding level)
        np.testing.assert_allclose(out[name][0], out["off"][0], rtol=1e-6)
        assert frob_rel(out[name][1], out["off"][1]) < 1e-6 and frob_rel(out[name][2], out["off"][2]) < 1e-6
    Wr, Hr, lr, _ = oracle.c_fit_hals(data, W0, H0, max_itr=3, check_convergence=False, l1W=0.05, l2W=0.1, l1H=0.05, l2H=0.1)
    np.testing.assert_allclose(out["on"][0], lr[1:], rtol=1e-4)
    assert frob_rel(out["on"][1], Wr) < 1e-4 and frob_rel(out["on"][2], Hr) < 1e-4


def test_hals_chasing_conv_leaves_when_the_pipeline_aborts(cmf, oracle):
    """The chasing tiles' waits are bounded like the pipeline's own: when the pipeline aborts (its pullers leave at once: every
    sweeper's wait runs out) the chasing launch sees the abort word and drains, the sweep and the conv are redone, the result is the
    unchased one's."""
    import os

    N, T, K, L = 40, 5000, 32, 8
    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=8, seed=5)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=2)
    ref = cmf.HALSUpdate(data, W0, H0)
    ref.set_option("hals_chase", 0)
    ref.update_motifs()
    want = ref.update_feature_maps()
    Wref, Href = ref.download()
    ref.close()
    rule = cmf.HALSUpdate(data, W0, H0)
    rule.update_motifs()
    os.environ["CMF_TEST_HOOKS"] = "1"
    try:
        rule.set_option("hals_debug", 3)
        got = rule.update_feature_maps()
        rule.set_option("hals_debug", 0)
    finally:
        os.environ.pop("CMF_TEST_HOOKS", None)
    assert rule.counter("hals_pipeline_reruns") == 1
    Wg, Hg = rule.download()
    assert abs(got - want) <= 1e-6 * want and frob_rel(Hg, Href) < 1e-6 and np.array_equal(Wg, Wref)  # (W: the same first W sweep)
    rule.update_motifs()
    rule.update_feature_maps()  # the handle stays usable
    rule.close()


def test_waiting_for_a_loss_puts_nothing_into_the_stream(cmf):
    """Until round 6 the host's wait for a loss called hipStreamQuery every 4096 polls as its liveness check, and the runtime answered
    each query by putting a marker packet (a barrier with a system-scope release) into the stream: 5.9 us of idle device behind every
    loss conv whose loss took longer than those polls (profiles/r06_stream_query_gap.txt).  The checks are now made after 50 ms of
    waiting: a healthy run makes none (cmf_get_counter "liveness_checks", process-wide)."""
    rng = np.random.default_rng(3)
    for (N, T, K, L, n) in [(256, 20000, 5, 20, 300), (512, 6250, 32, 20, 60)]:
        data = rng.random((N, T))
        rule = cmf.MultUpdate(data, *cmf.init_rand(data, L=L, K=K, seed=1))
        try:
            rule.iterate(5)
            before = rule.counter("liveness_checks")
            rule.iterate(n)                       # pipelined: a wait per iteration
            for _ in range(20):                   # call by call: a wait per update_feature_maps!
                rule.update_motifs()
                rule.update_feature_maps()
            assert rule.counter("liveness_checks") == before
        finally:
            rule.close()
