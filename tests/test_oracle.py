"""CPU tests pinning the oracle itself.

The reference has no golden vectors for the MU path, so the
oracle is pinned by three independently written restatements agreeing with each
other (C, numpy per-lag GEMM, index-level brute force), by the algebraic
properties the reference's notebooks rely on, and by committed fixtures.
"""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rand_problem(rng, N, T, K, L):
    W = rng.random((K, N, L))
    H = rng.random((K, T))
    X = rng.random((N, T))
    return W, H, X


SHAPES = [
    (7, 23, 3, 4),
    (5, 9, 1, 1),     # K=1, L=1
    (1, 17, 2, 5),    # N=1
    (6, 3, 2, 5),     # T < L
    (4, 5, 3, 5),     # T == L
    (33, 70, 5, 10),
]


@pytest.mark.parametrize("N,T,K,L", SHAPES)
def test_conv_three_way(oracle, N, T, K, L):
    rng = np.random.default_rng(N * 1000 + T)
    W, H, _ = _rand_problem(rng, N, T, K, L)
    a = oracle.tensor_conv(W, H)
    b = oracle.c_tensor_conv(W, H)
    c = oracle.brute_conv(W, H)
    np.testing.assert_allclose(a, c, rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(b, c, rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("N,T,K,L", SHAPES)
def test_transconv_three_way(oracle, N, T, K, L):
    rng = np.random.default_rng(N * 1000 + T + 1)
    W, _, X = _rand_problem(rng, N, T, K, L)
    a = oracle.tensor_transconv(W, X)
    b = oracle.c_tensor_transconv(W, X)
    c = oracle.brute_transconv(W, X)
    np.testing.assert_allclose(a, c, rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(b, c, rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("N,T,K,L", SHAPES)
def test_hxt_matches_reference_slices(oracle, N, T, K, L):
    """mult.jl:31-34: numW[:,:,lag] = H[:, :T-lag] @ X[:, lag:]'."""
    rng = np.random.default_rng(7)
    _, H, X = _rand_problem(rng, N, T, K, L)
    out = oracle.c_hxt(H, X, L)
    for lag in range(L):
        ref = H[:, : T - lag] @ X[:, lag:].T if lag < T else np.zeros((K, N))
        np.testing.assert_allclose(out[:, :, lag], ref, rtol=1e-13, atol=1e-13)


def test_adjointness(oracle):
    """<conv(W,H), X> == <H, transconv(W,X)>."""
    rng = np.random.default_rng(11)
    W, H, X = _rand_problem(rng, 19, 57, 4, 6)
    lhs = np.sum(oracle.tensor_conv(W, H) * X)
    rhs = np.sum(H * oracle.tensor_transconv(W, X))
    assert abs(lhs - rhs) <= 1e-12 * abs(lhs)


def test_conv_equals_unfolded_gemm(oracle):
    """conv(W,H) = W_unfold * shift_and_stack(H, L) (common.jl:133-142; notebooks/benchmarks.ipynb tconv3)."""
    rng = np.random.default_rng(12)
    W, H, _ = _rand_problem(rng, 13, 41, 3, 7)
    K, N, L = W.shape
    W_unf = np.concatenate([W[:, :, l].T for l in range(L)], axis=1)  # N x (L*K), col = K*lag + k
    np.testing.assert_allclose(W_unf @ oracle.shift_and_stack(H, L), oracle.tensor_conv(W, H), rtol=1e-13, atol=1e-13)


def test_conv_equals_fft(oracle):
    """notebooks/test_fft.ipynb cells 6-7: direct conv == FFT conv to ~1e-15."""
    rng = np.random.default_rng(13)
    W, H, _ = _rand_problem(rng, 6, 64, 3, 9)
    K, N, L = W.shape
    T = H.shape[1]
    nfft = T + L
    est = np.zeros((N, T))
    Hf = np.fft.rfft(H, nfft, axis=1)
    for n in range(N):
        Wf = np.fft.rfft(W[:, n, :], nfft, axis=1)
        est[n] = np.fft.irfft((Wf * Hf).sum(0), nfft)[:T]
    np.testing.assert_allclose(est, oracle.tensor_conv(W, H), rtol=1e-12, atol=1e-12)


def test_converged_semantics(oracle):
    """model.jl:91-107."""
    import ctypes

    def both(lh, patience, tol):
        a = oracle.converged(lh, patience, tol)
        arr = np.asarray(lh, dtype=np.float64)
        b = bool(oracle.c_lib().oracle_converged(arr.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), len(arr), patience, tol))
        assert a == b
        return a

    assert not both([1.0, 1.0, 1.0], 3, 1e-4)            # length <= patience
    assert both([1.0, 1.0, 1.0, 1.0], 3, 1e-4)           # 3 diffs, all < tol
    assert not both([2.0, 1.0, 1.0, 1.0], 3, 1e-4)       # first of the 3 diffs too big
    assert both([5.0, 2.0, 1.0, 1.0, 1.0, 1.0], 3, 1e-4)
    assert both([1.0, 1.00005], 1, 1e-4)
    assert not both([1.0, 1.0002], 1, 1e-4)


@pytest.mark.parametrize("reg", [dict(), dict(l1W=0.1, l2W=0.5, l1H=0.1, l2H=0.2)])
def test_fit_c_vs_numpy(oracle, reg):
    data, _, _ = oracle.c_gen_synthetic(N=40, T=150, K=3, L=8, seed=1234)
    W0, H0 = oracle.c_init_rand(data, L=6, K=4, seed=0)
    Wa, Ha, la, ta = oracle.fit_mult(data, W0, H0, max_itr=25, check_convergence=False, **reg)
    Wb, Hb, lb, tb = oracle.c_fit_mult(data, W0, H0, max_itr=25, check_convergence=False, **reg)
    assert len(la) == len(lb) == 26 and ta[0] == 0.0 and tb[0] == 0.0
    np.testing.assert_allclose(la, lb, rtol=1e-10)
    np.testing.assert_allclose(Wa, Wb, rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(Ha, Hb, rtol=1e-8, atol=1e-12)


def test_unregularised_loss_nonincreasing(oracle):
    data, _, _ = oracle.c_gen_synthetic(N=30, T=200, K=3, L=8, seed=5)
    W0, H0 = oracle.c_init_rand(data, L=8, K=3, seed=1)
    _, _, lh, _ = oracle.fit_mult(data, W0, H0, max_itr=40, check_convergence=False)
    assert np.all(np.diff(lh) <= 1e-12)


def test_exactly_factorisable_drives_loss_down(oracle):
    """datasets/toy.jl-style noiseless data: MU should fit it well."""
    data, Wt, Ht = oracle.c_gen_synthetic(N=20, T=300, K=2, L=5, noise_scale=0.0, seed=3)
    np.testing.assert_allclose(data, oracle.tensor_conv(Wt, Ht), rtol=1e-12, atol=1e-12)
    W0, H0 = oracle.c_init_rand(data, L=5, K=2, seed=2)
    _, _, lh, _ = oracle.fit_mult(data, W0, H0, max_itr=300, check_convergence=False)
    assert lh[-1] < 0.25 * lh[0]


def test_early_stop_and_eval_mode(oracle):
    data, _, _ = oracle.c_gen_synthetic(N=25, T=120, K=3, L=6, seed=9)
    W0, H0 = oracle.c_init_rand(data, L=6, K=3, seed=0)
    Wa, Ha, la, _ = oracle.fit_mult(data, W0, H0, max_itr=500, tol=1e-3, patience=2)
    Wb, Hb, lb, _ = oracle.c_fit_mult(data, W0, H0, max_itr=500, tol=1e-3, patience=2)
    assert len(la) == len(lb) < 501
    np.testing.assert_allclose(la, lb, rtol=1e-10)
    # eval_mode: W untouched (alternating.jl:51-53)
    Wc, Hc, lc, _ = oracle.c_fit_mult(data, W0, H0, max_itr=5, eval_mode=True, check_convergence=False)
    np.testing.assert_array_equal(Wc, W0)
    assert lc[-1] < lc[0]


def test_init_rand_scale_is_least_squares(oracle):
    """model.jl:113-125: after scaling by sqrt|alpha| each, <data - est, est> == 0."""
    data, _, _ = oracle.c_gen_synthetic(N=30, T=100, K=3, L=7, seed=21)
    W, H = oracle.c_init_rand(data, L=5, K=4, seed=0)
    est = oracle.tensor_conv(W, H)
    assert abs(np.sum((data - est) * est)) <= 1e-9 * np.sum(est * est)
    assert W.min() >= 0 and H.min() >= 0


def test_gen_synthetic_statistics(oracle):
    """datasets/synthetic.jl:29-61 semantics."""
    data, W, H = oracle.c_gen_synthetic(N=200, T=4000, K=3, L=20, seed=1234)
    assert data.shape == (200, 4000) and data.min() >= 0.0
    frac = np.mean(H > 0)
    assert 0.45 < frac < 0.55                       # Bernoulli(p_h=0.5)
    assert 0.9 < H[H > 0].mean() < 1.1              # Exponential(1)
    # W[k,n,:] = mW[n,k] * pdf(Normal(cent, sigma)) on linspace(-1,1,L); Dirichlet rows sum to 1
    assert np.all(W >= 0)
    resid = data - np.maximum(0.0, oracle.tensor_conv(W, H))
    assert abs(resid.std() - 1.0) < 0.2             # noise_scale=1 (rectified, so loose)


def test_rng_known_answers(oracle):
    """Known answers of the portable RNG spec (pins it across rebuilds / platforms)."""
    lib = oracle.c_lib()
    got = [lib.oracle_rng_u01(1234, 0, i) for i in range(4)]
    path = os.path.join(GOLDEN, "rng_kat.npz")
    kat = np.load(path)
    np.testing.assert_array_equal(np.asarray(got), kat["u01_seed1234_stream0"][:4])
    gotn = [lib.oracle_rng_normal(1234, 16, i) for i in range(4)]
    np.testing.assert_allclose(np.asarray(gotn), kat["normal_seed1234_stream16"][:4], rtol=1e-14)


@pytest.mark.parametrize("name,rule", [("hals_small", "hals"), ("pgd_small", "pgd")])
def test_golden_fixtures_hals_pgd(oracle, name, rule):
    """Both restatements of the HALS and PGD rules reproduce their committed fixtures."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    assert str(g["rule"]) == rule
    n = int(g["max_itr"])
    if rule == "hals":
        fits = (lambda: oracle.c_fit_hals(g["data"], g["W0"], g["H0"], max_itr=n, check_convergence=False),
                lambda: oracle.fit_hals(g["data"], g["W0"], g["H0"], max_itr=n, check_convergence=False))
    else:
        fits = (lambda: oracle.fit_pgd(g["data"], g["W0"], g["H0"], max_itr=n),
                lambda: oracle.c_fit_pgd(g["data"], g["W0"], g["H0"], max_itr=n))
    for fit in fits:
        W, H, lh, _ = fit()
        np.testing.assert_allclose(lh, g["loss_hist"], rtol=1e-9)
        np.testing.assert_allclose(W, g["W"], rtol=1e-7, atol=1e-11)
        np.testing.assert_allclose(H, g["H"], rtol=1e-7, atol=1e-11)


@pytest.mark.parametrize("name", ["mu_small", "mu_small_reg", "mu_k5"])
def test_golden_fixtures(oracle, name):
    """Both restatements reproduce the committed fixtures (tests/golden/make_golden.py)."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    kw = dict(max_itr=int(g["max_itr"]), check_convergence=False,
              l1W=float(g["l1W"]), l2W=float(g["l2W"]), l1H=float(g["l1H"]), l2H=float(g["l2H"]))
    for fit in (oracle.fit_mult, oracle.c_fit_mult):
        W, H, lh, _ = fit(g["data"], g["W0"], g["H0"], **kw)
        np.testing.assert_allclose(lh, g["loss_hist"], rtol=1e-9)
        np.testing.assert_allclose(W, g["W"], rtol=1e-7, atol=1e-12)
        np.testing.assert_allclose(H, g["H"], rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(oracle.tensor_conv(g["W0"], g["H0"]), g["conv0"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(oracle.tensor_transconv(g["W0"], g["data"]), g["transconv0"], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("reg", [dict(), dict(l1W=0.1, l2W=0.5, l1H=0.1, l2H=0.2)])
def test_hals_c_vs_numpy(oracle, reg):
    """hals.jl restated twice (literal residual form), plus the Gram-projected form the GPU kernels use."""
    data, _, _ = oracle.c_gen_synthetic(N=30, T=120, K=3, L=8, seed=7)
    W0, H0 = oracle.c_init_rand(data, L=6, K=4, seed=0)
    Wa, Ha, la, _ = oracle.fit_hals(data, W0, H0, max_itr=6, check_convergence=False, **reg)
    Wb, Hb, lb, _ = oracle.c_fit_hals(data, W0, H0, max_itr=6, check_convergence=False, **reg)
    np.testing.assert_allclose(la, lb, rtol=1e-11)
    np.testing.assert_allclose(Wa, Wb, rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(Ha, Hb, rtol=1e-9, atol=1e-13)
    assert np.all(np.diff(la) <= 1e-12) and Wa.min() >= 0.0 and Ha.min() >= 0.0
    assert np.mean(Ha == 0.0) > 0.05  # HALS produces exact zeros (clamp at 0)


def test_hals_gram_form_equals_residual_form(oracle):
    """The algebra behind the GPU HALS kernels: sweeping on G = resid*H_unfold' / P = transconv(W, resid)
    with Gram updates reproduces the reference's residual sweeps exactly (incl. the right-edge truncation)."""
    from hals_gram_form import h_sweep, h_sweep_pull_pipeline, w_sweep

    data, _, _ = oracle.c_gen_synthetic(N=12, T=40, K=3, L=6, seed=3)
    W0, H0 = oracle.c_init_rand(data, L=5, K=3, seed=0)
    for l1, l2 in ((0.0, 0.0), (0.1, 0.3)):
        W, H = W0.copy(), H0.copy()
        rule = oracle.HALSUpdate(data, W, H)
        oracle.hals_update_motifs(rule, data, W, H, l1W=l1, l2W=l2)
        np.testing.assert_allclose(w_sweep(oracle, W0, H0, data, l1, l2), W, rtol=1e-10, atol=1e-13)
        Hn = H.copy()
        oracle.hals_update_feature_maps(rule, data, W, Hn, l1H=l1, l2H=l2)
        np.testing.assert_allclose(h_sweep(oracle, W, H0, data, l1, l2), Hn, rtol=1e-10, atol=1e-13)
        # ... and so does the pull form of the persistent pipeline, in whatever order its flags let the rows advance
        for seed in (0, 1, 2):
            np.testing.assert_allclose(h_sweep_pull_pipeline(oracle, W, H0, data, l1, l2, block=8, seed=seed), Hn, rtol=1e-10, atol=1e-13)


def test_hals_hh_from_lag_correlations(oracle):
    """The Gram matrix of H_unfold assembled from the K x K lag correlations of H (what compute_hh does on the GPU with one
    C2 contraction on K columns) is H_unfold * H_unfold' (hals.jl:56-60), including the truncation at the right end and
    lags beyond T."""
    from hals_gram_form import hh_from_lag_correlations

    rng = np.random.default_rng(1)
    for K, T, L in ((3, 40, 6), (2, 5, 8), (4, 9, 1)):
        H = rng.uniform(0, 1, (K, T)) * (rng.uniform(size=(K, T)) > 0.3)
        Hu = oracle.shift_and_stack(H, L)  # row l*K + k = H[k] shifted right by l
        np.testing.assert_allclose(hh_from_lag_correlations(oracle, H, L), Hu @ Hu.T, rtol=1e-12, atol=1e-13)


def test_pgd_masked_loss_gradient_and_reduction(oracle):
    """MaskedLoss(SquareLoss(), mask) in the PGD restatement (pgd.jl:58-70): with an all-ones mask it is the plain
    SquareLoss; the direction pgd! takes is the gradient of eval(MaskedLoss) (finite differences)."""
    rng = np.random.default_rng(0)
    N, T, K, L = 7, 40, 3, 4
    W = rng.uniform(0.1, 1, (K, N, L))
    H = rng.uniform(0.1, 1, (K, T))
    data = rng.uniform(0, 2, (N, T))
    a = oracle.fit_pgd(data, W, H, max_itr=5)
    b = oracle.fit_pgd(data, W, H, max_itr=5, mask=np.ones((N, T)))
    np.testing.assert_allclose(a[2], b[2], rtol=1e-13)
    np.testing.assert_allclose(a[0], b[0], rtol=1e-13)
    mask = (rng.uniform(size=(N, T)) > 0.4) * rng.uniform(0.5, 1.5, (N, T))

    def J(Wx, Hx):
        return np.linalg.norm(mask * data - mask * oracle.tensor_conv(Wx, Hx)) ** 2

    # one unpenalised, unconstrained step with a tiny step size moves along -grad J / ||grad J||
    W1, H1 = W.copy(), H.copy()
    rule = oracle.PGDUpdate(data, W1, H1)
    rule.stepW = 1e-6
    oracle.pgd_update_motifs(rule, data, W1, H1, penaltiesW_sq=(), nonneg=False, mask=mask)
    e = 1e-6
    # the reference's mask enters the gradient once (grad .*= mask) but the loss twice (mask.^2): equal for 0/1
    # masks; for a real-valued mask the step follows sum(mask * 2 * (est - data) ...), so check with mask in {0,1}
    mask01 = (mask > 0).astype(float)

    def J01(Wx):
        return np.linalg.norm(mask01 * data - mask01 * oracle.tensor_conv(Wx, H)) ** 2

    g01 = np.zeros_like(W)
    for idx in np.ndindex(*W.shape):
        Wp = W.copy(); Wp[idx] += e
        Wm = W.copy(); Wm[idx] -= e
        g01[idx] = (J01(Wp) - J01(Wm)) / (2 * e)
    W2, H2 = W.copy(), H.copy()
    rule2 = oracle.PGDUpdate(data, W2, H2)
    rule2.stepW = 1e-6
    oracle.pgd_update_motifs(rule2, data, W2, H2, penaltiesW_sq=(), nonneg=False, mask=mask01)
    step = W2 - W
    np.testing.assert_allclose(step, -1e-6 * g01 / np.linalg.norm(g01), rtol=1e-5, atol=1e-13)
    assert rule2.cur_loss == pytest.approx(J01(W2), rel=1e-12)
    assert rule.cur_loss == pytest.approx(J(W1, H1), rel=1e-12)


@pytest.mark.parametrize("loss,constrW,constrH,masked", [("square", "nonneg", "nonneg", False), ("square", "unitnorm", "nonneg", False),
                                                          ("abs", "nonneg", "nonneg", False), ("square", None, "unitnorm", True),
                                                          ("abs", "unitnorm", "nonneg", True)])
def test_pgd_two_restatements_agree(oracle, loss, constrW, constrH, masked):
    """PGD has two independent restatements like MU and HALS: numpy (cmf_oracle.py) and C (oracle_fit_pgd), covering
    SquareLoss / AbsoluteLoss (pgd.jl:29-47), MaskedLoss (:58-70), both penalties (:73-89), NonnegConstraint (:92-96) and
    UnitNormConstraint (:100-110)."""
    N, T, K, L, iters = 14, 60, 3, 5, 6
    rng = np.random.default_rng(3)
    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=2, L=L, seed=5)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=1)
    mask = (rng.random((N, T)) < 0.8).astype(float) if masked else None
    Wa, Ha, la, sa = oracle.fit_pgd(data, W0, H0, max_itr=iters, penaltiesW_sq=(0.7,), penaltiesW_abs=(0.2,), penaltiesH_sq=(0.1,),
                                    penaltiesH_abs=(0.3,), loss=loss, constrW=constrW, constrH=constrH, mask=mask)
    Wb, Hb, lb, sb = oracle.c_fit_pgd(data, W0, H0, max_itr=iters, penW_sq=0.7, penW_abs=0.2, penH_sq=0.1, penH_abs=0.3,
                                      loss=loss, constrW=constrW, constrH=constrH, mask=mask)
    np.testing.assert_allclose(la, lb, rtol=1e-10)
    np.testing.assert_allclose(Wa, Wb, rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(Ha, Hb, rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(sa, sb, rtol=1e-12)  # the same accept / reject decisions
    if constrW == "unitnorm":  # every component's motif has norm <= 1 (and some were actually scaled)
        nrm = np.array([np.linalg.norm(Wa[k]) for k in range(K)])
        assert np.all(nrm <= 1 + 1e-12)


def test_unit_norm_projection_and_absolute_loss_statements(oracle):
    """UnitNormConstraint scales only the slices (components k) whose norm exceeds 1 (pgd.jl:100-110); AbsoluteLoss is
    the l1 norm with the sign gradient (pgd.jl:41-47)."""
    x = np.array([[3.0, 4.0], [0.3, 0.4]])  # component 0: norm 5 -> scaled; component 1: norm 0.5 -> kept
    y = x.copy()
    oracle.unit_norm_projection(y)
    np.testing.assert_allclose(y, [[0.6, 0.8], [0.3, 0.4]])
    t = np.arange(24.0).reshape(2, 3, 4) / 10
    z = t.copy()
    oracle.unit_norm_projection(z)
    for k in range(2):
        n = np.linalg.norm(t[k])
        np.testing.assert_allclose(z[k], t[k] / n if n > 1 else t[k])
