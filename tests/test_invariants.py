"""Closed-form properties of the update rules themselves (mult.jl:23-58, hals.jl:90-154, pgd.jl:158-255) that hold for the
reference's algorithm whatever the arithmetic underneath: they pin the oracle (CPU) and the HIP path (GPU) to the
ALGORITHM without going through each other -- the reference holds no vectors for this path and cannot run here, so
reference-independent known answers are the strongest pins there are.

  * fixed point: if data = tensor_conv(W, H) exactly, num = denom in both MU updates, so W and H stay where they are
    (mult.jl:37,51: x * (num / (denom + eps))) and the loss is 0;
  * scale equivariance: fit(c * data, c * W0, H0) = (c * W, H) with the same loss_hist (every quantity in mult.jl:28-57 is
    homogeneous of the right degree; eps terms enter at 1e-16);
  * component permutation: permuting the K components of W0 and H0 permutes the result (nothing in the rules orders k,
    except the HALS sweeps, whose visiting order is part of the algorithm: not tested for HALS);
  * unit permutation: permuting the N rows of data and of W0 permutes W, leaves H and loss_hist (sums over n commute up to
    rounding);
  * K = 1, L = 1 is plain rank-1 NMF with its textbook updates  w <- w * (X h) / (w h'h),  h <- h * (w'X) / (w'w h).
"""
import numpy as np
import pytest


def _problem(oracle, N=30, T=200, K=3, L=6, seed=11):
    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=min(L, 20) if L > 1 else 2, seed=seed)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=seed + 1)
    return data, W0, H0


def _fits(oracle):
    """(name, fit) pairs: the oracle's two restatements, and the HIP path when a GPU is there."""
    fits = [("numpy", lambda d, W, H, n, **kw: oracle.fit_mult(d, W, H, max_itr=n, check_convergence=False, **kw)[:3]),
            ("c", lambda d, W, H, n, **kw: oracle.c_fit_mult(d, W, H, max_itr=n, check_convergence=False, **kw)[:3])]
    return fits


def _hip_fit():
    import cmf_jl_amd as cmf

    if cmf.load_library().cmf_device_count() < 1:
        pytest.skip("no HIP device")

    def fit(d, W, H, n, **kw):
        rule = cmf.MultUpdate(d, W, H)
        ls = [rule.compute_loss()] + list(rule.iterate(n, **kw))
        Wg, Hg = rule.download()
        rule.close()
        return Wg, Hg, np.asarray(ls)

    return fit


def _rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def _check_all(fit, oracle, tol):
    # fixed point
    rng = np.random.default_rng(0)
    W = rng.random((3, 20, 5)) + 0.1
    H = rng.random((3, 150)) + 0.1
    data = oracle.tensor_conv(W, H)
    W1, H1, ls = fit(data, W, H, 3)
    assert ls.max() < max(tol, 1e-12) * 10 and _rel(W1, W) < tol and _rel(H1, H) < tol
    # scale equivariance
    data, W0, H0 = _problem(oracle)
    Wa, Ha, la = fit(data, W0, H0, 5)
    c = 7.5
    Wb, Hb, lb = fit(c * data, c * W0, H0, 5)
    np.testing.assert_allclose(lb, la, rtol=tol)
    assert _rel(Wb, c * Wa) < tol and _rel(Hb, Ha) < tol
    # component permutation
    perm = np.array([2, 0, 1])
    Wc, Hc, lc = fit(data, W0[perm], H0[perm], 5)
    np.testing.assert_allclose(lc, la, rtol=tol)
    assert _rel(Wc, Wa[perm]) < tol and _rel(Hc, Ha[perm]) < tol
    # unit permutation
    pn = np.random.default_rng(1).permutation(data.shape[0])
    Wd, Hd, ld = fit(data[pn], W0[:, pn, :], H0, 5)
    np.testing.assert_allclose(ld, la, rtol=tol)
    assert _rel(Wd, Wa[:, pn, :]) < tol and _rel(Hd, Ha) < tol
    # K = 1, L = 1: rank-1 NMF, textbook multiplicative updates (eps as in mult.jl:37-38,51-52)
    X = np.abs(np.random.default_rng(2).normal(size=(12, 40))) + 0.05
    w = np.random.default_rng(3).random(12) + 0.1
    h = np.random.default_rng(4).random(40) + 0.1
    eps = np.finfo(np.float64).eps
    W1, H1, l1 = fit(X, w.reshape(1, 12, 1), h.reshape(1, 40), 4)
    ls = [np.linalg.norm(np.outer(w, h) - X) / np.linalg.norm(X)]
    for _ in range(4):
        w = np.maximum(eps, w * ((X @ h) / (w * (h @ h) + eps)))
        h = np.maximum(eps, h * ((w @ X) / ((w @ w) * h + eps)))
        ls.append(np.linalg.norm(np.outer(w, h) - X) / np.linalg.norm(X))
    np.testing.assert_allclose(l1, ls, rtol=tol)
    assert _rel(W1[0, :, 0], w) < tol and _rel(H1[0], h) < tol


def test_mu_invariants_oracle(oracle):
    for name, fit in _fits(oracle):
        _check_all(fit, oracle, 1e-9)


@pytest.mark.gpu
def test_mu_invariants_hip(oracle):
    _check_all(_hip_fit(), oracle, 2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("devices,gram", [([0, 0, 0], 0), (None, 1), ([0, 0], 1)])
def test_mu_invariants_hip_groups_and_gram(oracle, devices, gram):
    """The same known answers through the T-sharded group and the Gram form."""
    import cmf_jl_amd as cmf

    if cmf.load_library().cmf_device_count() < 1:
        pytest.skip("no HIP device")

    def fit(d, W, H, n, **kw):
        if devices is not None and d.shape[1] < 4 * W.shape[2] * len(devices):
            rule = cmf.MultUpdate(d, W, H)  # too short to shard (every shard needs L-1 columns, the Gram form T >= 4 L)
        else:
            rule = cmf.MultUpdate(d, W, H, devices=devices)
        if gram:
            rule.set_option("gram", gram)
        ls = [rule.compute_loss()] + list(rule.iterate(n, **kw))
        Wg, Hg = rule.download()
        rule.close()
        return Wg, Hg, np.asarray(ls)

    _check_all(fit, oracle, 5e-5)


def test_pgd_and_hals_fixed_point_oracle(oracle):
    """data = tensor_conv(W, H): the residual is 0, so HALS leaves W and H where they are (hals.jl:104-110: the update of a
    column whose residual projection vanishes is the column itself) and PGD's gradient is the penalty term alone."""
    rng = np.random.default_rng(0)
    W = rng.random((3, 20, 5)) + 0.1
    H = rng.random((3, 150)) + 0.1
    data = oracle.tensor_conv(W, H)
    Wh, Hh, lh, _ = oracle.c_fit_hals(data, W, H, max_itr=2, check_convergence=False)
    assert lh.max() < 1e-12 and _rel(Wh, W) < 1e-10 and _rel(Hh, H) < 1e-10
    Wp, Hp, lp, _ = oracle.fit_pgd(data, W, H, max_itr=1, penaltiesW_sq=(), penaltiesW_abs=())
    assert lp[0] < 1e-12
