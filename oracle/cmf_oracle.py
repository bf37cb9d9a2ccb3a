"""cmf_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

NumPy fp64 restatement of CMF.jl's convolutive-NMF multiplicative-update
path, written independently of oracle/cmf_oracle.c so the two cross-check
each other.  It keeps the reference's structure (one GEMM per lag with
beta=1 accumulation), so it is also the closest stand-in for
"Julia + OpenBLAS" and is what bench.py times as ``cpu_baseline``
(kind "port").

PARITY UNPINNED BY THE REFERENCE: the reference holds no golden vectors or
assertions for this path and Julia cannot run in the build container; see
oracle/cmf_oracle.c's header and DESIGN.md.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.

Array conventions are Julia's: ``W[k, n, l]`` (K,N,L), ``H[k, t]`` (K,T),
``data[n, t]`` (N,T).  Citations are relative to the reference checkout.
"""
from __future__ import annotations

import ctypes
import os
import time

import numpy as np

EPS = float(np.finfo(np.float64).eps)  # eps(): src/CMF.jl:20, src/algs/mult.jl:37-38


# --------------------------------------------------------------------------
# convolution primitives
# --------------------------------------------------------------------------
def tensor_conv(W, H, out=None):
    """tensor_conv!: src/common.jl:24-34 (s_dot! :108-118).

    est = 0; for lag: est[:, lag:] += W[:, :, lag]' @ H[:, :T-lag]
    """
    K, N, L = W.shape
    T = H.shape[1]
    est = np.zeros((N, T)) if out is None else out
    est[...] = 0.0
    for lag in range(min(L, T)):
        est[:, lag:] += W[:, :, lag].T @ H[:, : T - lag]
    return est


def tensor_transconv(W, X, out=None):
    """tensor_transconv!: src/common.jl:71-81.

    out = 0; for lag: out[:, :T-lag] += W[:, :, lag] @ X[:, lag:]
    """
    K, N, L = W.shape
    T = X.shape[1]
    res = np.zeros((K, T)) if out is None else out
    res[...] = 0.0
    for lag in range(min(L, T)):
        res[:, : T - lag] += W[:, :, lag] @ X[:, lag:]
    return res


def shift_and_stack(H, L):
    """shift_and_stack: src/common.jl:133-142 (used for a property test)."""
    K, T = H.shape
    out = np.zeros((L * K, T))
    for lag in range(min(L, T)):
        out[K * lag : K * (lag + 1), lag:] = H[:, : T - lag]
    return out


def compute_loss(data, W, H):
    """compute_loss: src/common.jl:54-59."""
    return np.linalg.norm(tensor_conv(W, H) - data) / np.linalg.norm(data)


# --------------------------------------------------------------------------
# MU rule
# --------------------------------------------------------------------------
class MultUpdate:
    """MultUpdate state + ctor: src/algs/mult.jl:1-20."""

    def __init__(self, data, W, H):
        self.resids = tensor_conv(W, H) - data
        self.data_norm = np.linalg.norm(data)
        self.numW = np.zeros(W.shape)
        self.denomW = np.zeros(W.shape)
        self.numH = np.zeros(H.shape)
        self.denomH = np.zeros(H.shape)
        self.est = np.zeros(data.shape)


def update_motifs(rule, data, W, H, l1W=0.0, l2W=0.0):
    """update_motifs!(::MultUpdate): src/algs/mult.jl:23-39.  W in place."""
    K, N, L = W.shape
    T = H.shape[1]
    tensor_conv(W, H, out=rule.est)  # :28
    for lag in range(L):  # :31-34
        if lag < T:
            rule.numW[:, :, lag] = H[:, : T - lag] @ data[:, lag:].T
            rule.denomW[:, :, lag] = H[:, : T - lag] @ rule.est[:, lag:].T
        else:
            rule.numW[:, :, lag] = 0.0
            rule.denomW[:, :, lag] = 0.0
    # :37  W *= numW / (denomW + l1W + 2*l2W*W + eps())   (left-fold sums)
    den = ((rule.denomW + l1W) + (2.0 * l2W) * W) + EPS
    W *= rule.numW / den
    np.maximum(W, EPS, out=W)  # :38
    return W


def update_feature_maps(rule, data, W, H, l1H=0.0, l2H=0.0):
    """update_feature_maps!(::MultUpdate): src/algs/mult.jl:42-58.  H in place; returns loss."""
    tensor_conv(W, H, out=rule.est)  # :44
    tensor_transconv(W, data, out=rule.numH)  # :47
    tensor_transconv(W, rule.est, out=rule.denomH)  # :48
    den = ((rule.denomH + l1H) + (2.0 * l2H) * H) + EPS  # :51
    H *= rule.numH / den
    np.maximum(H, EPS, out=H)  # :52
    tensor_conv(W, H, out=rule.est)  # :55
    np.subtract(rule.est, data, out=rule.resids)  # :56
    return np.linalg.norm(rule.resids) / rule.data_norm  # :57


def converged(loss_hist, patience, tol):
    """converged: src/model.jl:91-107."""
    if len(loss_hist) <= patience:
        return False
    d = np.diff(np.asarray(loss_hist[len(loss_hist) - patience - 1 :]))
    return bool(np.all(np.abs(d) < tol))


def fit_mult(data, W_init, H_init, max_itr=100, max_time=np.inf, check_convergence=True,
             patience=3, tol=1e-4, eval_mode=False, l1W=0.0, l2W=0.0, l1H=0.0, l2H=0.0):
    """fit(::AlternatingOptimizer{MultUpdate}): src/algs/alternating.jl:16-71.

    Returns (W, H, loss_hist, time_hist)."""
    assert patience >= 1  # :30
    W = np.array(W_init, dtype=np.float64, copy=True)  # :33-34 deepcopy
    H = np.array(H_init, dtype=np.float64, copy=True)
    rule = MultUpdate(data, W, H)
    loss_hist = [compute_loss(data, W, H)]  # :37
    time_hist = [0.0]  # :38
    itr = 1
    while itr <= max_itr and time_hist[-1] <= max_time:  # :45
        itr += 1
        t0 = time.time()
        if not eval_mode:  # :51-53
            update_motifs(rule, data, W, H, l1W=l1W, l2W=l2W)
        loss = update_feature_maps(rule, data, W, H, l1H=l1H, l2H=l2H)  # :54
        time_hist.append(time_hist[-1] + (time.time() - t0))  # :57-58
        loss_hist.append(loss)  # :59
        if check_convergence and converged(loss_hist, patience, tol):  # :63-66
            break
    return W, H, np.asarray(loss_hist), np.asarray(time_hist)


# --------------------------------------------------------------------------
# HALS rule (BASELINE config 5): src/algs/hals.jl
# --------------------------------------------------------------------------
class HALSUpdate:
    """HALSUpdate state + ctor: src/algs/hals.jl:6-28 (resids is carried across iterations)."""

    def __init__(self, data, W, H):
        self.resids = tensor_conv(W, H) - data
        self.data_norm = np.linalg.norm(data)


def hals_update_motifs(rule, data, W, H, l1W=0.0, l2W=0.0):
    """update_motifs!(::HALSUpdate): hals.jl:31-34, :53-61, :90-112.  W and rule.resids in place."""
    K, N, L = W.shape
    H_unfold = shift_and_stack(H, L)  # :56  row = K*lag + k
    H_norms = np.linalg.norm(H_unfold, axis=1)  # :57-60
    R = rule.resids
    for k in range(K):  # :92  k outer
        for l in range(L):  # :93  lag inner
            ind = l * K + k  # :102 (0-based)
            h = H_unfold[ind]
            R -= np.outer(W[k, :, l], h)  # :104
            # :110 _next_W_col: max.((-resid * Hkl .- l1_W) ./ (norm_Hkl^2 + EPSILON + l2_W), 0.0)
            W[k, :, l] = np.maximum((-(R @ h) - l1W) / (H_norms[ind] ** 2 + EPS + l2W), 0.0)
            R += np.outer(W[k, :, l], h)  # :106
    return W


def hals_update_feature_maps(rule, data, W, H, l1H=0.0, l2H=0.0):
    """update_feature_maps!(::HALSUpdate): hals.jl:37-42, :64-80, :121-154.  H and rule.resids in place."""
    K, N, L = W.shape
    T = H.shape[1]
    W_norms = np.linalg.norm(W, axis=1)  # :67-72  (K, L)
    R = rule.resids
    for k in range(K):  # :124
        Wk = W[k]  # N x L (:76-79)
        for t in range(T):  # :125
            Lt = min(T - t, L)  # :136 (1-based min(T-t+1, L))
            norm_Wkt = np.linalg.norm(W_norms[k, :Lt])  # :136
            rem = R[:, t: t + Lt]
            rem += (-H[k, t]) * Wk[:, :Lt]  # :139-140
            trace = np.sum(Wk[:, :Lt] * (-rem))  # :152
            H[k, t] = max((trace - l1H) / (norm_Wkt ** 2 + EPS + l2H), 0.0)  # :153
            rem += H[k, t] * Wk[:, :Lt]  # :146
    return np.linalg.norm(R) / rule.data_norm  # :41


def fit_hals(data, W_init, H_init, max_itr=100, max_time=np.inf, check_convergence=True,
             patience=3, tol=1e-4, eval_mode=False, l1W=0.0, l2W=0.0, l1H=0.0, l2H=0.0):
    """fit(::AlternatingOptimizer{HALSUpdate}): src/algs/alternating.jl:16-71."""
    assert patience >= 1
    W = np.array(W_init, dtype=np.float64, copy=True)
    H = np.array(H_init, dtype=np.float64, copy=True)
    rule = HALSUpdate(data, W, H)
    loss_hist = [compute_loss(data, W, H)]
    time_hist = [0.0]
    itr = 1
    while itr <= max_itr and time_hist[-1] <= max_time:
        itr += 1
        t0 = time.time()
        if not eval_mode:
            hals_update_motifs(rule, data, W, H, l1W=l1W, l2W=l2W)
        loss = hals_update_feature_maps(rule, data, W, H, l1H=l1H, l2H=l2H)
        time_hist.append(time_hist[-1] + (time.time() - t0))
        loss_hist.append(loss)
        if check_convergence and converged(loss_hist, patience, tol):
            break
    return W, H, np.asarray(loss_hist), np.asarray(time_hist)


# --------------------------------------------------------------------------
# PGD rule: src/algs/pgd.jl.  numpy restatement here, C restatement in
# cmf_oracle.c (oracle_fit_pgd / c_fit_pgd below); the two are cross-checked in tests/test_oracle.py.
# Pieces: SquareLoss (:30-36), AbsoluteLoss (:41-47; `loss="abs"`), MaskedLoss(loss, mask) (:58-70; `mask=`),
# SquarePenalty / AbsolutePenalty (:74-89), NonnegConstraint (:92-96), UnitNormConstraint (:100-110;
# `constr="unitnorm"`) or no constraint.
# --------------------------------------------------------------------------
class PGDUpdate:
    """PGDUpdate state + ctor: src/algs/pgd.jl:112-155."""

    def __init__(self, data, W, H):
        self.datanorm = np.linalg.norm(data)
        self.est = tensor_conv(W, H)
        self.gradW = np.zeros(W.shape)
        self.gradH = np.zeros(H.shape)
        self.stepW = 5.0
        self.stepH = 5.0
        self.cur_loss = self.datanorm  # :151 (sic: the norm, not its square)
        self.step_incr = 1.05
        self.step_decr = 0.70


def _constr(nonneg, constr):
    """The constraint of one factor: `constr` ("nonneg" | "unitnorm" | None) when given, else the legacy `nonneg` flag."""
    if constr == "default":
        return "nonneg" if nonneg else None
    return constr


def unit_norm_projection(x):
    """projection!(::UnitNormConstraint, x): src/algs/pgd.jl:100-110 -- every slice along the FIRST dimension (a
    component k: W[k, :, :] or H[k, :]) with norm > 1 is scaled to norm 1."""
    for m in range(x.shape[0]):
        mag = np.linalg.norm(x[m])
        if mag > 1:
            x[m] /= mag


def _pgd(rule, x, gradx, compute_grad, step, data, W, H, pen_sq, pen_abs, nonneg, mask=None, loss="square", constr="default"):
    """pgd!: src/algs/pgd.jl:224-255 (x is W or H, updated in place)."""
    T = H.shape[1]
    constr = _constr(nonneg, constr)
    if loss == "abs":
        rule.est[...] = np.sign(rule.est - data)  # :42-44  grad!(AbsoluteLoss): sign(est - data)
    else:
        rule.est[...] = 2.0 * (rule.est - data)  # :230  grad!(SquareLoss): 2*(est - data), in place in r.est
    if mask is not None:
        rule.est *= mask  # :64-67  grad!(MaskedLoss): grad .*= mask
    compute_grad(gradx, rule.est)  # :231
    for w in pen_sq:
        gradx += 2.0 * w * x  # :78-80
    for w in pen_abs:
        gradx += w * np.sign(x)  # :87-89
    alpha = step / (np.linalg.norm(gradx) + EPS)  # :237
    x -= alpha * gradx  # :240
    if constr == "nonneg":
        np.maximum(x, EPS, out=x)  # :94-96 max(eps(), x)
    elif constr == "unitnorm":
        unit_norm_projection(x)  # :100-110
    tensor_conv(W, H, out=rule.est)  # :245
    b, e = (data, rule.est) if mask is None else (mask * data, mask * rule.est)  # :68-70 eval(MaskedLoss)
    if loss == "abs":
        lossv = np.abs(b - e).sum()  # :45-47 norm(b - est, 1)
    else:
        lossv = np.linalg.norm(b - e) ** 2  # :246, :34-36
    step = step * (rule.step_incr if lossv < rule.cur_loss else rule.step_decr)  # :248-252
    rule.cur_loss = lossv  # :253
    return step


def pgd_update_motifs(rule, data, W, H, penaltiesW_sq=(1.0,), penaltiesW_abs=(), nonneg=True, mask=None, loss="square", constrW="default"):
    """update_motifs!(::PGDUpdate): pgd.jl:158-177 (defaults: SquarePenalty(1), NonnegConstraint)."""
    K, N, L = W.shape
    T = H.shape[1]

    def grad(gradw, est):  # compute_gradW!: :206-214
        for lag in range(L):
            gradw[:, :, lag] = (H[:, : T - lag] @ est[:, lag:].T) if lag < T else 0.0

    rule.stepW = _pgd(rule, W, rule.gradW, grad, rule.stepW, data, W, H, penaltiesW_sq, penaltiesW_abs, nonneg, mask, loss, constrW)


def pgd_update_feature_maps(rule, data, W, H, penaltiesH_sq=(), penaltiesH_abs=(), nonneg=True, mask=None, loss="square", constrH="default"):
    """update_feature_maps!(::PGDUpdate): pgd.jl:180-202 -> sqrt(cur_loss / datanorm^2)."""

    def grad(gradh, est):  # compute_gradH!: :218-221
        tensor_transconv(W, est, out=gradh)

    rule.stepH = _pgd(rule, H, rule.gradH, grad, rule.stepH, data, W, H, penaltiesH_sq, penaltiesH_abs, nonneg, mask, loss, constrH)
    return np.sqrt(rule.cur_loss / rule.datanorm ** 2)


def fit_pgd(data, W_init, H_init, max_itr=100, check_convergence=False, patience=3, tol=1e-4, eval_mode=False, **kw):
    """fit(::AlternatingOptimizer{PGDUpdate}): src/algs/alternating.jl:16-71."""
    W = np.array(W_init, dtype=np.float64, copy=True)
    H = np.array(H_init, dtype=np.float64, copy=True)
    rule = PGDUpdate(data, W, H)
    loss_hist = [compute_loss(data, W, H)]
    kW = {k: v for k, v in kw.items() if k.startswith("penaltiesW")}
    kH = {k: v for k, v in kw.items() if k.startswith("penaltiesH")}
    nonneg = kw.get("nonneg", True)
    mask = kw.get("mask", None)
    loss = kw.get("loss", "square")
    cW, cH = kw.get("constrW", "default"), kw.get("constrH", "default")
    for _ in range(max_itr):
        if not eval_mode:
            pgd_update_motifs(rule, data, W, H, nonneg=nonneg, mask=mask, loss=loss, constrW=cW, **kW)
        loss_hist.append(pgd_update_feature_maps(rule, data, W, H, nonneg=nonneg, mask=mask, loss=loss, constrH=cH, **kH))
        if check_convergence and converged(loss_hist, patience, tol):
            break
    return W, H, np.asarray(loss_hist), (rule.stepW, rule.stepH)


# --------------------------------------------------------------------------
# index-level brute force (tiny sizes only; third, independent statement)
# --------------------------------------------------------------------------
def brute_conv(W, H):
    K, N, L = W.shape
    T = H.shape[1]
    est = np.zeros((N, T))
    for n in range(N):
        for t in range(T):
            s = 0.0
            for l in range(L):
                if t - l >= 0:
                    for k in range(K):
                        s += W[k, n, l] * H[k, t - l]
            est[n, t] = s
    return est


def brute_transconv(W, X):
    K, N, L = W.shape
    T = X.shape[1]
    out = np.zeros((K, T))
    for k in range(K):
        for t in range(T):
            s = 0.0
            for l in range(L):
                if t + l < T:
                    for n in range(N):
                        s += W[k, n, l] * X[n, t + l]
            out[k, t] = s
    return out


# --------------------------------------------------------------------------
# ctypes binding of the C restatement (oracle/libcmf_oracle.so, built by
# oracle/Makefile).  Arrays go in Julia memory order (Fortran order here).
# --------------------------------------------------------------------------
_HERE = os.path.dirname(os.path.abspath(__file__))
_C = None


def _f(a):
    return np.asfortranarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def c_lib():
    """Load oracle/libcmf_oracle.so (raises if it has not been built)."""
    global _C
    if _C is None:
        lib = ctypes.CDLL(os.path.join(_HERE, "libcmf_oracle.so"))
        i64, u64, dbl, pd = ctypes.c_int64, ctypes.c_uint64, ctypes.c_double, ctypes.POINTER(ctypes.c_double)
        lib.oracle_tensor_conv.argtypes = [i64] * 4 + [pd] * 3
        lib.oracle_tensor_conv.restype = None
        lib.oracle_tensor_transconv.argtypes = [i64] * 4 + [pd] * 3
        lib.oracle_tensor_transconv.restype = None
        lib.oracle_hxt.argtypes = [i64] * 4 + [pd] * 3
        lib.oracle_hxt.restype = None
        lib.oracle_compute_loss.argtypes = [i64] * 4 + [pd] * 4
        lib.oracle_compute_loss.restype = dbl
        lib.oracle_converged.argtypes = [pd, i64, i64, dbl]
        lib.oracle_converged.restype = ctypes.c_int
        lib.oracle_fit_mult.argtypes = ([i64] * 4 + [pd] * 3 + [i64, dbl, ctypes.c_int, i64, dbl, ctypes.c_int]
                                        + [dbl] * 4 + [pd, pd, ctypes.POINTER(i64)])
        lib.oracle_fit_mult.restype = ctypes.c_int
        lib.oracle_fit_hals.argtypes = ([i64] * 4 + [pd] * 3 + [i64, dbl, ctypes.c_int, i64, dbl, ctypes.c_int]
                                        + [dbl] * 4 + [pd, pd, ctypes.POINTER(i64)])
        lib.oracle_fit_hals.restype = ctypes.c_int
        lib.oracle_init_rand.argtypes = [i64] * 4 + [u64] + [pd] * 3
        lib.oracle_init_rand.restype = None
        lib.oracle_gen_synthetic.argtypes = [i64] * 4 + [dbl] * 4 + [u64] + [pd] * 3
        lib.oracle_gen_synthetic.restype = None
        lib.oracle_rng_u01.argtypes = [u64] * 3
        lib.oracle_rng_u01.restype = dbl
        lib.oracle_rng_normal.argtypes = [u64] * 3
        lib.oracle_rng_normal.restype = dbl
        _C = lib
    return _C


def c_tensor_conv(W, H):
    K, N, L = W.shape
    T = H.shape[1]
    Wf, Hf = _f(W), _f(H)
    est = np.zeros((N, T), order="F")
    c_lib().oracle_tensor_conv(N, T, K, L, _p(Wf), _p(Hf), _p(est))
    return est


def c_tensor_transconv(W, X):
    K, N, L = W.shape
    T = X.shape[1]
    Wf, Xf = _f(W), _f(X)
    out = np.zeros((K, T), order="F")
    c_lib().oracle_tensor_transconv(N, T, K, L, _p(Wf), _p(Xf), _p(out))
    return out


def c_hxt(H, X, L):
    K, T = H.shape
    N = X.shape[0]
    Hf, Xf = _f(H), _f(X)
    out = np.zeros((K, N, L), order="F")
    c_lib().oracle_hxt(N, T, K, L, _p(Hf), _p(Xf), _p(out))
    return out


def c_fit_mult(data, W_init, H_init, max_itr=100, max_time=np.inf, check_convergence=True,
               patience=3, tol=1e-4, eval_mode=False, l1W=0.0, l2W=0.0, l1H=0.0, l2H=0.0):
    K, N, L = W_init.shape
    T = H_init.shape[1]
    d = _f(data)
    W = np.array(W_init, dtype=np.float64, order="F", copy=True)
    H = np.array(H_init, dtype=np.float64, order="F", copy=True)
    lh = np.zeros(max_itr + 1)
    th = np.zeros(max_itr + 1)
    n = ctypes.c_int64(0)
    c_lib().oracle_fit_mult(N, T, K, L, _p(d), _p(W), _p(H), max_itr, float(max_time),
                            int(check_convergence), patience, tol, int(eval_mode),
                            l1W, l2W, l1H, l2H, _p(lh), _p(th), ctypes.byref(n))
    return W, H, lh[: n.value].copy(), th[: n.value].copy()


def c_fit_hals(data, W_init, H_init, max_itr=100, max_time=np.inf, check_convergence=True,
               patience=3, tol=1e-4, eval_mode=False, l1W=0.0, l2W=0.0, l1H=0.0, l2H=0.0):
    K, N, L = W_init.shape
    d = _f(data)
    W = np.array(W_init, dtype=np.float64, order="F", copy=True)
    H = np.array(H_init, dtype=np.float64, order="F", copy=True)
    T = H.shape[1]
    lh = np.zeros(max_itr + 1)
    th = np.zeros(max_itr + 1)
    n = ctypes.c_int64(0)
    c_lib().oracle_fit_hals(N, T, K, L, _p(d), _p(W), _p(H), max_itr, float(max_time),
                            int(check_convergence), patience, tol, int(eval_mode),
                            l1W, l2W, l1H, l2H, _p(lh), _p(th), ctypes.byref(n))
    return W, H, lh[: n.value].copy(), th[: n.value].copy()


_CONSTR_CODE = {None: 0, "nonneg": 1, "unitnorm": 2}


def c_fit_pgd(data, W_init, H_init, max_itr=100, penW_sq=1.0, penW_abs=0.0, penH_sq=0.0, penH_abs=0.0,
              constrW="nonneg", constrH="nonneg", loss="square", mask=None):
    """fit(::AlternatingOptimizer{PGDUpdate}) through the C restatement (oracle_fit_pgd): exactly max_itr iterations.
    Returns W, H, loss_hist, (stepW, stepH)."""
    K, N, L = W_init.shape
    d = _f(data)
    W = np.array(W_init, dtype=np.float64, order="F", copy=True)
    H = np.array(H_init, dtype=np.float64, order="F", copy=True)
    T = H.shape[1]
    lh = np.zeros(max_itr + 1)
    steps = np.zeros(2)
    m = None if mask is None else _f(mask)
    lib = c_lib()
    lib.oracle_fit_pgd.restype = None
    lib.oracle_fit_pgd(ctypes.c_int64(N), ctypes.c_int64(T), ctypes.c_int64(K), ctypes.c_int64(L), _p(d), _p(W), _p(H),
                       ctypes.c_int64(max_itr), ctypes.c_double(penW_sq), ctypes.c_double(penW_abs), ctypes.c_double(penH_sq),
                       ctypes.c_double(penH_abs), ctypes.c_int(_CONSTR_CODE[constrW]), ctypes.c_int(_CONSTR_CODE[constrH]),
                       ctypes.c_int(1 if loss == "abs" else 0), None if m is None else _p(m), _p(lh), _p(steps))
    return W, H, lh, (steps[0], steps[1])


def c_init_rand(data, L, K, seed):
    N, T = data.shape
    d = _f(data)
    W = np.zeros((K, N, L), order="F")
    H = np.zeros((K, T), order="F")
    c_lib().oracle_init_rand(N, T, K, L, seed, _p(d), _p(W), _p(H))
    return W, H


def c_gen_synthetic(N=100, T=500, K=3, L=20, alpha=0.1, p_h=0.5, sigma=0.2, noise_scale=1.0, seed=1234):
    """Defaults: datasets/synthetic.jl:30-37.  Returns (data, W, H)."""
    data = np.zeros((N, T), order="F")
    W = np.zeros((K, N, L), order="F")
    H = np.zeros((K, T), order="F")
    c_lib().oracle_gen_synthetic(N, T, K, L, alpha, p_h, sigma, noise_scale, seed, _p(data), _p(W), _p(H))
    return data, W, H
