"""Host-side mirror of CMF.jl's interface for the multiplicative-update path.

The reference host language is Julia, which is not available where this is built,
so the host layer above the C ABI is written in Python with the reference's own
names, argument meaning and loop semantics (the Julia ``ccall`` twin of this file
is cmf.jl_amd/julia/CMFHip.jl, shown in INTEGRATION.md).  Citations are relative to
the reference checkout.

    fit_cnmf(data; L, K, alg=:mult, max_itr, max_time, l1_*, l2_*, seed, ...)   src/model.jl:58-85
    CNMF_results(data, W, H, time_hist, loss_hist)                                src/model.jl:11-30
    init_rand(data, L, K)                                                         src/model.jl:113-125
    converged(loss_hist, patience, tol)                                           src/model.jl:91-107
    AlternatingOptimizer / fit                                                    src/algs/alternating.jl:10-71
    MultUpdate: ctor, update_motifs!, update_feature_maps!                        src/algs/mult.jl:1-58
    tensor_conv / tensor_transconv / compute_loss                                 src/common.jl:17-81
    gen_synthetic                                                                 README.md:14, datasets/synthetic.jl:29-61

Arrays use Julia's index order: ``data[n, t]`` (N,T), ``W[k, n, l]`` (K,N,L), ``H[k, t]`` (K,T).
All arithmetic runs on the GPU through libcmf_hip.so; nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes
import math
import os
import time
import warnings

import numpy as np

from . import _lib
from ._lib import CMFError, check, farr, ptr

EPSILON = float(np.finfo(np.float64).eps)  # src/CMF.jl:20


# --------------------------------------------------------------------------------------
# results
# --------------------------------------------------------------------------------------
class CNMF_results:
    """Holds results from a single CNMF fit (src/model.jl:11-17)."""

    def __init__(self, data, W, H, time_hist, loss_hist):
        self.data = data
        self.W = W
        self.H = H
        self.time_hist = time_hist
        self.loss_hist = loss_hist

    # accessors: src/model.jl:21-30
    def num_lags(self):
        return self.W.shape[2]

    def num_units(self):
        return self.W.shape[1]

    def num_components(self):
        return self.W.shape[0]

    def num_iter(self):
        return len(self.loss_hist)


# --------------------------------------------------------------------------------------
# stand-alone primitives (src/common.jl)
# --------------------------------------------------------------------------------------
def _dims(W, H=None, X=None):
    K, N, L = W.shape
    T = H.shape[1] if H is not None else X.shape[1]
    return N, T, K, L


def tensor_conv(W, H, device=None):
    """tensor_conv(W, H) -> est (N x T): src/common.jl:17-34."""
    lib = _lib.load()
    W = farr(W)
    N, T, K, L = _dims(W, H=np.asarray(H))
    H = farr(H, (K, T))
    est = np.zeros((N, T), order="F")
    check(lib.cmf_tensor_conv(_dev(device), N, T, K, L, ptr(W), ptr(H), ptr(est)))
    return est


def tensor_transconv(W, X, device=None):
    """tensor_transconv(W, X) -> (K x T): src/common.jl:62-81."""
    lib = _lib.load()
    W = farr(W)
    N, T, K, L = _dims(W, X=np.asarray(X))
    X = farr(X, (N, T))
    out = np.zeros((K, T), order="F")
    check(lib.cmf_tensor_transconv(_dev(device), N, T, K, L, ptr(W), ptr(X), ptr(out)))
    return out


def compute_loss(data, W, H, device=None):
    """compute_loss(data, W, H): src/common.jl:54-59."""
    rule = MultUpdate(data, W, H, device=device)
    try:
        return rule.compute_loss()
    finally:
        rule.close()


def converged(loss_hist, patience, tol):
    """Check for model convergence: src/model.jl:91-107."""
    lib = _lib.load()
    lh = np.ascontiguousarray(loss_hist, dtype=np.float64)
    return bool(lib.cmf_converged(ptr(lh), len(lh), int(patience), float(tol)))


def _dev(device):
    return _lib.default_device() if device is None else int(device)


def rccl_version():
    """{"version": code, "lib": path} of the RCCL this process binds (cmf_rccl_version); raises CMFError without one."""
    lib = _lib.load()
    v = ctypes.c_int()
    buf = ctypes.create_string_buffer(1024)
    check(lib.cmf_rccl_version(ctypes.byref(v), buf, 1024))
    return {"version": v.value, "lib": buf.value.decode()}


# --------------------------------------------------------------------------------------
# update rules (the plugin boundary: abstract type AbstractCFUpdate, alternating.jl:1-8)
# --------------------------------------------------------------------------------------
class AbstractCFUpdate:
    """An update rule that updates both W and H (src/algs/alternating.jl:1-8).

    Must implement ``Rule(data, W, H)``, ``update_motifs(data, W, H, **kwargs)`` and
    ``update_feature_maps(data, W, H, **kwargs) -> loss``.
    """


class MultUpdate(AbstractCFUpdate):
    """MultUpdate on MI355X: drop-in for src/algs/mult.jl behind the rule interface.

    ``MultUpdate(data, W, H)`` mirrors the reference constructor (mult.jl:11-20): it
    uploads ``data`` and the factors and owns the rule's scratch (est, numW, denomW,
    numH, denomH) on the device.  The working copies of W and H stay device-resident
    between calls; ``update_motifs`` / ``update_feature_maps`` advance them, and
    :meth:`download` writes them back into the caller's arrays (``fit`` does this
    once at the end, which is observably the same as the reference's in-place
    mutation for every caller that only reads W and H after ``fit`` returns).  If the
    caller changes W or H on the host between calls, it must call :meth:`upload`.

    ``devices=[d0, d1, ...]`` builds the rule as a T-sharded group on several GPUs of this node, driven by this one
    process (cmf_create_multi): the rule methods keep their meaning and the library runs the sharded iteration --
    ONE RCCL all-reduce of [numW | denomW | loss tail | H halos] per iteration (the Gram form and PGD keep
    an H-halo all-gather of their own).  Listing
    one device several times puts that many shards on it (loopback transport; tests).
    """

    def __init__(self, data, W, H, device=None, devices=None, transport=_lib.CMF_COMM_AUTO, sync_every_call=None, verify_args=None,
                 strict_inplace=None):
        lib = _lib.load()
        self._lib = lib
        for name, val in (("sync_every_call", sync_every_call), ("verify_args", verify_args), ("strict_inplace", strict_inplace)):
            if val is not None:  # (CMFHip.jl's constructor keywords; the class attributes below are the defaults)
                setattr(self, name, val)
        if self.verify_args not in ("sample", "full", "none"):
            raise ValueError('verify_args must be "sample", "full" or "none"')
        self._h = ctypes.c_void_p()
        data = farr(data)
        if data.ndim != 2:
            raise ValueError("data must be a matrix (N x T)")
        W = farr(W)
        if W.ndim != 3:
            raise ValueError("W must be a K x N x L tensor")
        K, N, L = W.shape
        if data.shape[0] != N:
            raise ValueError(f"DimensionMismatch: data has {data.shape[0]} rows, W has N={N}")
        T = data.shape[1]
        H = farr(H, (K, T))
        self.N, self.T, self.K, self.L = N, T, K, L
        if devices is not None:
            devs = [int(x) for x in devices]
            if not devs:
                raise ValueError("devices must not be empty")
            self.device, self.devices = devs[0], devs
            arr = (ctypes.c_int * len(devs))(*devs)
            check(lib.cmf_create_multi(ctypes.byref(self._h), len(devs), arr, int(transport), N, T, K, L, ptr(data)))
        else:
            self.device, self.devices = _dev(device), None
            check(lib.cmf_create(ctypes.byref(self._h), self.device, N, T, K, L, ptr(data)))
        try:
            check(lib.cmf_set_factors(self._h, ptr(W), ptr(H)))
        except Exception:
            self.close()
            raise
        ss = ctypes.c_double()
        check(lib.cmf_get_data_sumsq(self._h, ctypes.byref(ss)))
        self.data_norm = math.sqrt(ss.value)  # mult.jl:13
        # what the rule has read from the caller's arrays (the rule calls compare their arguments with it under sync_every_call)
        self._seen = {"W": self._fingerprint(W), "H": self._fingerprint(H)}

    # ``rule.sync_every_call = True`` (CMFHip.jl's default): every update_feature_maps call also writes the new factors into
    # the W and H it is handed -- the reference's in-place semantics (mult.jl:37-38,51-52) for a caller that looks at its
    # arrays between calls.  The download rides underneath the call's own kernels (cmf_arm_writeback).
    sync_every_call = False
    # The reference's rules READ their W and H arguments (mult.jl:23,42); here the working copies are device-resident.  Under
    # sync_every_call the two are kept equivalent: every rule call fingerprints the arrays it is handed (cmf_fingerprint) and
    # compares with what the rule last read from / wrote into the caller's arrays -- arrays with other contents (other arrays, or
    # the same ones edited by the caller) are uploaded first (``reuploads`` counts).  "sample": one 64-byte line per 4 KB (bulk edits -- rescaling, a new
    # initialisation, zeroed rows -- are seen, a single poked element may not be); "full": every element (+ ~1 ms per call at
    # config 2); "none": the round-5 behaviour (the caller calls upload()).
    verify_args = "sample"
    # W reaches the caller's array when update_feature_maps returns (the write-back), not when update_motifs does: in between
    # the caller's W is the one update_motifs started from.  A caller that edits W there gets a clear error -- unless
    # strict_inplace is set: update_motifs then also downloads W (synchronously: + ~0.5 ms at config 2), so the caller sees the
    # reference's state at every point and its edits are honoured like the reference's (tests/test_dropin_contract.py).
    strict_inplace = False
    reuploads = 0
    _seen = None       # {"W": fingerprint, "H": fingerprint} of the caller's arrays as this rule last read or wrote them
    _w_pending = False  # update_motifs has run and W has not been written back yet

    def _fingerprint(self, a):
        fp = ctypes.c_uint64()
        check(self._lib.cmf_fingerprint(ptr(a), a.size, 1 if self.verify_args == "full" else 64, ctypes.byref(fp)))
        return (self.verify_args, fp.value)  # (contents only, tagged with the form that was taken -- a rule switched to another form uploads once: `fit` deep-copies the initial factors, alternating.jl:33-34 -- equal arrays at another address are the same factors)

    def _check_arrays(self, W, H):
        for a, shape, nm in ((W, (self.K, self.N, self.L), "W"), (H, (self.K, self.T), "H")):
            if a is not None and not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.shape == shape
                                      and a.flags.f_contiguous and a.flags.writeable):
                raise ValueError(f"sync_every_call: {nm} must be a writeable Float64 array of shape {shape} in Julia's (column-major) order")

    def _sync_args(self, W, H):
        """The rule reads its arguments (mult.jl:23,42): arrays this rule has not seen in this state are uploaded first."""
        if not self.sync_every_call or self.verify_args == "none" or W is None or H is None:
            return
        self._check_arrays(W, H)
        now = {"W": self._fingerprint(W), "H": self._fingerprint(H)}
        if self._seen is None:  # (after upload(): the arrays a rule call is handed are taken as they are)
            self._seen = {"W": None, "H": None}
        dW, dH = now["W"] != self._seen["W"], now["H"] != self._seen["H"]
        if dW and self._w_pending:
            raise RuntimeError(
                "W was modified (or another array was passed) between update_motifs and update_feature_maps: with sync_every_call the "
                "caller's W holds the motifs update_motifs started from until update_feature_maps returns, so the edit was made to an "
                "outdated W.  Set rule.strict_inplace = True (update_motifs then writes W back and edits are honoured like the "
                "reference's, mult.jl:42), or call rule.upload(W, H) with the factors you mean (INTEGRATION.md section 3).")
        if dW or dH:
            check(self._lib.cmf_set_factors(self._h, ptr(W) if dW else None, ptr(H) if dH else None))
            self.reuploads += 1
        self._seen = now

    def _after_motifs(self, W):
        if not self.sync_every_call or self.verify_args == "none" or W is None:
            return
        if self.strict_inplace:
            self._check_arrays(W, None)
            check(self._lib.cmf_get_factors(self._h, ptr(W), None))
            if self._seen is not None:
                self._seen["W"] = self._fingerprint(W)
        else:
            self._w_pending = True

    def _after_feature_maps(self, W, H):
        if not self.sync_every_call or self.verify_args == "none" or W is None or H is None:
            return
        self._seen = {"W": self._fingerprint(W), "H": self._fingerprint(H)}  # (the write-back has just filled them)
        self._w_pending = False

    def _arm_writeback(self, W, H):
        if not self.sync_every_call or (W is None and H is None):
            return
        self._check_arrays(W, H)
        check(self._lib.cmf_arm_writeback(self._h, None if W is None else ptr(W), None if H is None else ptr(H)))

    # -- the two rule methods -----------------------------------------------------------
    def update_motifs(self, data=None, W=None, H=None, l1W=0, l2W=0, **kwargs):
        """update_motifs!(rule, data, W, H; l1W=0, l2W=0): src/algs/mult.jl:23-39."""
        self._sync_args(W, H)
        check(self._lib.cmf_update_motifs(self._h, float(l1W), float(l2W)))
        self._after_motifs(W)

    def update_feature_maps(self, data=None, W=None, H=None, l1H=0, l2H=0, **kwargs):
        """update_feature_maps!(rule, data, W, H; l1H=0, l2H=0) -> loss: src/algs/mult.jl:42-58."""
        loss = ctypes.c_double()
        self._sync_args(W, H)
        self._arm_writeback(W, H)
        check(self._lib.cmf_update_feature_maps(self._h, float(l1H), float(l2H), ctypes.byref(loss)))
        self._after_feature_maps(W, H)
        return loss.value

    # -- helpers ------------------------------------------------------------------------
    def compute_loss(self):
        """compute_loss(data, W, H) on the resident factors: src/common.jl:54-59."""
        loss = ctypes.c_double()
        check(self._lib.cmf_compute_loss(self._h, ctypes.byref(loss)))
        return loss.value

    def iterate(self, n, eval_mode=False, l1W=0, l2W=0, l1H=0, l2H=0, stamps=False):
        """n x (update_motifs!; update_feature_maps!) back to back (alternating.jl:51-54) in one ccall (cmf_iterate):
        the losses of the n iterations, read one iteration late so the device never waits for the host."""
        losses = np.zeros(int(n))
        st = np.zeros(int(n))
        check(self._lib.cmf_iterate(self._h, int(n), int(bool(eval_mode)), float(l1W), float(l2W), float(l1H), float(l2H),
                                    ptr(losses), ptr(st)))
        return (losses, st) if stamps else losses

    def synchronize(self):
        """Wait for everything the rule has enqueued (every stream of every local shard of a group)."""
        check(self._lib.cmf_synchronize(self._h))

    def counter(self, name):
        """Event counter of the handle (cmf_get_counter), e.g. "hals_pipeline_reruns"."""
        v = ctypes.c_int64()
        check(self._lib.cmf_get_counter(self._h, name.encode(), ctypes.byref(v)))
        return v.value

    overlap = False
    transport_fallback = None

    def set_overlap(self, flag):
        """Group handles: switch between the two forms of the W phase (library option "allreduce_overlap")."""
        self.set_option("allreduce_overlap", int(bool(flag)))
        self.overlap = bool(flag)

    def comm_info(self):
        buf = ctypes.create_string_buffer(1024)
        check(self._lib.cmf_comm_info(self._h, buf, 1024))
        return buf.value.decode()

    def shard_bounds(self, rank):
        a, b = ctypes.c_int64(), ctypes.c_int64()
        check(self._lib.cmf_shard_bounds(self._h, int(rank), ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def set_option(self, name, value):
        """Library option, e.g. ``set_option("reuse_est", 0)`` to recompute est in update_motifs! like the reference."""
        check(self._lib.cmf_set_option(self._h, name.encode(), int(value)))

    def upload(self, W, H):
        W = farr(W, (self.K, self.N, self.L))
        H = farr(H, (self.K, self.T))
        check(self._lib.cmf_set_factors(self._h, ptr(W), ptr(H)))
        self._seen, self._w_pending = None, False  # (the next rule call takes the arrays it is handed as they are)

    def download(self, W=None, H=None):
        """Write the resident factors into W, H (in place when given) and return them."""
        Wf = np.zeros((self.K, self.N, self.L), order="F")
        Hf = np.zeros((self.K, self.T), order="F")
        check(self._lib.cmf_get_factors(self._h, ptr(Wf), ptr(Hf)))
        if W is not None:
            W[...] = Wf
            Wf = W
        if H is not None:
            H[...] = Hf
            Hf = H
        return Wf, Hf

    def fit_native(self, max_itr, max_time, check_convergence, patience, tol, eval_mode,
                   l1W=0.0, l2W=0.0, l1H=0.0, l2H=0.0):
        """The whole alternating.jl:16-71 loop inside the library (one ccall)."""
        lh = np.zeros(int(max_itr) + 1)
        th = np.zeros(int(max_itr) + 1)
        n = ctypes.c_int64(0)
        early = ctypes.c_int(0)
        check(self._lib.cmf_fit(self._h, int(max_itr), float(max_time), int(bool(check_convergence)), int(patience),
                                float(tol), int(bool(eval_mode)), float(l1W), float(l2W), float(l1H), float(l2H),
                                ptr(lh), ptr(th), ctypes.byref(n), ctypes.byref(early)))
        return lh[: n.value].copy(), th[: n.value].copy(), bool(early.value)

    def kernel_times(self, name):
        """(mean ms, launches) recorded for one kernel class since set_option("profile", 1)."""
        ms, n = ctypes.c_double(), ctypes.c_int64()
        check(self._lib.cmf_kernel_times(self._h, name.encode(), ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def time_kernel(self, name, reps=5):
        """(avg ms, algorithmic flops per launch) of one hot kernel, timed with HIP events."""
        ms, fl = ctypes.c_double(), ctypes.c_double()
        check(self._lib.cmf_time_kernel(self._h, name.encode(), int(reps), ctypes.byref(ms), ctypes.byref(fl)))
        return ms.value, fl.value

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.cmf_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


HIPMultUpdate = MultUpdate


class HALSUpdate(MultUpdate):
    """HALSUpdate on MI355X: drop-in for src/algs/hals.jl behind the same rule interface.

    ``HALSUpdate(data, W, H)`` mirrors hals.jl:18-28.  The residual the reference carries in the
    rule (``resids = tensor_conv(W, H) - data``) is kept implicitly (est on the device); the K*L column
    updates of W (hals.jl:90-112) and the K*T entry updates of H (hals.jl:121-154) run in the
    reference's Gauss-Seidel order on Gram-projected state (see cmf_kernels.h).  Clamp at 0,
    ``+ l2`` regularisation and the incremental-residual loss are the reference's.
    """

    def __init__(self, data, W, H, device=None, **kw):
        super().__init__(data, W, H, device=device, **kw)
        try:  # the rule constructor (hals.jl:18-28): scratch + the shape limits of the on-chip sweeps, reported here
            self.set_option("hals_prepare", 1)
        except Exception:
            self.close()
            raise

    def update_motifs(self, data=None, W=None, H=None, l1W=0, l2W=0, **kwargs):
        """update_motifs!(rule::HALSUpdate, data, W, H; l1W=0, l2W=0): src/algs/hals.jl:31-34."""
        self._sync_args(W, H)
        check(self._lib.cmf_hals_update_motifs(self._h, float(l1W), float(l2W)))
        self._after_motifs(W)

    def update_feature_maps(self, data=None, W=None, H=None, l1H=0, l2H=0, **kwargs):
        """update_feature_maps!(rule::HALSUpdate, data, W, H; l1H=0, l2H=0) -> loss: src/algs/hals.jl:37-42."""
        loss = ctypes.c_double()
        self._sync_args(W, H)
        self._arm_writeback(W, H)
        check(self._lib.cmf_hals_update_feature_maps(self._h, float(l1H), float(l2H), ctypes.byref(loss)))
        self._after_feature_maps(W, H)
        return loss.value

    def fit_native(self, *a, **kw):
        raise NotImplementedError("cmf_fit runs the multiplicative-update rule; drive HALSUpdate with fit()")

    def iterate(self, *a, **kw):
        raise NotImplementedError("cmf_iterate runs the multiplicative-update rule; call update_motifs / update_feature_maps")


HIPHALSUpdate = HALSUpdate


# pgd.jl's loss / penalty / constraint types, as far as the GPU rule supports them
class SquareLoss:
    """D(b, est) = ||b - est||^2 (src/algs/pgd.jl:29-36)."""


class AbsoluteLoss:
    """D(b, est) = ||b - est||_1 with gradient sign(est - b) (src/algs/pgd.jl:41-47)."""


class MaskedLoss:
    """MaskedLoss(loss, mask): gradient and loss of `loss` restricted by an N x T mask (src/algs/pgd.jl:58-70;
    the loss_func of the reference's test/test.jl:45).  `loss` is SquareLoss() or AbsoluteLoss()."""

    def __init__(self, loss, mask):
        if not isinstance(loss, (SquareLoss, AbsoluteLoss)):
            raise NotImplementedError("MaskedLoss on the GPU wraps SquareLoss or AbsoluteLoss")
        self.loss = loss
        self.mask = farr(mask)


class SquarePenalty:
    """R(x) = weight * ||x||_2^2 (src/algs/pgd.jl:73-80)."""

    def __init__(self, weight):
        self.weight = float(weight)


class AbsolutePenalty:
    """R(x) = weight * ||x||_1 (src/algs/pgd.jl:83-89)."""

    def __init__(self, weight):
        self.weight = float(weight)


class NonnegConstraint:
    """x_i >= 0, projected as max(eps(), x) (src/algs/pgd.jl:92-96)."""


class UnitNormConstraint:
    """Every slice along the first dimension (a component k) with norm > 1 is scaled to norm 1 (src/algs/pgd.jl:100-110;
    `constrW=CMF.UnitNormConstraint()` in figures/thesis/exp_reconstruct_synth.jl:69)."""


def _penalty_weights(penalties):
    sq = sum(p.weight for p in penalties if isinstance(p, SquarePenalty))
    ab = sum(p.weight for p in penalties if isinstance(p, AbsolutePenalty))
    if any(not isinstance(p, (SquarePenalty, AbsolutePenalty)) for p in penalties):
        raise NotImplementedError("PGDUpdate on the GPU supports SquarePenalty and AbsolutePenalty")
    return sq, ab


def _nonneg_flag(constr):
    if constr is None:
        return 0
    if isinstance(constr, NonnegConstraint) or constr is NonnegConstraint:
        return 1
    if isinstance(constr, UnitNormConstraint) or constr is UnitNormConstraint:
        return 2
    raise NotImplementedError("PGDUpdate on the GPU supports NonnegConstraint, UnitNormConstraint or no constraint")


class PGDUpdate(MultUpdate):
    """PGDUpdate on MI355X: drop-in for src/algs/pgd.jl:112-202.

    The gradients are the same contractions as the MU numerators (compute_gradW! is the H_shift * X'
    product of mult.jl:31-34, compute_gradH! is tensor_transconv!), taken on the stored residual.
    The rule state (stepW, stepH, cur_loss) lives in the library handle.  ``devices=[...]`` shards T over several
    GPUs like MultUpdate does (one all-reduce of the partial gradW per iteration; the long recordings of
    notebooks/test_mouse.ipynb are fitted with this rule)."""

    def __init__(self, data, W, H, device=None, devices=None, transport=_lib.CMF_COMM_AUTO, **kw):
        super().__init__(data, W, H, device=device, devices=devices, transport=transport, **kw)
        check(self._lib.cmf_pgd_reset(self._h))
        self._mask_key = None

    def _select_loss(self, loss_func):
        """loss_func=SquareLoss() (default), AbsoluteLoss() or MaskedLoss(either, mask): uploads the mask when it changes."""
        base = loss_func.loss if isinstance(loss_func, MaskedLoss) else loss_func
        if base is None or isinstance(base, SquareLoss) or base is SquareLoss:
            kind = 0
        elif isinstance(base, AbsoluteLoss) or base is AbsoluteLoss:
            kind = 1
        else:
            raise NotImplementedError("PGDUpdate on the GPU supports SquareLoss, AbsoluteLoss and MaskedLoss of either")
        if kind != getattr(self, "_loss_kind", 0):
            check(self._lib.cmf_pgd_set_loss(self._h, kind))
            self._loss_kind = kind
        key = id(loss_func) if isinstance(loss_func, MaskedLoss) else None
        if key == self._mask_key:
            return
        if key is None:
            self._upload_mask(None)
        else:
            if loss_func.mask.shape != (self.N, self.T):
                raise ValueError(f"mask must be {self.N} x {self.T} like data, got {loss_func.mask.shape}")
            self._upload_mask(loss_func.mask)
        self._mask_key = key

    def _upload_mask(self, mask):
        check(self._lib.cmf_set_mask(self._h, None if mask is None else ptr(mask)))

    def update_motifs(self, data=None, W=None, H=None, loss_func=None, constrW=NonnegConstraint, penaltiesW=None, **kwargs):
        """update_motifs!(rule::PGDUpdate, ...; loss_func=SquareLoss(), constrW=NonnegConstraint(),
        penaltiesW=[SquarePenalty(1)]): src/algs/pgd.jl:158-177."""
        self._select_loss(loss_func)
        sq, ab = _penalty_weights([SquarePenalty(1)] if penaltiesW is None else penaltiesW)
        self._sync_args(W, H)
        check(self._lib.cmf_pgd_update_motifs(self._h, sq, ab, _nonneg_flag(constrW)))
        self._after_motifs(W)

    def update_feature_maps(self, data=None, W=None, H=None, loss_func=None, constrH=NonnegConstraint, penaltiesH=None, **kwargs):
        """update_feature_maps!(rule::PGDUpdate, ...; constrH=NonnegConstraint(), penaltiesH=[]) -> loss:
        src/algs/pgd.jl:180-202."""
        self._select_loss(loss_func)
        sq, ab = _penalty_weights([] if penaltiesH is None else penaltiesH)
        loss = ctypes.c_double()
        self._sync_args(W, H)
        self._arm_writeback(W, H)
        check(self._lib.cmf_pgd_update_feature_maps(self._h, sq, ab, _nonneg_flag(constrH), ctypes.byref(loss)))
        self._after_feature_maps(W, H)
        return loss.value

    @property
    def steps(self):
        a, b = ctypes.c_double(), ctypes.c_double()
        check(self._lib.cmf_pgd_get_steps(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def fit_native(self, *a, **kw):
        raise NotImplementedError("cmf_fit runs the multiplicative-update rule; drive PGDUpdate with fit()")

    def iterate(self, *a, **kw):
        raise NotImplementedError("cmf_iterate runs the multiplicative-update rule; call update_motifs / update_feature_maps")


HIPPGDUpdate = PGDUpdate


def _resolve_alg(alg):
    """alg may be a rule type (HEAD, model.jl:60) or a README-style symbol (README.md:30-33)."""
    if isinstance(alg, str):
        name = alg.lstrip(":").lower()
        if name in ("mult", "mu"):
            return MultUpdate
        if name == "hals":
            return HALSUpdate
        if name == "pgd":
            return PGDUpdate
        if name in ("anls", "admm", "sep"):
            raise NotImplementedError(f"alg=:{name} is outside the MI355X hot path built here (:mult, :hals, :pgd)")
        raise ValueError(f"unknown algorithm {alg!r}")
    if isinstance(alg, type) and issubclass(alg, AbstractCFUpdate):
        return alg
    raise TypeError(f"alg must be an update-rule type or a name like ':mult', got {alg!r}")


# --------------------------------------------------------------------------------------
# driver (src/algs/alternating.jl)
# --------------------------------------------------------------------------------------
class AlternatingOptimizer:
    """AlternatingOptimizer(update_rule, max_itr, max_time): src/algs/alternating.jl:10-14."""

    def __init__(self, update_rule, max_itr, max_time):
        self.update_rule = update_rule
        self.max_itr = max_itr
        self.max_time = max_time


def fit(alg, data, L, K, W_init, H_init, verbose=False, **kwargs):
    """fit(alg::AlternatingOptimizer, data, L, K, W_init, H_init; kwargs...): alternating.jl:16-71."""
    # Load keyword args (:23-31)
    check_convergence = kwargs.get("check_convergence", True)
    patience = kwargs.get("patience", 3)
    eval_mode = kwargs.get("eval_mode", False)
    assert patience >= 1
    tol = kwargs.get("tol", 1e-4)

    W = np.array(W_init, dtype=np.float64, order="F", copy=True)  # :33-34 deepcopy
    H = np.array(H_init, dtype=np.float64, order="F", copy=True)
    rule = alg.update_rule
    device_resident = hasattr(rule, "download")

    # Set up optimization tracking (:37-38)
    loss_hist = [rule.compute_loss() if device_resident else compute_loss(data, W, H)]
    time_hist = [0.0]

    if verbose:
        print("Starting ", end="", flush=True)

    itr = 1
    while itr <= alg.max_itr and time_hist[-1] <= alg.max_time:  # :45
        itr += 1
        t0 = time.time()
        if not eval_mode:  # Skip motif update in evaluation mode (:51-53)
            rule.update_motifs(data, W, H, **kwargs)
        loss = rule.update_feature_maps(data, W, H, **kwargs)  # :54 (synchronises)
        dur = time.time() - t0
        if hasattr(rule, "agree_scalar"):
            dur = rule.agree_scalar(dur)  # sharded rule: all ranks follow rank 0's clock, so they stop together
        time_hist.append(time_hist[-1] + dur)  # :57-59
        loss_hist.append(loss)
        if verbose:
            print(".", end="", flush=True)
        if check_convergence and converged(loss_hist, patience, tol):  # :63-66
            print("Converged early.")
            break
    if verbose:
        print(" fit!")

    if device_resident:
        rule.download(W, H)
    return CNMF_results(data, W, H, np.asarray(time_hist), np.asarray(loss_hist))  # :70


# --------------------------------------------------------------------------------------
# public API (src/model.jl)
# --------------------------------------------------------------------------------------
_REG_ALIASES = {"l1_W": "l1W", "l2_W": "l2W", "l1_H": "l1H", "l2_H": "l2H"}  # README.md:44-52 -> mult.jl:23,42
_KNOWN_KW = {"seed", "W_init", "H_init", "check_convergence", "patience", "eval_mode", "tol", "verbose",
             "l1W", "l2W", "l1H", "l2H", "device", "devices", "options",
             "loss_func", "constrW", "constrH", "penaltiesW", "penaltiesH"}  # PGDUpdate (pgd.jl:158-202)


def init_rand(data, L, K, seed=None, device=None):
    """Initialize randomly, scaling to minimize square error: src/model.jl:113-125.

    Uses the library's portable counter RNG (Julia's MersenneTwister streams are not
    reproducible across Julia versions); ``seed=None`` draws a fresh seed.
    """
    lib = _lib.load()
    data = farr(data)
    N, T = data.shape
    if seed is None:
        seed = int.from_bytes(os.urandom(8), "little")
    W = np.zeros((K, N, L), order="F")
    H = np.zeros((K, T), order="F")
    check(lib.cmf_init_rand(_dev(device), N, T, K, L, int(seed) & (2**64 - 1), ptr(data), ptr(W), ptr(H)))
    return W, H


def fit_cnmf(data, L=10, K=5, alg=MultUpdate, max_itr=100, max_time=math.inf, **kwargs):
    """fit_cnmf(data; L=10, K=5, alg=MultUpdate, max_itr=100, max_time=Inf, kwargs...): src/model.jl:58-85.

    Accepts both HEAD's rule types and the README's symbols (``alg=":mult"``), and both
    spellings of the regularisers (``l1_W`` of README.md:44-52 and ``l1W`` of mult.jl:23,42),
    which HEAD silently drops.
    """
    kw = {}
    for k, v in kwargs.items():
        k2 = _REG_ALIASES.get(k, k)
        if k2 in kw:
            raise TypeError(f"regulariser given twice: {k} and {k2}")
        kw[k2] = v
    unknown = set(kw) - _KNOWN_KW
    if unknown:
        warnings.warn(f"fit_cnmf: ignoring unknown keyword arguments {sorted(unknown)} "
                      "(the reference ignores them silently)", stacklevel=2)
    device = kw.pop("device", None)
    devices = kw.pop("devices", None)  # several GPUs of this node: the T-sharded group form of the :mult rule
    options = kw.pop("options", None)  # {name: value} for cmf_set_option on the rule (include/cmf_hip.h lists them)
    rule_type = _resolve_alg(alg)
    data = farr(data)

    seed = kw.get("seed", None)  # :64-67
    # Initialize (:70) -- always runs, like the reference (it consumes the RNG even when inits are given)
    W_init, H_init = init_rand(data, L, K, seed=seed, device=device)
    W_init = kw.get("W_init", W_init)  # :72-73
    H_init = kw.get("H_init", H_init)

    if devices is not None and rule_type not in (MultUpdate, PGDUpdate):
        raise NotImplementedError("devices=[...] (T sharding) is available for alg=:mult and :pgd; HALS sweeps H sequentially along T")
    if devices is not None:
        rule = rule_type(data, W_init, H_init, devices=devices)
    else:
        rule = (rule_type(data, W_init, H_init, device=device) if issubclass(rule_type, MultUpdate)
                else rule_type(data, W_init, H_init))
    try:
        for name, value in (options or {}).items():
            rule.set_option(name, value)
        opt = AlternatingOptimizer(rule, max_itr, max_time)  # :78-82
        loop_kw = {k: v for k, v in kw.items() if k not in ("seed", "W_init", "H_init")}
        return fit(opt, data, L, K, W_init, H_init, **loop_kw)  # :84
    finally:
        if hasattr(rule, "close"):
            rule.close()


def gen_synthetic(N=100, T=500, K=3, L=20, alpha=0.1, p_h=0.5, sigma=0.2, noise_scale=1.0, seed=1234,
                  return_factors=False, device=None):
    """gen_synthetic(N=, T=) -> data (README.md:14), following synthetic_sequences
    (datasets/synthetic.jl:29-61; same defaults).  ``return_factors=True`` also returns (W, H)."""
    lib = _lib.load()
    data = np.zeros((N, T), order="F")
    W = np.zeros((K, N, L), order="F")
    H = np.zeros((K, T), order="F")
    check(lib.cmf_gen_synthetic(_dev(device), N, T, K, L, float(alpha), float(p_h), float(sigma), float(noise_scale),
                                int(seed) & (2**64 - 1), ptr(data), ptr(W), ptr(H)))
    return (data, W, H) if return_factors else data


# --------------------------------------------------------------------------------------
# callers either side of the path: evaluation, sweeps, results on disk
# --------------------------------------------------------------------------------------
def evaluate_mse(r, device=None):
    """evaluate_mse(r::CNMF_results): src/evaluate.jl:1-5."""
    return compute_loss(r.data, r.W, r.H, device=device)


def evaluate_test(r, test, num_iter=30, device=None):
    """evaluate_test(r, test; num_iter=30): src/evaluate.jl:8-25 -- refit H on held-out data with the
    motifs fixed (HALS H sweeps from H = 0, no regularisation), then the normalised loss.  (The
    reference's version calls a `HALS` module that no longer exists at HEAD; this is its intent.)"""
    test = farr(test)
    K = r.W.shape[0]
    rule = HALSUpdate(test, r.W, np.zeros((K, test.shape[1])), device=device)
    try:
        for _ in range(int(num_iter)):
            rule.update_feature_maps()
        return rule.compute_loss()
    finally:
        rule.close()


def evaluate_convergence(r, thresh=0.01):
    """evaluate_convergence(r; thresh=0.01): src/evaluate.jl:29-44 -- first iteration whose loss is within
    `thresh` (relative) of the final loss."""
    min_loss = r.loss_hist[-1]
    for i, loss in enumerate(r.loss_hist):
        if loss / min_loss < 1 + thresh:
            return i
    return len(r.loss_hist)


def parameter_sweep(data, L_vals=(7,), K_vals=(3,), alg_vals=(":mult",), max_itr=100, max_time=math.inf, group=None, **kwargs):
    """parameter_sweep(data; L_vals, K_vals, alg_vals, max_itr, max_time): src/model.jl:132-145.
    Returns {(L, K, alg): CNMF_results}; other keywords go to every fit_cnmf call (HEAD passes stale
    `lambda1/initW` names that fit_cnmf ignores; here they would be reported as unknown).

    Multi-GPU: the fits are independent, so when a torch.distributed process group is up
    (one process per GPU) each rank runs every world-th combination on its own device and the results are gathered
    on all ranks -- replicas, no data-path collective.  A `seed` keyword is used as given by every fit."""
    combos = [(L, K, alg) for L in L_vals for K in K_vals for alg in alg_vals]
    dist = None
    import sys

    if "torch" in sys.modules:  # a process group can only be up if torch is already imported: never import it from here
        # (importing torch AFTER libcmf_hip.so maps PyTorch's bundled HIP / HSA runtime next to the system one this
        # library is already bound to, and RCCL then picks the uninitialised copy)
        import torch.distributed as _dist

        if _dist.is_available() and _dist.is_initialized():
            dist = _dist
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist else (0, 1)
    mine = {}
    for idx, (L, K, alg) in enumerate(combos):
        if idx % world == rank:
            mine[(L, K, alg)] = fit_cnmf(data, L=L, K=K, alg=alg, max_itr=max_itr, max_time=max_time, **kwargs)
    if world == 1:
        return mine
    # results travel without their copy of `data` (every rank holds it already)
    packed = {k: (r.W, r.H, r.time_hist, r.loss_hist) for k, r in mine.items()}
    parts = [None] * world
    dist.all_gather_object(parts, packed, group=group)
    data_f = np.asarray(data, dtype=np.float64)
    results = {}
    for key in combos:  # the reference's insertion order
        for part in parts:
            if key in part:
                W, H, th, lh = part[key]
                results[key] = CNMF_results(data_f, W, H, th, lh)
    return results


_MODEL_KEYS = ("W", "H", "data", "loss_hist", "time_hist")


_META_KEYS = ("l1_H", "l2_H", "l1_W", "l2_W", "alg")


def _is_hdf5_path(path):
    return str(path).lower().endswith((".h5", ".hdf5", ".hdf"))


def save_model(results, path, **meta):
    """save_model(results, path): src/model.jl:149-163.  The reference's schema: datasets W, H, data, loss_hist,
    time_hist plus whatever of l1_H, l2_H, l1_W, l2_W, alg is passed as keywords (CNMF_results itself does not carry
    them, which is why the reference's own save_model is broken at HEAD).  A path ending in .h5 / .hdf5 is written as
    a real HDF5 file in HDF5.jl's conventions (cmf.jl_amd/_hdf5.py over the system libhdf5), readable by the
    reference's load_model; any other path is a NumPy .npz container with the same names."""
    arrays = {k: np.asarray(getattr(results, k)) for k in _MODEL_KEYS}
    if _is_hdf5_path(path):
        from . import _hdf5

        items = dict(arrays)
        for k, v in meta.items():
            items[k] = str(v).lstrip(":") if k == "alg" or isinstance(v, str) else np.asarray(v, dtype=np.float64)
        _hdf5.write_file(path, items)
        return
    for k, v in meta.items():  # (`alg` without the colon of a Julia symbol, in both containers)
        arrays[k] = np.asarray(str(v).lstrip(":") if k == "alg" else v)
    np.savez_compressed(path, **arrays)


def load_model(path):
    """load_model(path): src/model.jl:167-181 -> (CNMF_results, meta dict); .h5 / .hdf5 files as written by either
    side, otherwise the .npz container."""
    if _is_hdf5_path(path):
        from . import _hdf5

        d = _hdf5.read_file(path, _MODEL_KEYS + _META_KEYS)
        missing = [k for k in _MODEL_KEYS if k not in d]
        if missing:
            raise KeyError(f"{path}: datasets {missing} are missing")
        r = CNMF_results(d["data"], d["W"], d["H"], d["time_hist"], d["loss_hist"])
        return r, {k: d[k] for k in _META_KEYS if k in d}
    with np.load(path, allow_pickle=False) as f:
        r = CNMF_results(f["data"], f["W"], f["H"], f["time_hist"], f["loss_hist"])
        meta = {k: f[k][()] for k in f.files if k not in _MODEL_KEYS}
    return r, meta
