"""TEST INFRASTRUCTURE: numpy statement of the Gram-projected HALS sweeps the GPU kernels implement
(cmf.jl_amd/csrc/cmf_kernels.h, "HALS rule").  tests/test_oracle.py checks it against the literal
residual-form restatement of src/algs/hals.jl in the oracle."""
import numpy as np


def w_sweep(o, W, H, data, l1, l2):
    """hals.jl:90-112 on G = resid * H_unfold' and HH = H_unfold * H_unfold'."""
    K, N, L = W.shape
    R = o.tensor_conv(W, H) - data
    Hu = o.shift_and_stack(H, L)  # row = l*K + k
    G = R @ Hu.T                  # (= denomW - numW of the MU path)
    HH = Hu @ Hu.T
    W = W.copy()
    for k in range(K):
        for l in range(L):
            j = l * K + k
            wo = W[k, :, l].copy()
            v = G[:, j] - wo * HH[j, j]
            wn = np.maximum((-v - l1) / (HH[j, j] + o.EPS + l2), 0.0)
            G += np.outer(wn - wo, HH[j])
            W[k, :, l] = wn
    return W


def h_sweep(o, W, H, data, l1, l2):
    """hals.jl:121-154 on P = transconv(W, resid) and the lag-Gram taps of W."""
    K, N, L = W.shape
    T = H.shape[1]
    R = o.tensor_conv(W, H) - data
    P = o.tensor_transconv(W, R)  # (= denomH - numH of the MU path)
    PW = np.einsum("knl,jnm->lmkj", W, W)  # PW[l, l', k, k'] = <W[k,:,l], W[k',:,l']>

    def taps(k, Lt):  # g[k', e+L-1] = sum_{l < Lt} PW[l, l-e, k, k']
        g = np.zeros((K, 2 * L - 1))
        for e in range(-(L - 1), L):
            for l in range(Lt):
                if 0 <= l - e < L:
                    g[:, e + L - 1] += PW[l, l - e, k, :]
        return g

    H = H.copy()
    for k in range(K):
        full = taps(k, L)
        D = np.zeros(T)
        gs = {}
        for t in range(T):
            Lt = min(L, T - t)
            g = full if Lt == L else gs.setdefault(t, taps(k, Lt))
            nrm = g[k, L - 1]
            ho = H[k, t]
            hn = max((ho * nrm - P[k, t] - l1) / (nrm + o.EPS + l2), 0.0)
            D[t] = hn - ho
            H[k, t] = hn
            for e in range(1, L):           # same row, later columns
                if t + e < T:
                    P[k, t + e] += D[t] * g[k, e + L - 1]
        for t in range(T):                   # later rows
            Lt = min(L, T - t)
            g = full if Lt == L else gs[t]
            for e in range(-(L - 1), L):
                if 0 <= t + e < T:
                    P[k + 1:, t + e] += D[t] * g[k + 1:, e + L - 1]
    return H


def h_sweep_pull_pipeline(o, W, H, data, l1, l2, block=8, seed=0):
    """The same H sweep in the form of the persistent pipeline (hals_h_persist_kernel): rows advance in blocks of
    `block` columns; before row k sweeps block b, the cross-row terms of that block are PULLED from the changes D of all
    rows above (one writer per block of P), which is allowed as soon as row k-1 has finished block b+1 (L-1 <= block);
    which eligible (row, block) runs next is drawn at random: any order the flags permit must give the same H."""
    K, N, L = W.shape
    T = H.shape[1]
    assert L - 1 <= block
    R = o.tensor_conv(W, H) - data
    P = o.tensor_transconv(W, R)
    PW = np.einsum("knl,jnm->lmkj", W, W)

    def taps(k, Lt):
        g = np.zeros((K, 2 * L - 1))
        for e in range(-(L - 1), L):
            for l in range(Lt):
                if 0 <= l - e < L:
                    g[:, e + L - 1] += PW[l, l - e, k, :]
        return g

    tap_cache = {}

    def tap(k, t):  # taps of source column t of row k: truncated window at the right edge (hals.jl:136)
        Lt = min(L, T - t)
        key = (k, Lt)
        if key not in tap_cache:
            tap_cache[key] = taps(k, Lt)
        return tap_cache[key]

    H = H.copy()
    D = np.zeros((K, T))
    nblk = -(-T // block)
    done = [0] * K      # blocks swept per row (the sweeper's progress flag)
    rng = np.random.default_rng(seed)
    while min(done) < nblk:
        ready = [k for k in range(K) if done[k] < nblk and (k == 0 or done[k - 1] >= min(done[k] + 2, nblk))]
        k = int(rng.choice(ready))
        b = done[k]
        t0, t1 = b * block, min((b + 1) * block, T)
        for k2 in range(k):  # the pull: sources of rows above within L-1 columns of the block, in row order
            for tp in range(t0, t1):
                for e in range(-(L - 1), L):
                    t = tp - e
                    if 0 <= t < T:
                        P[k, tp] += D[k2, t] * tap(k2, t)[k, e + L - 1]
        for t in range(t0, t1):  # the sweep of the block (same-row terms pushed ahead as before)
            g = tap(k, t)
            nrm = g[k, L - 1]
            ho = H[k, t]
            hn = max((ho * nrm - P[k, t] - l1) / (nrm + o.EPS + l2), 0.0)
            D[k, t] = hn - ho
            H[k, t] = hn
            for e in range(1, L):
                if t + e < T:
                    P[k, t + e] += D[k, t] * g[k, e + L - 1]
        done[k] += 1
    return H


def hh_from_lag_correlations(o, H, L):
    """HH = H_unfold * H_unfold' (hals.jl:56-60) the way compute_hh / hals_hh_kernel form it: from the lag correlations
    C[d][a][b] = sum_t H[a][t-d] * H[b][t] of H with itself, minus the terms the shift cuts off at the right end."""
    K, T = H.shape
    C = np.zeros((L, K, K))
    for d in range(L):
        C[d] = H[:, : T - d] @ H[:, d:].T if d < T else 0.0
    HH = np.zeros((L * K, L * K))
    for l in range(L):
        for lp in range(L):
            d = abs(l - lp)
            cut = min(l, lp)
            for k in range(K):
                for kp in range(K):
                    a, b = (k, kp) if l >= lp else (kp, k)  # a: the row with the larger lag
                    v = C[d][a][b] if d < T else 0.0
                    for u in range(max(T - cut, d), T):
                        v -= H[a, u - d] * H[b, u]
                    HH[l * K + k, lp * K + kp] = v
    return HH
