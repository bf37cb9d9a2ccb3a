"""CPU stand-in for the per-rank shard engine (TEST INFRASTRUCTURE).

An independent fp64 numpy statement of what one rank computes in the T-sharded MU iteration
(SURVEY.md section 8e), with the same interface as cmf_jl_amd.sharded.HipShardEngine, so the
orchestration in ShardedMultUpdate (partition, the single all-reduce of [numW | denomW], the
H halo exchange, the loss reduction) can be checked against the unsharded oracle with the
gloo backend on machines without a GPU."""
import numpy as np
import torch

EPS = float(np.finfo(np.float64).eps)


class OracleShardEngine:
    def __init__(self, data_local, W, H_local, t_offset, T_global, device):
        K, N, L = W.shape
        self.K, self.N, self.L = K, N, L
        self.Tl = H_local.shape[1]
        self.halo_r = data_local.shape[1] - self.Tl
        assert self.halo_r == min(L - 1, T_global - t_offset - self.Tl)
        self.data_ext = np.array(data_local, dtype=np.float64)          # own + right halo columns
        self.data = self.data_ext[:, : self.Tl]
        self.W = np.array(W, dtype=np.float64)
        self.Hl = np.zeros((K, max(L - 1, 0)))                           # left halo (zeros at the global edge)
        self.Hr = np.zeros((K, max(L - 1, 0)))                           # right halo
        self.H = np.array(H_local, dtype=np.float64)
        self.numden = torch.zeros(2 * K * N * L, dtype=torch.float64)
        self.halo_count = max((L - 1) * K, 1)
        self.halo_send = torch.zeros(2 * self.halo_count, dtype=torch.float64)
        self.halo = [self.halo_send[: self.halo_count], self.halo_send[self.halo_count:], None, None]
        self._norm = None

    # conv on columns [0, ncols) of the shard, using the halos
    def _est(self, ncols):
        K, N, L = self.W.shape
        Hext = np.concatenate([self.Hl, self.H, self.Hr], axis=1)       # column j <-> local t = j - (L-1)
        off = L - 1
        est = np.zeros((N, ncols))
        for l in range(L):
            est += self.W[:, :, l].T @ Hext[:, off - l: off - l + ncols]
        return est, Hext

    def data_sumsq(self):
        return float(np.sum(self.data ** 2))

    def set_data_norm(self, x):
        self._norm = x

    def set_factors(self, W, H_local):
        self.W[...] = W
        self.H[...] = H_local

    def get_factors(self):
        return self.W.copy(), self.H.copy()

    def w_partial(self):
        K, N, L = self.W.shape
        est, Hext = self._est(self.Tl)
        off = L - 1
        num = np.zeros((K, N, L))
        den = np.zeros((K, N, L))
        for l in range(L):
            Hs = Hext[:, off - l: off - l + self.Tl]                    # H[t - l] for own t (left halo for t < l)
            num[:, :, l] = Hs @ self.data.T
            den[:, :, l] = Hs @ est.T
        self.numden[:] = torch.from_numpy(np.concatenate([num.ravel(), den.ravel()]))

    def w_partial_num(self):
        K, N, L = self.W.shape
        _, Hext = self._est(self.Tl)
        off = L - 1
        num = np.zeros((K, N, L))
        for l in range(L):
            num[:, :, l] = Hext[:, off - l: off - l + self.Tl] @ self.data.T
        self.numden[: K * N * L] = torch.from_numpy(num.ravel())

    def w_partial_den(self):
        K, N, L = self.W.shape
        est, Hext = self._est(self.Tl)
        off = L - 1
        den = np.zeros((K, N, L))
        for l in range(L):
            den[:, :, l] = Hext[:, off - l: off - l + self.Tl] @ est.T
        self.numden[K * N * L:] = torch.from_numpy(den.ravel())

    def w_apply(self, l1W, l2W):
        K, N, L = self.W.shape
        nd = self.numden.numpy()
        num = nd[: K * N * L].reshape(K, N, L)
        den = nd[K * N * L:].reshape(K, N, L)
        self.W *= num / (((den + l1W) + (2.0 * l2W) * self.W) + EPS)
        np.maximum(self.W, EPS, out=self.W)

    def h_update(self, l1H, l2H):
        K, N, L = self.W.shape
        next_ = self.Tl + self.halo_r
        est_ext, _ = self._est(next_)
        num = np.zeros((K, self.Tl))
        den = np.zeros((K, self.Tl))
        for l in range(L):
            w = min(self.Tl, next_ - l)                                  # t + l must exist (own or right halo)
            if w <= 0:
                continue
            num[:, :w] += self.W[:, :, l] @ self.data_ext[:, l: l + w]
            den[:, :w] += self.W[:, :, l] @ est_ext[:, l: l + w]
        self.H *= num / (((den + l1H) + (2.0 * l2H) * self.H) + EPS)
        np.maximum(self.H, EPS, out=self.H)

    def attach_gathered_halos(self, gathered, rank, world):
        c = self.halo_count
        if rank > 0:
            self.halo[2] = gathered[(2 * (rank - 1) + 1) * c: (2 * (rank - 1) + 2) * c]
        if rank < world - 1:
            self.halo[3] = gathered[(2 * (rank + 1)) * c: (2 * (rank + 1) + 1) * c]

    def halo_pack(self):
        L, K = self.L, self.K
        if L < 2:
            return
        self.halo[0][:] = torch.from_numpy(np.ascontiguousarray(self.H[:, : L - 1].T).ravel())
        self.halo[1][:] = torch.from_numpy(np.ascontiguousarray(self.H[:, -(L - 1):].T).ravel())

    def halo_unpack(self, has_left, has_right):
        L, K = self.L, self.K
        if L < 2:
            return
        if has_left:
            self.Hl[...] = self.halo[2].numpy().reshape(L - 1, K).T
        if has_right:
            self.Hr[...] = self.halo[3].numpy().reshape(L - 1, K).T

    def loss_partial(self):
        est, _ = self._est(self.Tl)
        return float(np.sum((est - self.data) ** 2))

    def close(self):
        pass
