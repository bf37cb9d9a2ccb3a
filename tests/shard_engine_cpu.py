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
    def __init__(self, data_local, W, H_local, t_offset, T_global, device, data_left=None):
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
        # "halo in the all-reduce" (cmf_groups.hip, round 6): the shard also updates the L-1 columns in front of its own, from an H
        # that is valid 2(L-1) columns out on the left (Hl2 = columns [-2(L-1), -(L-1))) and the L-1 columns of data in front
        self.Hl2 = np.zeros((K, max(L - 1, 0)))
        self.data_left = None if data_left is None else np.array(data_left, dtype=np.float64)
        self.slot_count = max(3 * (L - 1) * K, 1)
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

    def h_update_front(self, l1H, l2H):
        """h_update_impl(..., front = true): mult.jl:44-52 on columns [-(L-1), Tl) of the shard."""
        K, N, L = self.W.shape
        h = L - 1
        Hext = np.concatenate([self.Hl2, self.Hl, self.H, self.Hr], axis=1)   # column j <-> local t = j - 2h
        ncols = h + self.Tl + self.halo_r                                     # est on t in [-h, Tl + halo_r)
        est = np.zeros((N, ncols))
        for l in range(L):
            est += self.W[:, :, l].T @ Hext[:, h - l: h - l + ncols]            # H[t - l], t = -h + i  ->  j = h - l + i
        X = np.concatenate([self.data_left, self.data_ext], axis=1)            # column i <-> t = -h + i
        nupd = h + self.Tl
        num = np.zeros((K, nupd))
        den = np.zeros((K, nupd))
        for l in range(L):
            w = min(nupd, ncols - l)                                           # t + l must exist (own or right halo)
            if w <= 0:
                continue
            num[:, :w] += self.W[:, :, l] @ X[:, l: l + w]
            den[:, :w] += self.W[:, :, l] @ est[:, l: l + w]
        Hc = np.concatenate([self.Hl, self.H], axis=1)
        Hc *= num / (((den + l1H) + (2.0 * l2H) * Hc) + EPS)
        np.maximum(Hc, EPS, out=Hc)
        self.Hl[...] = Hc[:, :h]
        self.H[...] = Hc[:, h:]

    def halo_pack3(self):
        """halo_pack3_kernel: [last 2(L-1) | first L-1] own columns of H (zeros where the shard is shorter)."""
        L, K = self.L, self.K
        h = L - 1
        out = np.zeros((3 * h, K))
        for c in range(3 * h):
            t = self.Tl - 2 * h + c if c < 2 * h else c - 2 * h
            if 0 <= t < self.Tl:
                out[c] = self.H[:, t]
        return torch.from_numpy(out.ravel().copy())

    def halo_unpack3(self, left_slot, right_slot):
        """halo_unpack3_kernel: H[-2(L-1), 0) from the left neighbour's last columns, H[Tl, Tl + L-1) from the right neighbour's first."""
        L, K = self.L, self.K
        h = L - 1
        if left_slot is not None:
            a = left_slot.numpy().reshape(3 * h, K)
            self.Hl2[...] = a[:h].T
            self.Hl[...] = a[h:2 * h].T
        if right_slot is not None:
            a = right_slot.numpy().reshape(3 * h, K)
            self.Hr[...] = a[2 * h:].T

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
