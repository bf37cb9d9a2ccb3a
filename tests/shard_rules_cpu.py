"""CPU mirrors of the group rules beyond the plain MU iteration (TEST INFRASTRUCTURE): the Gram form of the MU iteration
and the PGD rule on T-sharded groups, as csrc/cmf_groups.hip / cmf_api.hip run them, stated in numpy over torch.distributed
(gloo) so that the sharded ALGEBRA -- which sums are local, which are all-reduced, where the halos enter, what only the last
shard does -- is checked against the unsharded oracle on machines without a GPU.

The statements here are deliberately NOT the library's: HH = H_unfold H_unfold' is formed per shard from the definition
(the columns of H_unfold the shard owns), where the library assembles it from lag correlations minus cut terms; denomH is
formed from the pairwise products of W applied to H with its halos, term by term.  Both must give the oracle's fit.
"""
import math

import numpy as np
import torch
import torch.distributed as dist

from shard_engine_cpu import EPS, OracleShardEngine
from shard_protocol_cpu import ProtocolShardedMultUpdate


class GramShardEngine(OracleShardEngine):
    """One rank's share of the Gram-form iteration ("On T-sharded groups")."""

    def unfold_own(self):
        """H_unfold restricted to the columns this shard owns: row (l, k) holds H[k][t - l] for own t (the left halo supplies
        t - l < 0 locally; at the global left edge the halo is zero, which is the reference's truncation)."""
        K, L, off = self.K, self.L, self.L - 1
        Hext = np.concatenate([self.Hl, self.H, self.Hr], axis=1)
        U = np.zeros((L * K, self.Tl))
        for l in range(L):
            U[l * K:(l + 1) * K] = Hext[:, off - l: off - l + self.Tl]
        return U

    def w_partial_gram(self):
        """numW and this shard's additive share of HH (both sums over own t) -> [numW | HH]."""
        K, N, L = self.W.shape
        U = self.unfold_own()
        num = np.zeros((K, N, L))
        for l in range(L):
            num[:, :, l] = U[l * K:(l + 1) * K] @ self.data.T
        hh = U @ U.T
        self.gram_buf[: K * N * L] = torch.from_numpy(num.ravel())
        self.gram_buf[K * N * L: K * N * L + (L * K) ** 2] = torch.from_numpy(hh.ravel())

    def w_apply_gram(self, l1W, l2W):
        K, N, L = self.W.shape
        g = self.gram_buf.numpy()
        num = g[: K * N * L].reshape(K, N, L)
        HH = g[K * N * L: K * N * L + (L * K) ** 2].reshape(L * K, L * K)
        Wunf = np.concatenate([self.W[:, :, l] for l in range(L)], axis=0)  # row (l, k): W[k, :, l]
        den_unf = HH @ Wunf                                                   # mult.jl:33 as (H_unfold H_unfold') W
        den = np.stack([den_unf[l * K:(l + 1) * K] for l in range(L)], axis=2)
        self.W *= num / (((den + l1W) + (2.0 * l2W) * self.W) + EPS)
        np.maximum(self.W, EPS, out=self.W)

    def h_update_gram(self, l1H, l2H, t_offset, T_global):
        """numH from the data (own columns + right halo); denomH[k][t] = sum_{l: t + l < T} sum_{l', k'} PW[l][l'][k][k'] *
        H[k'][t + l - l'] with PW[l][l'] = W[:, :, l] W[:, :, l']' -- only W (replicated) and H with both halos."""
        K, N, L = self.W.shape
        off = L - 1
        Hext = np.concatenate([self.Hl, self.H, self.Hr], axis=1)
        next_ = self.Tl + self.halo_r
        num = np.zeros((K, self.Tl))
        for l in range(L):
            w = min(self.Tl, next_ - l)
            if w > 0:
                num[:, :w] += self.W[:, :, l] @ self.data_ext[:, l: l + w]
        den = np.zeros((K, self.Tl))
        for l in range(L):
            # columns t with t + l inside the global problem (the truncation of common.jl:71-81 at the right edge)
            w = min(self.Tl, T_global - t_offset - l)
            if w <= 0:
                continue
            for lp in range(L):
                PW = self.W[:, :, l] @ self.W[:, :, lp].T  # [k][k']
                # H[k'][t + l - l'] for t in [0, w): index into Hext = t + l - l' + off
                den[:, :w] += PW @ Hext[:, off + l - lp: off + l - lp + w]
        self.H *= num / (((den + l1H) + (2.0 * l2H) * self.H) + EPS)
        np.maximum(self.H, EPS, out=self.H)


class ProtocolShardedGram(ProtocolShardedMultUpdate):
    """group_update_motifs / group_update_feature_maps with option gram = 1: the all-reduce carries [numW | HH | tail]."""

    def __init__(self, data, W, H, group=None):
        super().__init__(data, W, H, GramShardEngine, group=group, halo_in_allreduce=False)  # (the Gram form keeps the halo all-gather)
        K, N, L = self.K, self.N, self.L
        self.LKN, self.HHsz = K * N * L, (L * K) ** 2
        self.gred = torch.zeros(self.LKN + self.HHsz + self.tail, dtype=torch.float64)
        self.engine.gram_buf = self.gred[: self.LKN + self.HHsz]

    def loss_partials(self):  # the tail sits behind [numW | HH] in this form (group_tail_off)
        from shard_protocol_cpu import split_hi_lo

        ss = self.engine.loss_partial()
        t = self.gred[self.LKN + self.HHsz:]
        t.zero_()
        t[2 * self.rank], t[2 * self.rank + 1] = split_hi_lo(ss)

    def loss_now(self):
        own = self.gred[self.LKN + self.HHsz + 2 * self.rank: self.LKN + self.HHsz + 2 * self.rank + 2].clone()
        allp = torch.zeros(2 * self.world, dtype=torch.float64)
        dist.all_gather_into_tensor(allp, own, group=self.group)
        return self.decode_tail(allp)

    def update_motifs(self, l1W=0.0, l2W=0.0, want_tail=False):
        self.engine.w_partial_gram()
        dist.all_reduce(self.gred, group=self.group)  # THE bulk exchange: [numW | HH | tail]
        tail = self.gred[self.LKN + self.HHsz:].clone() if want_tail else None
        self.engine.w_apply_gram(l1W, l2W)
        return tail

    def update_feature_maps(self, l1H=0.0, l2H=0.0, sync_loss=True):
        self.engine.h_update_gram(l1H, l2H, self.t0, self.T)
        self.exchange_halos()
        self.loss_partials()
        return math.sqrt(self.loss_now()) / self.data_norm if sync_loss else None


class ProtocolShardedPGD(ProtocolShardedMultUpdate):
    """group_pgd_w / group_pgd_h (pgd.jl:158-255 with T cut into column blocks): one all-reduce of the partial gradW; the
    squared norm of gradH, the component norms of UnitNormConstraint and the loss are sums over the ranks in rank order; the
    step-size state machine is replicated.  loss: "square" | "abs"; mask: the GLOBAL mask or None."""

    def __init__(self, data, W, H, mask=None, loss="square", group=None):
        super().__init__(data, W, H, OracleShardEngine, group=group, halo_in_allreduce=False)  # (the PGD rule keeps the halo all-gather)
        self.loss_kind = loss
        e = self.engine
        self.mask_ext = None if mask is None else np.asarray(mask, dtype=np.float64)[:, self.t0:self.t1 + e.halo_r]
        self.stepW = self.stepH = 5.0
        self.cur_loss = self.data_norm  # pgd.jl:151 (the norm, not its square)

    def _sum_ranks(self, vals):
        """doubles per rank -> their sums over all ranks, added in rank order (group_sum_doubles)."""
        v = torch.tensor(np.atleast_1d(vals), dtype=torch.float64)
        parts = [torch.zeros_like(v) for _ in range(self.world)]
        dist.all_gather(parts, v, group=self.group)
        tot = np.zeros(v.numel())
        for p in parts:
            tot += p.numpy()
        return tot

    def _resid_grad(self, ncols):
        """the loss gradient on the shard's first ncols columns (own, or own + right halo): 2 (est - data) or sign(est - data),
        times the mask (pgd.jl:230, :42-44, :64-67)"""
        e = self.engine
        est, _ = e._est(ncols)
        r = est - e.data_ext[:, :ncols]
        g = np.sign(r) if self.loss_kind == "abs" else 2.0 * r
        return g if self.mask_ext is None else g * self.mask_ext[:, :ncols]

    def _loss(self):
        e = self.engine
        est, _ = e._est(e.Tl)
        b, est = (e.data, est) if self.mask_ext is None else (self.mask_ext[:, :e.Tl] * e.data, self.mask_ext[:, :e.Tl] * est)
        mine = np.abs(b - est).sum() if self.loss_kind == "abs" else float(np.sum((b - est) ** 2))
        return float(self._sum_ranks(mine)[0])

    def _finish(self, step):
        lossv = self._loss()  # pgd.jl:245-247 over all shards
        step *= 1.05 if lossv < self.cur_loss else 0.70
        self.cur_loss = lossv
        return step

    @staticmethod
    def _project(x, constr, knorm2=None):
        if constr == "nonneg":
            np.maximum(x, EPS, out=x)
        elif constr == "unitnorm":
            for k in range(x.shape[0]):
                mag = math.sqrt(knorm2[k])
                if mag > 1:
                    x[k] /= mag

    def update_motifs(self, pen_sq=1.0, pen_abs=0.0, constr="nonneg"):
        e = self.engine
        K, N, L = e.W.shape
        g_est = self._resid_grad(e.Tl)
        Hext = np.concatenate([e.Hl, e.H, e.Hr], axis=1)
        off = L - 1
        part = np.zeros((K, N, L))
        for l in range(L):  # compute_gradW! (pgd.jl:206-214) over own t
            part[:, :, l] = Hext[:, off - l: off - l + e.Tl] @ g_est.T
        t = torch.from_numpy(part.ravel().copy())
        dist.all_reduce(t, group=self.group)  # the one bulk exchange of the W phase
        grad = t.numpy().reshape(K, N, L) + 2.0 * pen_sq * e.W + pen_abs * np.sign(e.W)
        e.W -= self.stepW / (np.linalg.norm(grad) + EPS) * grad  # replicated
        self._project(e.W, constr, [np.sum(e.W[k] ** 2) for k in range(K)])
        self.stepW = self._finish(self.stepW)

    def update_feature_maps(self, pen_sq=0.0, pen_abs=0.0, constr="nonneg"):
        e = self.engine
        K, N, L = e.W.shape
        next_ = e.Tl + e.halo_r
        g_ext = self._resid_grad(next_)  # own columns and the right lag halo
        grad = np.zeros((K, e.Tl))
        for l in range(L):  # compute_gradH! = tensor_transconv! (pgd.jl:218-221)
            w = min(e.Tl, next_ - l)
            if w > 0:
                grad[:, :w] += e.W[:, :, l] @ g_ext[:, l: l + w]
        grad += 2.0 * pen_sq * e.H + pen_abs * np.sign(e.H)
        nrm2 = float(self._sum_ranks(float(np.sum(grad ** 2)))[0])  # norm(gradH)^2 over all shards
        e.H -= self.stepH / (math.sqrt(nrm2) + EPS) * grad
        kn = self._sum_ranks([float(np.sum(e.H[k] ** 2)) for k in range(K)]) if constr == "unitnorm" else None
        self._project(e.H, constr, kn)
        self.exchange_halos()
        self.stepH = self._finish(self.stepH)
        return math.sqrt(self.cur_loss / self.data_norm ** 2)  # pgd.jl:201
