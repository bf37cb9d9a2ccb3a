"""CPU mirror of the library's T-sharded group iteration (TEST INFRASTRUCTURE).

csrc/cmf_groups.hip runs the sharded MU iteration on the GPU; it cannot execute where there is no GPU.  This file
states the same protocol step by step in Python over torch.distributed (gloo), with the numpy stand-in engine of
tests/shard_engine_cpu.py computing each rank's block, so that the partition, the single all-reduce of
[numW | denomW | tail], the H halo all-gather, the (hi, lo) own-slot encoding of the loss scalar and the one-
iteration-late loss read-out of cmf_iterate are checked against the unsharded oracle on CPU-only machines.
Function names follow cmf_groups.hip (group_update_motifs, group_update_feature_maps, group_iterate ...).
"""
import math

import numpy as np
import torch
import torch.distributed as dist

from cmf_jl_amd.sharded import partition


def split_hi_lo(x):
    """double -> two floats whose sum reproduces it to ~2^-48 (loss_tail_kernel)."""
    hi = np.float32(x)
    lo = np.float32(x - np.float64(hi))
    return float(hi), float(lo)


class ProtocolShardedMultUpdate:
    def __init__(self, data, W, H, engine_cls, overlap=False, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        data, W, H = np.asarray(data), np.asarray(W), np.asarray(H)
        K, N, L = W.shape
        T = data.shape[1]
        self.N, self.T, self.K, self.L = N, T, K, L
        self.bounds = partition(T, self.world, L)
        t0, t1 = self.bounds[self.rank]
        self.t0, self.t1 = t0, t1
        halo_r = min(L - 1, T - t1)
        self.engine = eng = engine_cls(data[:, t0:t1 + halo_r], W, H[:, t0:t1], t0, T, 0)
        # group_alloc_buffers: red = [numW | denomW | tail], tail = 2 floats per rank rounded up to 64
        self.LKN2 = eng.numden.numel()
        self.tail = -(-2 * self.world // 64) * 64
        self.red = torch.zeros(self.LKN2 + self.tail, dtype=torch.float64)
        eng.numden = self.red[: self.LKN2]  # the shard's numden points into the group's buffer
        self.halo_all = torch.zeros(self.world * 2 * eng.halo_count, dtype=torch.float64)
        eng.attach_gathered_halos(self.halo_all, self.rank, self.world)
        self.overlap = bool(overlap)
        self.num_ready = False
        # group_finish_norm: every rank's sum of squares, added in rank order
        parts = [torch.zeros(1, dtype=torch.float64) for _ in range(self.world)]
        dist.all_gather(parts, torch.tensor([eng.data_sumsq()], dtype=torch.float64), group=group)
        self.data_sumsq = 0.0
        for p in parts:
            self.data_sumsq += float(p[0])
        self.data_norm = math.sqrt(self.data_sumsq)
        eng.set_data_norm(self.data_norm)
        self.exchange_halos()

    # ---- group_exchange_halos ---------------------------------------------------------------------
    def exchange_halos(self):
        if self.L < 2 or self.world == 1:
            return
        self.engine.halo_pack()
        dist.all_gather_into_tensor(self.halo_all, self.engine.halo_send.clone(), group=self.group)
        self.engine.halo_unpack(self.rank > 0, self.rank < self.world - 1)

    # ---- group_loss_partials / group_loss_now -------------------------------------------------------
    def loss_partials(self):
        ss = self.engine.loss_partial()
        t = self.red[self.LKN2:]
        t.zero_()
        hi, lo = split_hi_lo(ss)
        t[2 * self.rank], t[2 * self.rank + 1] = hi, lo  # own slots; every other rank's slots stay 0

    def decode_tail(self, tail):
        s = 0.0
        for r in range(self.world):
            s += float(tail[2 * r]) + float(tail[2 * r + 1])
        return s

    def loss_now(self):
        own = self.red[self.LKN2 + 2 * self.rank: self.LKN2 + 2 * self.rank + 2].clone()
        allp = torch.zeros(2 * self.world, dtype=torch.float64)
        dist.all_gather_into_tensor(allp, own, group=self.group)
        return self.decode_tail(allp)

    # ---- group_update_motifs / group_update_feature_maps -----------------------------------------------
    def start_num(self):
        self.engine.w_partial_num()
        half = self.LKN2 // 2
        t = self.red[:half].clone()
        dist.all_reduce(t, group=self.group)  # on the communication stream in the library
        self.red[:half] = t
        self.num_ready = True

    def update_motifs(self, l1W=0.0, l2W=0.0, want_tail=False):
        half = self.LKN2 // 2
        if self.overlap:
            if not self.num_ready:
                self.start_num()
            self.engine.w_partial_den()
            t = self.red[half:].clone()
            dist.all_reduce(t, group=self.group)
            self.red[half:] = t
            self.num_ready = False
        else:
            self.engine.w_partial()
            dist.all_reduce(self.red, group=self.group)  # THE bulk exchange: [numW | denomW | tail]
        tail = self.red[self.LKN2:].clone() if want_tail else None
        self.engine.w_apply(l1W, l2W)
        return tail

    def update_feature_maps(self, l1H=0.0, l2H=0.0, sync_loss=True):
        self.engine.h_update(l1H, l2H)
        self.num_ready = False
        self.exchange_halos()
        if self.overlap:
            self.start_num()
        self.loss_partials()
        return math.sqrt(self.loss_now()) / self.data_norm if sync_loss else None

    def compute_loss(self):
        self.loss_partials()
        return math.sqrt(self.loss_now()) / self.data_norm

    # ---- group_iterate: the loss of iteration i rides on iteration i+1's all-reduce ----------------------
    def iterate(self, n, l1W=0.0, l2W=0.0, l1H=0.0, l2H=0.0):
        losses = np.zeros(n)
        for it in range(n):
            tail = self.update_motifs(l1W, l2W, want_tail=it > 0)
            last = it + 1 == n
            loss = self.update_feature_maps(l1H, l2H, sync_loss=last)
            if it > 0:
                losses[it - 1] = math.sqrt(self.decode_tail(tail)) / self.data_norm
            if last:
                losses[it] = loss
        return losses

    def download(self):
        Wl, Hl = self.engine.get_factors()
        blocks = [None] * self.world
        dist.all_gather_object(blocks, np.ascontiguousarray(Hl), group=self.group)
        return Wl, np.concatenate(blocks, axis=1)

    def close(self):
        self.engine.close()
