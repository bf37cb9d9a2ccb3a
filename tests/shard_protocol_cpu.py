"""CPU mirror of the library's T-sharded group iteration (TEST INFRASTRUCTURE).

csrc/cmf_groups.hip runs the sharded MU iteration on the GPU; it cannot execute where there is no GPU.  This file
states the same protocol step by step in Python over torch.distributed (gloo), with the numpy stand-in engine of
tests/shard_engine_cpu.py computing each rank's block, so that the partition, the single all-reduce of
[numW | denomW | tail | halo slots] (round 6: ONE collective per iteration -- every shard with a left neighbour updates the L-1
columns in front of its own itself and the outer columns of the new H ride behind the loss tail of the next all-reduce;
halo_in_allreduce = False: the H halo all-gather of rounds 1-5), the (hi, lo) own-slot encoding of the loss scalar and the one-
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
    def __init__(self, data, W, H, engine_cls, overlap=False, group=None, halo_in_allreduce=True):
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
        h = L - 1
        # group_finish_norm's agreement: every shard holds at least 2(L-1) columns (the mirror has no K % 32 condition: that is the kernels')
        self.halo_can = self.world > 1 and h >= 1 and all(b - a >= 2 * h for a, b in self.bounds)
        self.halo_opt = bool(halo_in_allreduce)
        self.halos_current = self.halos_pending = self.halo_wide = False
        self.collectives = {"all_reduce": 0, "all_gather": 0}
        if self.halo_can and self.halo_opt and t0 > 0:
            self.engine = eng = engine_cls(data[:, t0:t1 + halo_r], W, H[:, t0:t1], t0, T, 0, data_left=data[:, t0 - h:t0])
        else:
            self.engine = eng = engine_cls(data[:, t0:t1 + halo_r], W, H[:, t0:t1], t0, T, 0)
        # group_alloc_buffers: red = [numW | denomW | tail], tail = 2 floats per rank rounded up to 64
        self.LKN2 = eng.numden.numel()
        self.tail = -(-2 * self.world // 64) * 64
        self.slot = getattr(eng, "slot_count", 0) if self.halo_can else 0
        self.red = torch.zeros(self.LKN2 + self.tail + self.world * self.slot, dtype=torch.float64)
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
    def halo_in_ar(self):
        return self.halo_opt and self.halo_can and not getattr(self, "gram", False)

    def _slot(self, buf, r):
        return buf[r * self.slot:(r + 1) * self.slot] if 0 <= r < self.world else None

    def exchange_halos(self):
        self.halos_current, self.halos_pending, self.halo_wide = True, False, False
        if self.L < 2 or self.world == 1:
            return
        self.collectives["all_gather"] += 1
        if self.halo_in_ar():  # the wide exchange as an all-gather of the slots (set-up; whenever no all-reduce follows an H phase)
            self.halo_wide = True
            allb = torch.zeros(self.world * self.slot, dtype=torch.float64)
            dist.all_gather_into_tensor(allb, self.engine.halo_pack3(), group=self.group)
            self.engine.halo_unpack3(self._slot(allb, self.rank - 1), self._slot(allb, self.rank + 1))
            return
        self.engine.halo_pack()
        dist.all_gather_into_tensor(self.halo_all, self.engine.halo_send.clone(), group=self.group)
        self.engine.halo_unpack(self.rank > 0, self.rank < self.world - 1)

    # ---- group_loss_partials / group_loss_now -------------------------------------------------------
    def loss_partials(self):
        ss = self.engine.loss_partial()
        t = self.red[self.LKN2:self.LKN2 + self.tail]
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
        self.collectives["all_gather"] += 1
        dist.all_gather_into_tensor(allp, own, group=self.group)
        return self.decode_tail(allp)

    # ---- group_update_motifs / group_update_feature_maps -----------------------------------------------
    def start_num(self):
        self.engine.w_partial_num()
        half = self.LKN2 // 2
        t = self.red[:half].clone()
        self.collectives["all_reduce"] += 1
        dist.all_reduce(t, group=self.group)  # on the communication stream in the library
        self.red[:half] = t
        self.num_ready = True

    def update_motifs(self, l1W=0.0, l2W=0.0, want_tail=False):
        half = self.LKN2 // 2
        if not self.halos_current and not self.halos_pending:  # (pending: the L-1 columns in front are valid -- all this phase reads)
            self.exchange_halos()
        carry = self.halo_in_ar() and self.halos_pending       # the halos of the H phase before ride behind the loss tail
        end = self.LKN2 + self.tail + (self.world * self.slot if carry else 0)
        if self.overlap:
            if not self.num_ready:
                self.start_num()
            self.engine.w_partial_den()
            t = self.red[half:end].clone()
            self.collectives["all_reduce"] += 1
            dist.all_reduce(t, group=self.group)
            self.red[half:end] = t
            self.num_ready = False
        else:
            self.engine.w_partial()
            t = self.red[:end].clone()
            self.collectives["all_reduce"] += 1
            dist.all_reduce(t, group=self.group)  # THE bulk exchange: [numW | denomW | tail | halo slots]
            self.red[:end] = t
        tail = self.red[self.LKN2:self.LKN2 + self.tail].clone() if want_tail else None
        self.engine.w_apply(l1W, l2W)
        if carry:  # every rank's outer columns of the new H have arrived with the sums
            slots = self.red[self.LKN2 + self.tail:]
            self.engine.halo_unpack3(self._slot(slots, self.rank - 1), self._slot(slots, self.rank + 1))
            self.halos_current, self.halos_pending, self.halo_wide = True, False, True
        return tail

    def update_feature_maps(self, l1H=0.0, l2H=0.0, sync_loss=True):
        in_ar = self.halo_in_ar()
        if not self.halos_current or (in_ar and not self.halo_wide):
            self.exchange_halos()
        if in_ar and self.rank > 0:
            self.engine.h_update_front(l1H, l2H)
        else:
            self.engine.h_update(l1H, l2H)
        self.num_ready = False
        if in_ar:  # no exchange: own slot <- own outer columns, every other rank's slot <- 0 (halo_pack3_kernel)
            slots = self.red[self.LKN2 + self.tail:]
            slots.zero_()
            slots[self.rank * self.slot:(self.rank + 1) * self.slot] = self.engine.halo_pack3()
            self.halos_current, self.halos_pending, self.halo_wide = False, True, False
        else:
            self.exchange_halos()
        if self.overlap:
            self.start_num()
        self.loss_partials()
        return math.sqrt(self.loss_now()) / self.data_norm if sync_loss else None

    def compute_loss(self):
        if not self.halos_current and not self.halos_pending:
            self.exchange_halos()
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
