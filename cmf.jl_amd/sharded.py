"""T-sharded multiplicative-update rule: one process per GPU, torch.distributed (RCCL over xGMI).

Partition: rank r owns a contiguous block of columns of ``data`` and
``H``; ``W`` is replicated.  Per MU iteration each rank runs the same kernels as the single-GPU
rule on its block and the ranks meet three times:

    update_motifs!        est = conv(W,H) on own columns; local [numW | denomW]   (mult.jl:28-34)
                          ONE all-reduce(sum) of that 2*L*Kpad*Npad fp32 buffer    <- the only bulk exchange
                          identical W update on every rank                         (mult.jl:37-38)
    update_feature_maps!  est on own columns + the right lag halo, numH, denomH, H update (mult.jl:44-52)
                          (L-1)-column H halo exchange: one all-gather of 2 x 2.4 KB per rank
                          local sum((conv(W,H) - data).^2); all-reduce of one scalar (mult.jl:55-57)

Global edges keep the reference's truncation (no halo = zeros).  The orchestration below only
talks to an *engine* object (the per-rank compute), so the same code runs on the HIP engine
(product) and, in tests/, on a CPU stand-in to check the partition / halo / reduction protocol
with the gloo backend.
"""
from __future__ import annotations

import ctypes
import math

import numpy as np

from . import _lib
from ._lib import check, farr, ptr
from .host import AbstractCFUpdate


def partition(T, world, L):
    """Contiguous column blocks [t0, t1) per rank; every block holds at least L-1 columns."""
    base = -(-T // world)
    if world > 1 and base < max(L - 1, 1):
        raise ValueError(f"T={T} is too short to shard over {world} ranks with L={L}")
    bounds = []
    for r in range(world):
        t0 = min(r * base, T)
        t1 = min(t0 + base, T)
        bounds.append((t0, t1))
    if world > 1 and bounds[-1][1] - bounds[-1][0] < max(L - 1, 1):
        raise ValueError(f"last shard of T={T} over {world} ranks is shorter than L-1={L - 1}")
    return bounds


class HipShardEngine:
    """Per-rank compute on libcmf_hip.so (phase-split C ABI).  Buffers that cross ranks are torch
    tensors whose storage the library writes into directly (cmf_set_numden_buffer /
    cmf_set_halo_buffer), and the library runs on torch's current stream, so collectives and
    kernels are ordered without extra synchronisation."""

    def __init__(self, data_local, W, H_local, t_offset, T_global, device):
        import torch

        self.torch = torch
        self._lib = lib = _lib.load()
        self._h = ctypes.c_void_p()
        K, N, L = W.shape
        self.K, self.N, self.L = K, N, L
        self.T_local = H_local.shape[1]
        self.device = int(device)
        data_local = farr(data_local)
        check(lib.cmf_create_shard(ctypes.byref(self._h), self.device, N, self.T_local, K, L, ptr(data_local),
                                   int(t_offset), int(T_global)))
        dev = torch.device("cuda", self.device)
        torch.cuda.set_device(dev)
        check(lib.cmf_set_stream(self._h, ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        p, n = ctypes.c_void_p(), ctypes.c_int64()
        check(lib.cmf_numden_ptr(self._h, ctypes.byref(p), ctypes.byref(n)))
        self.numden = torch.zeros(n.value, dtype=torch.float32, device=dev)
        check(lib.cmf_set_numden_buffer(self._h, ctypes.c_void_p(self.numden.data_ptr())))
        check(lib.cmf_halo_ptr(self._h, 0, ctypes.byref(p), ctypes.byref(n)))
        self.halo_count = max(n.value, 1)
        # [send-to-left | send-to-right] is one contiguous tensor, so the exchange is ONE all-gather; the
        # receive sides are pointed straight into the gathered buffer by attach_gathered_halos()
        self.halo_send = torch.zeros(2 * self.halo_count, dtype=torch.float32, device=dev)
        self.halo = [self.halo_send[: self.halo_count], self.halo_send[self.halo_count:], None, None]
        for w in range(2):
            check(lib.cmf_set_halo_buffer(self._h, w, ctypes.c_void_p(self.halo[w].data_ptr())))
        self.scalar = torch.zeros(4, dtype=torch.float64, device=dev)  # [0] = local sum of squared residuals
        check(lib.cmf_set_scalar_buffer(self._h, ctypes.c_void_p(self.scalar.data_ptr())))
        self.set_factors(W, H_local)

    def data_sumsq(self):
        v = ctypes.c_double()
        check(self._lib.cmf_get_data_sumsq(self._h, ctypes.byref(v)))
        return v.value

    def set_data_norm(self, x):
        check(self._lib.cmf_set_data_norm(self._h, float(x)))

    def set_factors(self, W, H_local):
        W = farr(W, (self.K, self.N, self.L))
        H = farr(H_local, (self.K, self.T_local))
        check(self._lib.cmf_set_factors(self._h, ptr(W), ptr(H)))

    def get_factors(self):
        W = np.zeros((self.K, self.N, self.L), order="F")
        H = np.zeros((self.K, self.T_local), order="F")
        check(self._lib.cmf_get_factors(self._h, ptr(W), ptr(H)))
        return W, H

    def w_partial(self):
        check(self._lib.cmf_w_partial(self._h))

    def w_partial_num(self):
        check(self._lib.cmf_w_partial_num(self._h))

    def w_partial_den(self):
        check(self._lib.cmf_w_partial_den(self._h))

    def w_apply(self, l1W, l2W):
        check(self._lib.cmf_w_apply(self._h, float(l1W), float(l2W)))

    def h_update(self, l1H, l2H):
        check(self._lib.cmf_h_update(self._h, float(l1H), float(l2H)))

    def halo_pack(self):
        check(self._lib.cmf_halo_pack(self._h))

    def attach_gathered_halos(self, gathered, rank, world):
        """gathered = all ranks' [send-to-left | send-to-right] blocks: my left halo is the left neighbour's
        send-to-right block, my right halo the right neighbour's send-to-left block (no copies)."""
        c = self.halo_count
        self._gathered = gathered
        if rank > 0:
            self.halo[2] = gathered[(2 * (rank - 1) + 1) * c: (2 * (rank - 1) + 2) * c]
            check(self._lib.cmf_set_halo_buffer(self._h, 2, ctypes.c_void_p(self.halo[2].data_ptr())))
        if rank < world - 1:
            self.halo[3] = gathered[(2 * (rank + 1)) * c: (2 * (rank + 1) + 1) * c]
            check(self._lib.cmf_set_halo_buffer(self._h, 3, ctypes.c_void_p(self.halo[3].data_ptr())))

    def halo_unpack(self, has_left, has_right):
        check(self._lib.cmf_halo_unpack(self._h, int(has_left), int(has_right)))

    def loss_partial(self):
        v = ctypes.c_double()
        check(self._lib.cmf_loss_partial(self._h, ctypes.byref(v)))
        return v.value

    def loss_partial_tensor(self):
        """Enqueue the local sum of squared residuals; returns the 1-element device tensor holding it."""
        check(self._lib.cmf_loss_partial_async(self._h))
        return self.scalar[:1]

    def set_option(self, name, value):
        check(self._lib.cmf_set_option(self._h, name.encode(), int(value)))

    def kernel_times(self, name):
        ms, n = ctypes.c_double(), ctypes.c_int64()
        check(self._lib.cmf_kernel_times(self._h, name.encode(), ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def time_kernel(self, name, reps=5):
        ms, fl = ctypes.c_double(), ctypes.c_double()
        check(self._lib.cmf_time_kernel(self._h, name.encode(), int(reps), ctypes.byref(ms), ctypes.byref(fl)))
        return ms.value, fl.value

    def close(self):
        if self._h:
            self.torch.cuda.synchronize(self.device)
            self._lib.cmf_destroy(self._h)
            self._h = ctypes.c_void_p()


class ShardedMultUpdate(AbstractCFUpdate):
    """MultUpdate (src/algs/mult.jl) with the T axis sharded over the ranks of a process group.

    ``data``, ``W``, ``H`` are the GLOBAL arrays (every rank passes the same ones; each keeps only
    its block).  The rule methods have the single-GPU rule's signatures and return the global loss
    on every rank."""

    def __init__(self, data, W, H, device=None, group=None, engine_cls=HipShardEngine, overlap=False):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        data = np.asarray(data)
        W = np.asarray(W)
        H = np.asarray(H)
        K, N, L = W.shape
        T = data.shape[1]
        if data.shape[0] != N or H.shape != (K, T):
            raise ValueError("DimensionMismatch between data, W and H")
        self.N, self.T, self.K, self.L = N, T, K, L
        self.bounds = partition(T, self.world, L)
        t0, t1 = self.bounds[self.rank]
        self.t0, self.t1 = t0, t1
        self.has_left = self.rank > 0
        self.has_right = self.rank < self.world - 1
        halo_r = min(L - 1, T - t1)
        dev = _lib.default_device() if device is None else device
        self._gathered = None
        # overlap=True: numW (which needs H only) is contracted and all-reduced right after the H update, underneath
        # the loss conv and the denominator contraction, so only the denomW half of the all-reduce stays exposed
        self.overlap = bool(overlap)
        self._num_work = None     # pending async all-reduce of the numW half
        self._num_ready = False   # the numW half of the buffer belongs to the current H
        self.engine = engine_cls(data[:, t0:t1 + halo_r], W, H[:, t0:t1], t0, T, dev)
        # data_norm = norm(data) over all shards (mult.jl:13)
        ss = self._allreduce_scalar(self.engine.data_sumsq())
        self.data_norm = math.sqrt(ss)
        self.engine.set_data_norm(self.data_norm)
        self._exchange_halos()

    # ---- collectives ----------------------------------------------------------------------
    def _allreduce(self, t):
        """In-place sum over ranks; device tensors go through RCCL, or through the host on gloo."""
        if t.is_cuda and self.backend != "nccl":
            tmp = t.cpu()
            self.dist.all_reduce(tmp, group=self.group)
            t.copy_(tmp)
        else:
            self.dist.all_reduce(t, group=self.group)

    def _allreduce_scalar(self, x):
        dev = self.engine.numden.device if self.backend == "nccl" else "cpu"
        t = self.torch.tensor([x], dtype=self.torch.float64, device=dev)
        self.dist.all_reduce(t, group=self.group)
        return float(t.item())

    def _exchange_halos(self):
        """Own first/last L-1 columns of H -> neighbours' right/left halos: one small
        all-gather of every rank's [first | last] block (2*(L-1)*Kpad floats per rank)."""
        if self.L < 2 or self.world == 1:
            return
        eng, dist = self.engine, self.dist
        eng.halo_pack()
        send = eng.halo_send
        if self._gathered is None:
            self._gathered = self.torch.zeros(self.world * send.numel(), dtype=send.dtype, device=send.device)
            eng.attach_gathered_halos(self._gathered, self.rank, self.world)
        if send.is_cuda and self.backend != "nccl":
            tmp_in = send.cpu()
            tmp_out = self.torch.zeros(self.world * send.numel(), dtype=send.dtype)
            dist.all_gather_into_tensor(tmp_out, tmp_in, group=self.group)
            self._gathered.copy_(tmp_out)
        else:
            dist.all_gather_into_tensor(self._gathered, send, group=self.group)
        eng.halo_unpack(self.has_left, self.has_right)

    # ---- the rule ---------------------------------------------------------------------------
    def _allreduce_async(self, t):
        """Sum over ranks without blocking the compute stream (RCCL); returns a work handle or None when done."""
        if t.is_cuda and self.backend == "nccl":
            return self.dist.all_reduce(t, group=self.group, async_op=True)
        self._allreduce(t)
        return None

    def set_overlap(self, flag):
        """Switch between the two forms of the W phase; a numW all-reduce still in flight is completed first."""
        if self._num_work is not None:
            self._num_work.wait()
            self._num_work = None
        self._num_ready = False
        self.overlap = bool(flag)

    def _start_num(self):
        half = self.engine.numden.numel() // 2
        self.engine.w_partial_num()
        self._num_work = self._allreduce_async(self.engine.numden[:half])
        self._num_ready = True

    def update_motifs(self, data=None, W=None, H=None, l1W=0, l2W=0, **kwargs):
        """update_motifs!: mult.jl:23-39, with the single all-reduce of [numW | denomW] (in two halves when
        overlap is on: the numW half is usually already in flight, started by the previous update_feature_maps!)."""
        if self.overlap:
            half = self.engine.numden.numel() // 2
            if not self._num_ready:
                self._start_num()
            self.engine.w_partial_den()
            self._allreduce(self.engine.numden[half:])
            if self._num_work is not None:
                self._num_work.wait()
                self._num_work = None
            self._num_ready = False
        else:
            self.engine.w_partial()
            self._allreduce(self.engine.numden)
        self.engine.w_apply(l1W, l2W)

    def update_feature_maps(self, data=None, W=None, H=None, l1H=0, l2H=0, **kwargs):
        """update_feature_maps!: mult.jl:42-58 -> global loss."""
        self.engine.h_update(l1H, l2H)
        self._exchange_halos()
        if self.overlap:
            self._start_num()  # for the next update_motifs!: H and its halos are final now
        return self.compute_loss()

    def compute_loss(self):
        if hasattr(self.engine, "loss_partial_tensor"):
            # stays on the device until the single .item() below: one host sync per iteration
            t = self.engine.loss_partial_tensor()
            self._allreduce(t)
            ss = float(t.item())
        else:
            ss = self._allreduce_scalar(self.engine.loss_partial())
        return math.sqrt(ss) / self.data_norm

    def agree_scalar(self, x):
        """Rank 0's value on every rank (keeps host-side stop decisions identical across ranks)."""
        box = [float(x)]
        self.dist.broadcast_object_list(box, src=(self.dist.get_process_group_ranks(self.group)[0]
                                                  if self.group is not None else 0), group=self.group)
        return box[0]

    def download(self, W=None, H=None):
        """Gather the factors: W from the local replica, H by all-gathering the blocks."""
        Wl, Hl = self.engine.get_factors()
        blocks = [None] * self.world
        self.dist.all_gather_object(blocks, np.ascontiguousarray(Hl), group=self.group)
        Hg = np.concatenate(blocks, axis=1)
        if W is not None:
            W[...] = Wl
            Wl = W
        if H is not None:
            H[...] = Hg
            Hg = H
        return Wl, Hg

    def close(self):
        if self._num_work is not None:
            self._num_work.wait()
            self._num_work = None
        self.engine.close()
