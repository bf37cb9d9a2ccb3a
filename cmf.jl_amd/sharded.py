"""T-sharded multiplicative-update rule, one process per GPU (torchrun-style launchers; bench.py --gpus N).

The sharded iteration itself lives in libcmf_hip.so (csrc/cmf_groups.hip): rank r owns a contiguous block of
columns of ``data`` and ``H``, ``W`` is replicated, and per MU iteration the ranks meet twice -- ONE RCCL
all-reduce of the [numW | denomW] partial sums (its tail carries every rank's loss partial of the previous
iteration) and one all-gather of the (L-1)-column H halos.  This module only

  * cuts the global arrays into the rank's block (:func:`partition`),
  * forms the library's communicator: ``transport="rccl"`` hands rank 0's ncclUniqueId to the other ranks through
    the torch.distributed process group (any backend -- it is used for this rendezvous only) and calls
    cmf_comm_init_rccl; ``transport="host"`` registers callbacks that perform the two collectives with
    torch.distributed on host buffers (gloo: several ranks can then share one GPU, which is how the multi-rank
    path is tested on a one-GPU box),
  * and forwards the rule methods to the same C entries the single-GPU rule uses
    (cmf_update_motifs / cmf_update_feature_maps / cmf_iterate).

The single-process form of the same group (one Julia task / Python process driving several GPUs) is
``MultUpdate(data, W, H, devices=[...])`` in host.py (cmf_create_multi).
"""
from __future__ import annotations

import ctypes
import math

import numpy as np

from . import _lib
from ._lib import CMFError, check, farr, ptr
from .host import MultUpdate, PGDUpdate


def partition(T, world, L):
    """Contiguous column blocks [t0, t1) per rank (the library's rule: blocks of ceil(T / world) columns); every
    block must hold at least L-1 columns."""
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


class ShardedMultUpdate(MultUpdate):
    """MultUpdate (src/algs/mult.jl) with the T axis sharded over the ranks of a torch.distributed process group.

    ``data``, ``W``, ``H`` are the GLOBAL arrays (every rank passes the same ones; each keeps only its block).
    The rule methods have the single-GPU rule's signatures and return the global loss on every rank."""

    def __init__(self, data, W, H, device=None, group=None, overlap=False, transport=None, fallback_to_host=False):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.pg = torch, dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        if transport is None:
            transport = "rccl" if self.backend == "nccl" else "host"
        if transport not in ("rccl", "host"):
            raise ValueError("transport must be 'rccl' or 'host'")
        self.transport = transport
        data = np.asarray(data)
        W = farr(W)
        H = np.asarray(H)
        if W.ndim != 3:
            raise ValueError("W must be a K x N x L tensor")
        K, N, L = W.shape
        T = data.shape[1]
        if data.shape[0] != N or H.shape != (K, T):
            raise ValueError("DimensionMismatch between data, W and H")
        self.N, self.T, self.K, self.L = N, T, K, L
        self.bounds = partition(T, self.world, L)
        t0, t1 = self.bounds[self.rank]
        self.t0, self.t1 = t0, t1
        halo_r = min(L - 1, T - t1)
        self.device = _lib.default_device() if device is None else int(device)
        self.devices = None
        self._lib = lib = _lib.load()
        self._h = ctypes.c_void_p()
        data_local = farr(data[:, t0:t1 + halo_r])
        # the L-1 columns of data in front of the shard: with them resident on every rank the group carries the halo of H in its W-phase
        # all-reduce (one collective per iteration) instead of in an all-gather of its own (cmf_shard_set_left_data)
        data_left = farr(data[:, t0 - (L - 1):t0]) if (t0 >= L - 1 > 0) else None

        def make_shard():
            check(lib.cmf_create_shard(ctypes.byref(self._h), self.device, N, t1 - t0, K, L, ptr(data_local), t0, T))
            if data_left is not None:
                check(lib.cmf_shard_set_left_data(self._h, ptr(data_left)))

        make_shard()
        self.transport_fallback = None
        try:
            if self.transport == "rccl":
                # All ranks must end up on the same transport, and none may enter the blocking ncclCommInitRank alone:
                # the ranks agree through the process group before and after it (_attach_rccl).  If any rank failed,
                # every rank rebuilds its shard and takes the host-collective transport instead -- or raises.
                bad = self._attach_rccl()
                if bad and not fallback_to_host:
                    raise CMFError(_lib.CMF_ERR_COMM, f"the RCCL communicator could not be formed: {bad[0]}")
                if bad:
                    self.transport_fallback = bad[0]
                    lib.cmf_destroy(self._h)
                    self._h = ctypes.c_void_p()
                    make_shard()
                    self.transport = "host"
                    self._attach()
            else:
                self._attach()
            check(lib.cmf_set_factors(self._h, ptr(W), ptr(farr(H[:, t0:t1]))))
            if overlap:
                self.set_overlap(True)
        except Exception:
            self.close()
            raise
        ss = ctypes.c_double()
        check(lib.cmf_get_data_sumsq(self._h, ctypes.byref(ss)))  # over all shards
        self.data_norm = math.sqrt(ss.value)  # mult.jl:13

    # ---- communicator ---------------------------------------------------------------------------
    def _src_rank(self):
        return self.dist.get_process_group_ranks(self.pg)[0] if self.pg is not None else 0

    def _agree(self, err):
        """Every rank's error text (None = fine) on every rank: the list of failures, empty when all succeeded."""
        flags = [None] * self.world
        self.dist.all_gather_object(flags, err, group=self.pg)
        return [f"rank {r}: {f}" for r, f in enumerate(flags) if f]

    def _attach_rccl(self):
        """Form the library's RCCL communicator in two agreed stages; returns the failures of all ranks ([] = attached).

        Stage 1: every rank checks that it can bind RCCL at all (rank 0 by creating the ncclUniqueId, the others with
        cmf_rccl_version) and rank 0 ALWAYS broadcasts -- the id or None -- so that no rank waits in a broadcast that
        never comes; then all ranks exchange a ready flag.  Stage 2: only if every rank is ready do they call
        cmf_comm_init_rccl (ncclCommInitRank blocks until all ranks have entered it), and exchange the outcome."""
        lib, dist = self._lib, self.dist
        err, payload = None, None
        try:
            if self.rank == 0:
                buf = ctypes.create_string_buffer(128)
                check(lib.cmf_comm_unique_id(buf))
                payload = buf.raw
            else:
                v = ctypes.c_int()
                check(lib.cmf_rccl_version(ctypes.byref(v), None, 0))
        except Exception as e:  # noqa: BLE001 - travels to the other ranks as text
            err = repr(e)
        box = [payload]
        dist.broadcast_object_list(box, src=self._src_rank(), group=self.pg)
        bad = self._agree(err)
        if bad:
            return bad
        try:
            idbuf = ctypes.create_string_buffer(box[0], 128)
            check(lib.cmf_comm_init_rccl(self._h, self.world, self.rank, idbuf))
        except Exception as e:  # noqa: BLE001
            err = repr(e)
        return self._agree(err)

    def _attach(self):
        """The host-collective transport: callbacks that run the two collectives with torch.distributed."""
        lib, dist = self._lib, self.dist
        torch, world, pg = self.torch, self.world, self.pg

        on_device = self.backend == "nccl"  # torch's RCCL backend only takes device tensors: bounce through the GPU
        dev = torch.device("cuda", self.device) if on_device else None

        def allreduce(_user, buf, count):
            try:
                t = torch.from_numpy(np.ctypeslib.as_array(buf, shape=(count,)))
                if on_device:
                    td = t.to(dev)
                    dist.all_reduce(td, group=pg)
                    t.copy_(td)
                else:
                    dist.all_reduce(t, group=pg)
                return 0
            except Exception as e:  # an exception must not unwind through the C frames
                print(f"cmf all-reduce callback failed: {e!r}", flush=True)
                return 1

        def allgather(_user, send, recv, count):
            try:
                s = torch.from_numpy(np.ctypeslib.as_array(send, shape=(count,)))
                r = torch.from_numpy(np.ctypeslib.as_array(recv, shape=(world * count,)))
                if on_device:
                    rd = torch.empty(world * count, dtype=torch.float32, device=dev)
                    dist.all_gather_into_tensor(rd, s.to(dev), group=pg)
                    r.copy_(rd)
                else:
                    dist.all_gather_into_tensor(r, s.clone(), group=pg)
                return 0
            except Exception as e:
                print(f"cmf all-gather callback failed: {e!r}", flush=True)
                return 1

        # the CFUNCTYPE objects must outlive the handle
        self._cb = (_lib.ALLREDUCE_FN(allreduce), _lib.ALLGATHER_FN(allgather))
        check(lib.cmf_comm_init_callbacks(self._h, self.world, self.rank, self._cb[0], self._cb[1], None))

    def set_overlap(self, flag):
        """The overlap form's bulk all-reduce runs on a second stream while collectives of the main stream are issued: in
        the RCCL transport that stream gets a communicator of its own (cmf_comm_init_overlap), formed here -- once, on all
        ranks together, from a second ncclUniqueId of rank 0 -- before the option is switched on.  If it cannot be formed on
        some rank, every rank stays with the single-stream form (and says so in `overlap_refused`)."""
        if flag and self.transport == "rccl" and not getattr(self, "_lane1", False):
            lib, dist = self._lib, self.dist
            err, payload = None, None
            if self.rank == 0:
                try:
                    buf = ctypes.create_string_buffer(128)
                    check(lib.cmf_comm_unique_id(buf))
                    payload = buf.raw
                except Exception as e:  # noqa: BLE001
                    err = repr(e)
            box = [payload]
            dist.broadcast_object_list(box, src=self._src_rank(), group=self.pg)
            bad = self._agree(err)
            if not bad:
                try:
                    check(lib.cmf_comm_init_overlap(self._h, ctypes.create_string_buffer(box[0], 128)))
                except Exception as e:  # noqa: BLE001
                    err = repr(e)
                bad = self._agree(err)
            if bad:
                self.overlap_refused = bad[0]
                flag = False
            else:
                self._lane1 = True
        MultUpdate.set_overlap(self, flag)

    overlap_refused = None

    # ---- host-side helpers ------------------------------------------------------------------------
    def agree_scalar(self, x):
        """Rank 0's value on every rank (keeps host-side stop decisions of fit() identical across ranks)."""
        box = [float(x)]
        self.dist.broadcast_object_list(box, src=self._src_rank(), group=self.pg)
        return box[0]

    def upload(self, W, H):
        W = farr(W, (self.K, self.N, self.L))
        H = farr(np.asarray(H)[:, self.t0:self.t1], (self.K, self.t1 - self.t0))
        check(self._lib.cmf_set_factors(self._h, ptr(W), ptr(H)))

    def download(self, W=None, H=None):
        """Gather the factors: W from the local replica, H by all-gathering the blocks."""
        Wl = np.zeros((self.K, self.N, self.L), order="F")
        Hl = np.zeros((self.K, self.t1 - self.t0), order="F")
        check(self._lib.cmf_get_factors(self._h, ptr(Wl), ptr(Hl)))
        blocks = [None] * self.world
        self.dist.all_gather_object(blocks, np.ascontiguousarray(Hl), group=self.pg)
        Hg = np.concatenate(blocks, axis=1)
        if W is not None:
            W[...] = Wl
            Wl = W
        if H is not None:
            H[...] = Hg
            Hg = H
        return Wl, Hg


class ShardedPGDUpdate(ShardedMultUpdate, PGDUpdate):
    """PGDUpdate (src/algs/pgd.jl:112-202) with the T axis sharded over the ranks of a torch.distributed process group:
    the same construction as ShardedMultUpdate, the rule methods of PGDUpdate (cmf_pgd_update_motifs /
    cmf_pgd_update_feature_maps on a group handle).  ``MaskedLoss(loss, mask)`` takes the GLOBAL N x T mask; each rank
    uploads its block with the right lag halo, like data."""

    def __init__(self, data, W, H, **kw):
        ShardedMultUpdate.__init__(self, data, W, H, **kw)
        check(self._lib.cmf_pgd_reset(self._h))
        self._mask_key = None

    def _upload_mask(self, mask):
        if mask is None:
            check(self._lib.cmf_set_mask(self._h, None))
            return
        halo_r = min(self.L - 1, self.T - self.t1)
        block = farr(np.asarray(mask)[:, self.t0:self.t1 + halo_r])
        check(self._lib.cmf_set_mask(self._h, ptr(block)))

    def fit_native(self, *a, **kw):
        return PGDUpdate.fit_native(self, *a, **kw)

    def iterate(self, *a, **kw):
        return PGDUpdate.iterate(self, *a, **kw)
