/*
 * cmf_hip.h -- C ABI of libcmf_hip.so: the MI355X (gfx950) implementation of
 * CMF.jl's convolutive-NMF multiplicative-update (MU) hot path.
 *
 * This is the drop-in boundary.  The reference is pure
 * Julia, so what a maintainer binds is a `ccall` per entry point
 * (INTEGRATION.md shows the Julia side: `struct HIPMultUpdate <: AbstractCFUpdate`).
 * Every entry cites the reference interface it replaces (paths relative to
 * the reference checkout).
 *
 * Conventions
 *   - All host arrays are Julia's own memory layout (column-major, first
 *     index fastest), Float64:
 *         data[n + N*t]          N x T      (Matrix{Float64})
 *         W[k + K*(n + N*l)]     K x N x L  (Array{Float64,3}, "Tensor")
 *         H[k + K*t]             K x T
 *     The device computes in fp32; conversion happens on upload/download.
 *   - Host pointers are borrowed for the duration of the call only.
 *   - Every entry returns 0 on success or a CMF_ERR_* code; the message is
 *     available from cmf_last_error() (thread-local).
 *   - A handle is not re-entrant: one host thread at a time.
 *   - There is no CPU fallback: without a usable HIP device every compute
 *     entry fails with CMF_ERR_HIP.
 *   - Environment.  Everything that selects a code path is an OPTION (cmf_set_option; cmf_option_names lists them).  The library
 *     reads these ten variables and no others:
 *         CMF_WAIT_TIMEOUT_S     bound of every host-side wait (loss words, helper threads, collectives): seconds, default 300
 *         CMF_WRITEBACK_THREADS  widening helpers of cmf_arm_writeback: default 4, at most 16
 *         CMF_ENQUEUE_THREADS    0: the calling thread enqueues every shard of a one-process group (see cmf_create_multi)
 *         CMF_RCCL_LIB           path of the RCCL to dlopen ("none": behave as if there were no RCCL)
 *         CMF_ROCTX              1: load the roctx marker library and bracket phases and collectives; 0: never; unset: only if the
 *                                process has already mapped it (a profiler)
 *         CMF_TEST_HOOKS         1: honour the test hooks below (and the option "hals_debug"); anything else: they do not exist
 *           CMF_MAX_COLUMNS        columns one handle holds before cmf_create cuts the recording into shards (tests of that path)
 *           CMF_TEST_FORCE_WORKERS 1: an enqueue worker for a ONE-shard RCCL group too
 *           CMF_TEST_FAIL_SHARD    that shard's next all-reduce call fails (tests of the failure path)
 *           CMF_TEST_NO_ARENA      1: every buffer of a handle is an allocation of its own (an overrun cannot hide in a neighbour)
 */
#ifndef CMF_HIP_H
#define CMF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CMF_OK 0
#define CMF_ERR_ARG 1         /* bad argument (null pointer, non-positive size, patience < 1 ...) */
#define CMF_ERR_HIP 2         /* HIP runtime error / no device */
#define CMF_ERR_STATE 3       /* call sequence error (e.g. factors not set) */
#define CMF_ERR_UNSUPPORTED 4 /* shape outside what the kernels implement */
#define CMF_ERR_COMM 5        /* RCCL / collective transport error */

typedef struct cmf_handle_s *cmf_handle;

/* Version of this interface.  3 (round 3): the phase-split entries of version 1 (cmf_w_partial*, cmf_w_apply,
 * cmf_h_update, cmf_loss_partial*, cmf_halo_*, cmf_numden_ptr, cmf_set_data_norm) are gone -- a sharded iteration
 * runs behind the rule entries of a group handle (cmf_create_multi / cmf_comm_init_*); cmf_abi_version,
 * cmf_source_digest, cmf_synchronize, cmf_rccl_version, cmf_get_counter are new.  5 (round 5): cmf_arm_writeback.  6 (round 6):
 * cmf_fingerprint, cmf_option_names, cmf_shard_set_left_data; cmf_set_factors takes one NULL factor; the measurement variables of the environment are gone
 * (CMF_HALS_*, CMF_CONV_*, CMF_SK_*, CMF_GRAM_FW, CMF_PGD_TRANSPOSE, CMF_LOSS_POLL, CMF_SPECULATE_W, CMF_SMALL_K, CMF_HXT_EXACT,
 * CMF_LOOPBACK_*): what tests still select is an option. */
#define CMF_ABI_VERSION 6
int cmf_abi_version(void);

/* Library / build identification: "cmf_hip gfx950 <version> abi=<n> src=<digest>". */
const char *cmf_version(void);
/* Hex SHA-256 prefix (16 characters) of the sources this library was compiled from (csrc/cmf_api.hip, cmf_rules.hip, cmf_groups.hip,
 * cmf_small.hip, cmf_internal.h, cmf_kernels.h, cmf_small_k.h, cmf_workers.h, cmf_writeback.h, cmf_rng.h, include/cmf_hip.h, in that order, each preceded by its base name and a newline).
 * A loader that has the tree at hand recomputes it and refuses (or rebuilds) a stale binary -- cmf.jl_amd/_lib.py does;
 * "unknown" when the library was built without the build script. */
const char *cmf_source_digest(void);
/* Message of the last failing call on this thread ("" if none). */
const char *cmf_last_error(void);
/* Number of visible HIP devices (0 if none / runtime unusable). */
int cmf_device_count(void);

/* ---- rule construction -------------------------------------------------
 * Replaces the MultUpdate constructor `MultUpdate(data, W, H)`
 * (src/algs/mult.jl:11-20, called at src/model.jl:79): uploads `data`
 * (N x T), allocates the rule's scratch (est, numW, denomW, numH, denomH)
 * on `device`, and computes data_norm = norm(data).
 * cmf_create is the single-GPU form of cmf_create_shard.  A recording with more columns than one handle's 32-bit buffer
 * offsets reach (8.3 M at K <= 64) is cut along T into several shards on the SAME device automatically (the handle then fronts
 * a loopback group, see cmf_create_multi): the MU and PGD rules run on it unchanged, HALS -- whose H sweep is one chain along
 * T -- reports that it cannot. */
int cmf_create(cmf_handle *h, int device, int64_t N, int64_t T, int64_t K, int64_t L,
               const double *data);

/* T-sharded form: this handle owns global columns
 * [t_offset, t_offset + T_local) of a T_global-column problem.
 * `data_local` holds columns [t_offset, t_offset + T_local + halo_r) where
 * halo_r = min(L-1, T_global - t_offset - T_local) (the static right halo of
 * `data` that tensor_transconv needs).  The shard joins its group with cmf_comm_init_rccl /
 * cmf_comm_init_callbacks (below), which also all-reduces data_norm. */
int cmf_create_shard(cmf_handle *h, int device, int64_t N, int64_t T_local, int64_t K, int64_t L,
                     const double *data_local, int64_t t_offset, int64_t T_global);
/* The L-1 columns of data in FRONT of a shard (N x (L-1), columns [t_offset - (L-1), t_offset); the shard with t_offset = 0 has none and
 * need not call this).  With them resident on every shard -- call it between cmf_create_shard and cmf_comm_init_* -- the group carries
 * the halo of H in the tail of its W-phase all-reduce (ONE collective per iteration; see the group section below) instead of in an
 * all-gather of its own behind every H update.  cmf_create_multi does this itself.  No reference counterpart. */
int cmf_shard_set_left_data(cmf_handle h, const double *cols);

/* ---- T-sharded groups: the same rule on several GPUs of one node -----------------
 * The T axis of data / est / H is cut into contiguous column blocks (shard r owns columns
 * [r*ceil(T/R), min(T, (r+1)*ceil(T/R)))), W is replicated.  Per MU iteration the shards meet ONCE: one RCCL
 * all-reduce of the [numW | denomW] partial sums (2*L*Kpad*Npad floats) + a tail that carries every shard's loss
 * partial of the previous iteration and -- since round 6 -- every shard's outer columns of the new H (its last 2(L-1) and
 * first L-1: 7.3 KB per rank at config 2), so that no exchange stands between the H update and the loss conv: every shard
 * with a left neighbour updates the L-1 columns in front of its own itself (0.6 - 2 % redundant work on a T/8 shard).
 * That form (option "halo_in_allreduce", default on) needs K a multiple of 32, 1 <= L-1 <= 64, shards of at least 2(L-1)
 * columns and the default formulation; otherwise, and for the Gram form and the PGD rule, the (L-1)-column halos travel in
 * an all-gather of their own behind every H update (2 x 2.4 KB per shard), as they did until round 5.
 * cmf_get_counter "allreduce_calls" / "allgather_calls" count the collectives a handle has issued, "halo_in_allreduce" says
 * which form is in force.
 * On a group handle cmf_set_factors / cmf_get_factors / cmf_update_motifs / cmf_update_feature_maps /
 * cmf_compute_loss / cmf_iterate / cmf_fit run the whole sharded iteration including the collectives, so the
 * reference's `fit` loop (src/algs/alternating.jl:44-67: one update_motifs! and one update_feature_maps! per
 * iteration) drives all GPUs with the same two calls.
 *
 * cmf_create_multi: ONE process drives `ndev` shards (the rule constructor `MultUpdate(data, W, H)`,
 * src/algs/mult.jl:11-20, for a Julia task).  devices[r] is the HIP device of shard r; `data` is the whole N x T
 * matrix, H crosses the ABI as the whole K x T matrix.  transport: CMF_COMM_AUTO = RCCL (ncclCommInitAll, one
 * stream per device, collectives inside ncclGroupStart/End) when the devices are distinct, loopback when they
 * are all the same device (the shards then share that device and the collectives are plain kernels: middle-rank
 * shards on a one-GPU box); CMF_COMM_RCCL forces RCCL (a 1-device group then still sends its buffers through
 * RCCL); CMF_COMM_LOOPBACK forces loopback (devices must all be equal). */
#define CMF_COMM_AUTO 0
#define CMF_COMM_RCCL 1
#define CMF_COMM_LOOPBACK 2
#define CMF_COMM_LOOPBACK_STREAMS 3 /* loopback with a stream per shard: the collectives keep RCCL's stream semantics (start when
                                       every shard's stream has arrived, every shard's stream continues when done) through
                                       events, so a missing dependency between shards cannot hide behind a shared stream */
#define CMF_COMM_PEER 4 /* direct peer access instead of RCCL (opt-in; one-process groups only): hipDeviceEnablePeerAccess between
                           all devices, the all-reduce is ONE kernel per device that sums its 1/R slice from the R buffers (reads
                           over xGMI) and stores the result into all of them (writes over xGMI) -- 1/R of the payload per link and
                           direction where a ring moves (R-1)/R of it twice -- between two event fences of all streams; visibility by
                           kernel boundaries only.  Sums in rank order: bitwise the loopback transport's results.  `devices` may also
                           list ONE device ndev times (rehearsal of the protocol on a one-GPU box).  RCCL stays the default: this
                           transport has never run across distinct devices. */
/* Who enqueues: groups whose shards have a stream each (RCCL, peer, loopback-streams) start one ENQUEUE WORKER thread per
 * shard; the calling thread only posts a shard's share of a phase to its worker, which issues the kernels and that shard's
 * collective calls (RCCL's one-thread-per-device mode: no ncclGroupStart/End).  cmf_set_option(h, "enqueue_threads", 0) or
 * CMF_ENQUEUE_THREADS=0 in the environment: the calling thread enqueues all shards, collectives in grouped calls.  Either
 * way a handle is used by one host thread at a time, every call returns with its work handed to the streams, and results
 * are bitwise the same. */
int cmf_create_multi(cmf_handle *h, int ndev, const int *devices, int transport,
                     int64_t N, int64_t T, int64_t K, int64_t L, const double *data);

/* One process per shard (torchrun-style launchers): create the shard with cmf_create_shard, then attach the
 * communicator.  cmf_comm_unique_id writes the 128-byte ncclUniqueId (rank 0 calls it and hands the bytes to the
 * other ranks by any out-of-band means); cmf_comm_init_rccl is ncclCommInitRank on the handle's device.  Both
 * all-reduce the data norm; H crosses the ABI as the LOCAL K x T_local block. */
int cmf_comm_unique_id(void *id128);
int cmf_comm_init_rccl(cmf_handle h, int nranks, int rank, const void *id128);
/* The overlap form (option "allreduce_overlap") runs its bulk all-reduce on a second stream while collectives of the main
 * stream are issued: that stream gets a communicator of its OWN, so that no two collectives in flight ever share one.
 * One process per shard: every rank calls this once with a SECOND id (rank 0's cmf_comm_unique_id, handed over like the
 * first) before switching the option on -- without it the option is refused (CMF_ERR_STATE).  cmf_create_multi groups
 * create their second set of communicators themselves (a second ncclCommInitAll) when the option is switched on.
 * No reference counterpart: the reference has no distributed code. */
int cmf_comm_init_overlap(cmf_handle h, const void *id128);
/* Host-collective transport (MPI / gloo style, used by the multi-process tests on one GPU): the library stages
 * each buffer through pinned host memory and calls back.  allreduce sums `count` floats in place over all ranks;
 * allgather fills recv (nranks * count floats, rank order) from every rank's `count`-float send block.  A non-zero
 * return aborts the entry with CMF_ERR_COMM. */
typedef int (*cmf_allreduce_fn)(void *user, float *host_buf, int64_t count);
typedef int (*cmf_allgather_fn)(void *user, const float *host_send, float *host_recv, int64_t count);
int cmf_comm_init_callbacks(cmf_handle h, int nranks, int rank, cmf_allreduce_fn allreduce, cmf_allgather_fn allgather, void *user);
/* "transport=rccl version=2.x.y lib=/path/librccl.so nranks=8 local=1 ranks=3" (truncated to len). */
int cmf_comm_info(cmf_handle h, char *buf, int64_t len);
/* Column block [t0, t1) of shard `rank` of a group handle (of the handle itself when it is not a group). */
int cmf_shard_bounds(cmf_handle h, int rank, int64_t *t0, int64_t *t1);

int cmf_destroy(cmf_handle h);

/* Block until everything this handle has enqueued -- on every stream of every local shard of a group -- has finished. */
int cmf_synchronize(cmf_handle h);
/* RCCL as this process would bind it (dlopen at first use, a copy the process has already mapped wins): its version
 * code (e.g. 22703) and, when path != NULL, the file it was loaded from.  CMF_ERR_COMM when no RCCL can be loaded. */
int cmf_rccl_version(int *version, char *path, int64_t path_len);
/* Event counters of a handle.  "hals_pipeline_reruns": H sweeps whose persistent pipeline ran out of a bounded wait and
 * were redone from the snapshot on the stage pipeline (cmf_hals_update_feature_maps).  "liveness_checks" (process-wide): stream queries the
 * waits for a loss have made -- 0 in a healthy run whose losses arrive within 50 ms (a query per iteration cost the device a marker
 * packet behind every loss conv until round 6).  "small_k_fused_h_updates": H updates of
 * the MU rule that ran inside the few-component contraction launch (option "small_k_fuse").  Group handles: "enqueue_ns" /
 * "enqueue_iters" = nanoseconds the calling thread spent enqueueing (or posting to the enqueue workers) the pipelined
 * iterations of cmf_iterate, and how many iterations that covers; "worker_ns" = time the busiest enqueue worker spent
 * inside its jobs (0 without workers). */
int cmf_get_counter(cmf_handle h, const char *name, int64_t *value);

/* Run all work of this handle on an existing HIP stream (hipStream_t passed
 * as void*), e.g. torch's current stream.  NULL is the HIP null (legacy
 * default) stream -- which is what torch's default "current stream" is.
 * A new handle runs on a private non-blocking stream until this is called.  A change of stream first waits for what the handle
 * has enqueued on the stream it leaves (a rule call may return with the next call's contraction in flight: option "speculate").
 * Group handles refuse it (CMF_ERR_STATE): they run on their own per-shard streams, and that includes the handle
 * cmf_create returns for a recording longer than one handle addresses (T > 8.3 M columns at K <= 64, see cmf_create) --
 * a caller that orders work through torch's current stream must cmf_synchronize such a handle instead. */
int cmf_set_stream(cmf_handle h, void *hip_stream);

/* Options (name, value):
 *   "reuse_est" (default 1): the est = tensor_conv(W, H) that closes update_feature_maps!
 *       (mult.jl:55) is kept, and the next update_motifs! (mult.jl:28), which would recompute the
 *       same est from the same W and H, reuses it (6 instead of 7 contractions per iteration,
 *       identical results).  0 = recompute it like the reference does.
 *   "speculate" (default 1): when the caller alternates update_motifs! / update_feature_maps! (alternating.jl:51-54), the latter
 *       enqueues the contraction the next update_motifs! starts with (numerator and denominator of mult.jl:31-34 need H, data and
 *       est only; l1W, l2W enter afterwards) behind its loss conv, so that the device works while the loss travels to the host and
 *       the caller's loop comes round; update_motifs! then only applies the update.  Identical results; anything that changes W, H,
 *       est or an option in between discards the work.  Single-GPU handles, with "reuse_est": the MU rule, and the HALS rule
 *       (hals.jl:13-38: the contraction of the residual with H_unfold and the lag correlations of H).  A caller that stops
 *       after an update_feature_maps! pays one contraction nobody reads (cmf_get_counter "speculated_contractions" counts the hits).
 *   "gram" (default 0): 1 = form denomW and denomH through Gram matrices (denomW = (H_unfold H_unfold') W,
 *       denomH = lag-Gram taps of W applied to H) instead of through est: exact rewritings of mult.jl:33,48 that
 *       execute 2.3 instead of 6 contractions plus the loss conv; results differ at rounding level only.
 *       2 = additionally take the loss from <H, denomH> - 2<H, numH> + ||data||^2 (no conv at all; the fp32
 *       cancellation limits its relative accuracy to about 1e-6 / loss^2; measured <= 5e-7 on the test shapes).  gram = 1 is
 *       also available on group handles: the all-reduce then carries [numW | HH | tail] (HH = H_unfold H_unfold', every
 *       shard's additive share) instead of [numW | denomW | tail]; it needs T >= 4 L.  gram = 2: unsharded handles only.
 *   "conv_kernel" (default 0 = chosen per launch; 2 = 128 x 128 workgroup tiles for the launches that only store or only
 *       sum the loss -- the epilogues that read data and store always take the one-wave kernel; 3 = one-wave 64 x 64 tiles) and
 *   "conv_split" (default 1: the tiles at the end of the one-wave kernel's grid -- a thin last round, and from four
 *       rounds on three more tiles per CU -- are cut into 32 x 32 quarter or 16 x 16 sixteenth tiles; 4 = quarter tiles
 *       only; 0 = whole tiles only): kernel selection of tensor_conv, for measurements.
 *   "small_k" (default 1 where it applies: K <= 16 and L <= 64): the few-component kernels (csrc/cmf_small_k.h) -- the three
 *       contractions of mult.jl:28-34,44-55 with the FLATTENED (lag, component) index on the MFMA axes instead of K padded to 32:
 *       the shapes the reference publishes on (README.md K = 5; figures/fast_bcd/synthetic_comparison.jl:58-64) run 2-3 times
 *       faster.  tensor_transconv keeps the general kernel when T is too short to fill the chip with its GEMM form; 2 = the
 *       few-component form whatever T is; 0 = the general kernels for every K.  Same arithmetic, another summation order.
 *   "small_k_fuse" (default 1): with the few-component kernels the element-wise update of H (mult.jl:51-52) runs inside the
 *       launch of the contraction before it -- the workgroup that completes a 128-column block's partial sums updates the
 *       block -- instead of in a launch of its own, where that launch is several rounds of workgroups long; bit for bit the
 *       same H (0 = the separate launch always; 2 = the fused form also on short launches, where it is slower; tests compare).
 *   "hals_prepare": allocate the HALS rule's scratch and check its shape limits now (see the HALS entries).
 *   "hals_persist" (default 1): how the H sweep of the HALS rule runs.  1 = as ONE persistent launch where its grid fits the chip
 *       (K sweepers + 4 (K-1) pullers, one workgroup per CU), 0 = one launch per pipeline stage, n > 1 = the persistent launch with at
 *       most n puller workgroups per row.  "hals_seg" (default 384; rounded up to a multiple of 64, at least 256) and "hals_lag"
 *       (default 2; 3 = the unshifted schedule) shape the stage pipeline; "hals_general" (default 0; bit 0 / bit 1) forces the general
 *       W / H sweeps at shapes the on-chip sweeps cover.  All four give the reference's visiting order and agree to rounding (the
 *       tests compare them); they exist for tests and measurements.  "hals_debug" (needs CMF_TEST_HOOKS=1; results are WRONG by
 *       design): 3 = the pullers of the persistent launch leave at once, so that every bounded wait runs out (the test of that
 *       path); 1 / 2 = no gating / pullers skip their work (timing).
 *   "hals_chase" (default -1 = a share estimated from the shape, 64 at config 5; 0 = off; 1..100 = that share): per cent of the tile rows of the residual conv behind the H sweep (hals.jl:41: the residual
 *       and the loss) that CHASE the persistent row pipeline instead of waiting for it: the pipeline (K + 4 (K-1) workgroups, VALU
 *       only) runs on a stream masked to the CUs it needs, those tile rows on a stream masked to the others, each tile waiting for
 *       the last row's progress flag; the rest of the conv follows on the whole chip (the pipeline then runs with three pullers per row instead of four: +1 % of its
 *       span for 32 more CUs on the conv's side).  The same arithmetic per tile (the two launches cut
 *       other tiles of their grids' tails into pieces: sums agree to rounding; run to run the results are bit for bit the same).
 *       K a multiple of 32, T >= 4096.
 *   "hals_gram" (default 2): where the HALS sweeps' projections come from.  2 = P of the H phase as denomH - numH of the MU
 *       quantities (one conv launch less; H within the residual form's accuracy), G of the W phase contracted from the
 *       stored residual; 0 = both from the residual; 1 = both as differences (no residual at all, but about 20x the
 *       rounding error in W).  ACCURACY: 0 and 2 meet the north star's 1e-4 (fp32 against the fp64 reference arithmetic)
 *       on W, H and loss_hist; hals_gram = 1 does NOT -- it is held to 3e-4 in the tests (measured: W 4-8e-5, H up to
 *       1.4e-4) and exists for measurements only.
 *   "profile" (n): bracket every n-th contraction launch with HIP events (cmf_kernel_times); 0 stops.
 *   "profile_mask" (bits): the kernel classes "profile" times -- bit i = the i-th name of cmf_kernel_times ("conv", "conv_t",
 *       "conv_loss", "conv_loss_store", "hxt", "transconv", ...); 0 (default) = all.  (An event pair idles the device a few
 *       microseconds: bench.py times only the dominant kernel inside its timed steps.)
 *   "halo_in_allreduce" (group handles, default 1): 0 = the halo of H in an all-gather of its own behind every H update (the form of
 *       rounds 1-5) even where the all-reduce could carry it (see the group section above).
 *   "allreduce_overlap" (group handles, default 0): 1 = numW (which needs H only) is contracted and all-reduced on a
 *       second stream right after the H update, underneath the loss conv and the denominator contraction, so only
 *       the denomW half of the all-reduce stays exposed; costs a second C2 launch per iteration. */
int cmf_set_option(cmf_handle h, const char *name, int value);
/* The names cmf_set_option accepts, comma-separated (truncation is an error: CMF_ERR_ARG; 512 bytes are plenty). */
int cmf_option_names(char *buf, int64_t len);

/* sum(data.^2) (fp64) over the columns this handle owns -- over all shards on a group handle
 * (data_norm: src/algs/mult.jl:13). */
int cmf_get_data_sumsq(cmf_handle h, double *sumsq);

/* ---- factors -------------------------------------------------------------
 * W (K x N x L) and H (K x T_local) in / out.  `fit` deep-copies the
 * initial factors and the rule mutates them in place
 * (src/algs/alternating.jl:33-34, src/algs/mult.jl:37-38,51-52): here the
 * working copies live on the device between calls.  cmf_set_factors: one of W, H may be NULL once both have been set -- that
 * factor keeps its resident value (a caller that edited only H between two rule calls uploads only H).  cmf_get_factors: either
 * may be NULL (not downloaded). */
int cmf_set_factors(cmf_handle h, const double *W, const double *H);
int cmf_get_factors(cmf_handle h, double *W, double *H);
/* The reference's rules READ the W and H they are handed (src/algs/mult.jl:23,42; hals.jl:31,37; pgd.jl:158,180); here the working
 * copies are device-resident, so a binding that wants the reference's semantics must notice when the caller hands it other arrays,
 * or arrays it has edited, and upload them first.  cmf_fingerprint is the cheap test the bindings use (CMFHip.jl `sync_args!`,
 * host.py `_sync_args`): a 64-bit fingerprint of a Float64 array of n elements over every line_stride-th 64-byte line (8 elements),
 * the last line and n itself (line_stride 1 = every element; the bindings' default 64 reads one line per 4 KB: 360 KB of the 23 MB
 * of factors at config 2).  Host arithmetic only (no handle, no device); equal arrays give equal fingerprints on every machine. */
int cmf_fingerprint(const double *a, int64_t n, int64_t line_stride, uint64_t *fingerprint);

/* In-place semantics for a caller that drives the rule call by call (the reference's own `fit`, src/algs/alternating.jl:51-54,
 * hands the SAME W and H arrays to every call and the rules mutate them: src/algs/mult.jl:37-38,51-52; hals.jl:110,153;
 * pgd.jl:293).  cmf_arm_writeback(h, W, H), called after update_motifs! and immediately before update_feature_maps! of any
 * rule (cmf_update_feature_maps, cmf_hals_update_feature_maps, cmf_pgd_update_feature_maps), makes THAT call also write the
 * new factors into W (K x N x L) and H (K x T), bit for bit what cmf_get_factors returns, before it returns -- without the
 * 23 MB synchronous fp64 download at config 2: W travels (fp32, pinned memory, a copy stream) underneath the H phase's first
 * contraction, H underneath the loss conv, and helper threads widen to Float64 while the caller waits for the loss scalar.
 * Either pointer may be NULL (that factor is not written); both NULL disarms.  EXCEPTION to "host pointers are borrowed for
 * the duration of the call only": W and H are borrowed from this call until the next *_update_feature_maps call on the handle
 * returns, and are written only inside that call.  One arm serves one call; cmf_iterate and cmf_fit (which run H phases of their own, not the
 * caller's rule calls) drop an arm that is still standing at their entry and write nothing.  If a helper thread ever fails to return within
 * CMF_WAIT_TIMEOUT_S (a hung device), the call fails and the handle refuses further arms (CMF_ERR_STATE); cmf_get_factors still works.
 * On a group handle (MU rule in either formulation, PGD rule)
 * every shard copies its own column block of H on its own device the same way (shard 0 also W) and the handle's helpers widen
 * block after block.
 * CMF_WRITEBACK_THREADS (environment, default 4): the widening helpers.
 * cmf_get_counter: "writeback_calls", "writeback_overlapped" (calls served by the copy stream). */
int cmf_arm_writeback(cmf_handle h, double *W, double *H);

/* ---- the update rule -------------------------------------------------------
 * update_motifs!(rule::MultUpdate, data, W, H; l1W=0, l2W=0)
 *   src/algs/mult.jl:23-39  (called at src/algs/alternating.jl:52). */
int cmf_update_motifs(cmf_handle h, double l1W, double l2W);
/* update_feature_maps!(rule::MultUpdate, data, W, H; l1H=0, l2H=0) -> loss
 *   src/algs/mult.jl:42-58  (called at src/algs/alternating.jl:54).
 * Synchronises: *loss is valid on return. */
int cmf_update_feature_maps(cmf_handle h, double l1H, double l2H, double *loss);
/* compute_loss(data, W, H) = norm(tensor_conv(W,H) - data) / norm(data)
 *   src/common.jl:54-59 (loss_hist[1], src/algs/alternating.jl:37). */
int cmf_compute_loss(cmf_handle h, double *loss);

/* The whole loop of fit(::AlternatingOptimizer, ...) with W, H device-resident:
 *   src/algs/alternating.jl:16-71.  loss_hist/time_hist must hold max_itr+1
 * doubles; *n_hist receives the number of entries written (iterations + 1);
 * *converged_early is 1 when the loop stopped on `converged` (:63-66; the
 * host wrapper prints "Converged early." like the reference).
 * Valid on single-GPU and group handles (on a group with a finite max_time every rank follows rank 0's clock).
 * time_hist: with a stop test armed (check_convergence != 0 or a finite max_time) entry i is the reference's cumulative
 * wall-clock time around the two rule calls of iteration i (alternating.jl:49,57-58).  When neither test can fire the loop runs
 * as ONE pipelined cmf_iterate batch (the host never stalls the device between iterations) and entry i is the DEVICE time,
 * from the start of the batch, at which iteration i was complete -- a HIP timing event recorded behind that iteration's loss
 * conv (shard 0's stream on a group) -- i.e. the same quantity without the host in it.  (Batches of more than 8192 iterations
 * keep host times: the moment each loss reached the host, half an iteration late.) */
int cmf_fit(cmf_handle h, int64_t max_itr, double max_time,
            int check_convergence, int64_t patience, double tol, int eval_mode,
            double l1W, double l2W, double l1H, double l2H,
            double *loss_hist, double *time_hist, int64_t *n_hist, int *converged_early);

/* n_iter MU iterations back to back: exactly `update_motifs!; update_feature_maps!` (alternating.jl:51-54; eval_mode
 * skips the motif update like :51) n_iter times, losses[i] = the loss update_feature_maps! returned in iteration i.
 * The host does not stall the device between iterations: the loss of iteration i is read one iteration late from
 * pinned memory (on a group handle it travels in the tail of iteration i+1's all-reduce, so an iteration costs one
 * all-reduce -- which also carries the halos of H, see the group section -- and one host wait).  stamps (may be NULL):
 * seconds since entry at which each loss
 * became known to the host.  cmf_fit uses this loop when check_convergence is 0 and max_time is infinite. */
int cmf_iterate(cmf_handle h, int64_t n_iter, int eval_mode, double l1W, double l2W, double l1H, double l2H,
                double *losses, double *stamps);

/* ---- HALS rule (BASELINE config 5) ----------------------------------------------
 * update_motifs!(rule::HALSUpdate, data, W, H; l1W=0, l2W=0)            src/algs/hals.jl:31-34, 90-112
 * update_feature_maps!(rule::HALSUpdate, data, W, H; l1H=0, l2H=0) -> loss   src/algs/hals.jl:37-42, 121-154
 * on the same handle (the HALSUpdate constructor, hals.jl:18-28, needs nothing beyond cmf_create +
 * cmf_set_factors: the residual it carries is est - data, kept implicitly).  Same Gauss-Seidel visiting
 * order as the reference; clamp at 0 and "+ l2" regularisation as in hals.jl:110,153.
 * Unsharded handles only (the H sweep is sequential along T).  Any K, L the reference accepts runs: the fast on-chip
 * sweeps cover L <= 64 (H) and L * Kpad <= 2048, K * L <= 2048 (W; Kpad = K rounded up to 32); beyond them general sweeps
 * run the same recurrences in the same order with their state in LDS (slower; up to L * Kpad = 16384).
 * cmf_set_option(h, "hals_prepare", 1) allocates the rule's scratch at construction time instead of at the first update.
 * The H sweep normally runs as one persistent launch whose workgroups (a sweeper per row of H, puller workgroups that
 * apply the cross-row terms) wait for each other through flags in device memory.  It needs the device to itself for
 * those ~2 ms: every such wait is bounded (about two seconds), and if one runs out -- another process or stream holding
 * the CUs -- cmf_hals_update_feature_maps restores H from the snapshot taken at the start of the sweep, redoes the sweep
 * with one launch per pipeline stage (no co-residency needed; the handle keeps to that form afterwards) and returns
 * CMF_OK; cmf_get_counter(h, "hals_pipeline_reruns") counts these events.
 * cmf_set_option(h, "hals_persist", 0) selects the one-launch-per-stage pipeline from the start. */
int cmf_hals_update_motifs(cmf_handle h, double l1W, double l2W);
int cmf_hals_update_feature_maps(cmf_handle h, double l1H, double l2H, double *loss);

/* ---- PGD rule ------------------------------------------
 * update_motifs!(rule::PGDUpdate, data, W, H; loss_func=SquareLoss(), constrW, penaltiesW)      src/algs/pgd.jl:158-177
 * update_feature_maps!(rule::PGDUpdate, data, W, H; loss_func, constrH, penaltiesH) -> loss    src/algs/pgd.jl:180-202
 * with SquareLoss (pgd.jl:29-36) or AbsoluteLoss (pgd.jl:41-47; cmf_pgd_set_loss); pen_sq / pen_abs are the summed
 * weights of the SquarePenalty / AbsolutePenalty entries of the penalty list (pgd.jl:74-89); `constraint` selects
 * 0 = none, 1 = NonnegConstraint (max(eps, x), pgd.jl:92-96), 2 = UnitNormConstraint (every component k whose slice
 * W[k,:,:] / H[k,:] has norm > 1 is scaled to norm 1, pgd.jl:100-110).
 * The rule's state (stepW = stepH = 5, cur_loss = norm(data), step_incr 1.05, step_decr 0.70; pgd.jl:139-154)
 * lives in the handle and is (re)initialised by cmf_create and cmf_pgd_reset.  Valid on single-GPU handles and on group
 * handles (cmf_create_multi / cmf_comm_init_*): one all-reduce of the partial gradW per iteration; the squared norm of gradH,
 * the component norms of UnitNormConstraint and the loss are summed over the shards in rank order, the step-size state
 * machine is replicated (every rank takes the same accept / reject decisions). */
int cmf_pgd_reset(cmf_handle h);
/* MaskedLoss(SquareLoss(), mask)  src/algs/pgd.jl:58-70 (the loss_func of the reference's own test/test.jl:45):
 * gradient 2*(est - data) .* mask, loss norm(mask.*data - mask.*est)^2.  `mask` is N x T column-major like data
 * (borrowed for the call); NULL restores the plain SquareLoss.  Only the PGD entries read the mask.  On a group handle the
 * mask is cut along T like data: cmf_create_multi groups take the whole N x T mask, a cmf_create_shard handle its own
 * columns followed by the right lag halo (the layout of data_local). */
int cmf_set_mask(cmf_handle h, const double *mask);
/* loss_func of the PGD entries: 0 = SquareLoss (default), 1 = AbsoluteLoss (gradient sign(est - data), loss
 * norm(data - est, 1); pgd.jl:41-47).  Combines with cmf_set_mask as MaskedLoss(loss, mask).
 * ACCURACY: with AbsoluteLoss the factors are held to 3e-4 (Frobenius-relative against the fp64 reference arithmetic), not
 * the 1e-4 of every other rule and loss: the gradient is sign(est - data), and an entry of est within fp32 rounding of
 * data takes the other sign than in fp64 -- a discontinuity no fp32 path can follow.  loss_hist stays within 1e-4. */
int cmf_pgd_set_loss(cmf_handle h, int loss_kind);
int cmf_pgd_update_motifs(cmf_handle h, double pen_sq, double pen_abs, int constraint);
int cmf_pgd_update_feature_maps(cmf_handle h, double pen_sq, double pen_abs, int constraint, double *loss);
int cmf_pgd_get_steps(cmf_handle h, double *stepW, double *stepH);

/* converged(loss_hist, patience, tol): src/model.jl:91-107 (host arithmetic). */
int cmf_converged(const double *loss_hist, int64_t len, int64_t patience, double tol);

/* ---- stand-alone primitives ------------------------------------------------
 * tensor_conv(W, H) -> est (N x T): src/common.jl:17-34. */
int cmf_tensor_conv(int device, int64_t N, int64_t T, int64_t K, int64_t L,
                    const double *W, const double *H, double *est);
/* tensor_transconv(W, X) -> out (K x T): src/common.jl:62-81. */
int cmf_tensor_transconv(int device, int64_t N, int64_t T, int64_t K, int64_t L,
                         const double *W, const double *X, double *out);

/* ---- initialisation / synthetic inputs -------------------------------------
 * init_rand(data, L, K): src/model.jl:113-125, with a portable counter-based
 * RNG in place of Julia's MersenneTwister (seed: fit_cnmf's `seed` kwarg,
 * src/model.jl:64-67).  Writes W (K x N x L), H (K x T). */
int cmf_init_rand(int device, int64_t N, int64_t T, int64_t K, int64_t L, uint64_t seed,
                  const double *data, double *W, double *H);
/* gen_synthetic (README.md:14) following synthetic_sequences
 * (datasets/synthetic.jl:29-61).  Writes data (N x T) and, when non-NULL,
 * the ground-truth W (K x N x L) and H (K x T). */
int cmf_gen_synthetic(int device, int64_t N, int64_t T, int64_t K, int64_t L,
                      double alpha, double p_h, double sigma, double noise_scale, uint64_t seed,
                      double *data, double *W, double *H);

/* ---- measurement hooks (bench.py) ------------------------------------------
 * Times `reps` launches of one named hot kernel with HIP events on the
 * handle's stream and returns the average duration in milliseconds plus the
 * algorithmic flop count of one launch.  name: "conv", "conv_t", "conv_loss", "conv_loss_store", "hxt", "transconv".
 * On a group handle "allreduce" times the group's bulk exchange alone (the buffer an iteration all-reduces); it is a
 * collective -- every rank of the group makes the call -- and *flops receives the payload in bytes. */
int cmf_time_kernel(cmf_handle h, const char *name, int reps, double *avg_ms, double *flops);
/* In-loop timing: after cmf_set_option(h, "profile", 1) every contraction launch of the update / loss entries is
 * bracketed by a HIP event pair on the launch stream; cmf_kernel_times synchronises and returns the mean duration
 * and the number of launches recorded for one class: "conv" (mult.jl:28), "conv_t" (:44), "conv_loss" /
 * "conv_loss_store" (:55-57), "hxt" (:31-34; "hxt_num" / "hxt_den" for the one-source launches of the overlap and
 * Gram forms), "transconv" (:47-48).  Setting the option again restarts it; a value
 * n > 1 brackets only every n-th launch of each class (an event pair costs a few microseconds on the stream). */
int cmf_kernel_times(cmf_handle h, const char *name, double *avg_ms, int64_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* CMF_HIP_H */
