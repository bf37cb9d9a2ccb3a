#pragma once
// cmf_api.hip -- host side of libcmf_hip.so: the C ABI of include/cmf_hip.h over the gfx950
// kernels in cmf_kernels.h.  No CPU fallback exists: every compute entry needs a HIP device.
#pragma GCC visibility push(default) // (the library is built with -fvisibility=hidden: only the C ABI is exported)
#include "cmf_hip.h"
#pragma GCC visibility pop
#include "cmf_kernels.h"
#include "cmf_rng.h"
#include "cmf_writeback.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
inline thread_local std::string g_err; // (one per thread for the whole library: C++17 inline variable)

static int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

// The library's test hooks (include/cmf_hip.h lists them): integer variables of the environment that count ONLY together with
// CMF_TEST_HOOKS=1, so that a stray variable in a production environment changes nothing.  Read by the calling thread at public
// entries only (getenv is not safe against a concurrent setenv of the host program).
static bool test_hooks_on()
{
    const char *hooks = getenv("CMF_TEST_HOOKS");
    return hooks && atoi(hooks) == 1;
}
static long long test_hook(const char *name, long long dflt)
{
    if (!test_hooks_on()) return dflt;
    const char *e = getenv(name);
    return e ? atoll(e) : dflt;
}

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(CMF_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define KCHK(name)                                                                                \
    do {                                                                                          \
        hipError_t e_ = hipGetLastError();                                                        \
        if (e_ != hipSuccess)                                                                     \
            return fail(CMF_ERR_HIP, "launch of %s failed: %s", name, hipGetErrorString(e_));     \
    } while (0)
#define CMFTRY(expr)              \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != CMF_OK) return rc_; \
    } while (0)

static inline int64_t rup(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// ------------------------------------------------------------------------------------------
// roctx ranges (tracing)
// ------------------------------------------------------------------------------------------
// The phases of an iteration and every collective are bracketed with roctx ranges on the thread that enqueues them, so that
// a `rocprofv3 --marker-trace --kernel-trace` timeline of a multi-GPU run reads as "W phase | all-reduce | H phase | halo
// all-gather | loss conv" per shard.  The marker library is bound at run time and only when it is wanted: a copy the
// process has already mapped (the profiler preloads it) is used, CMF_ROCTX=1 loads it on request; otherwise a range is
// two predictable branches.
#include <dlfcn.h>
struct RoctxApi {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
};
static const RoctxApi &roctx_api()
{
    static const RoctxApi api = [] {
        RoctxApi a;
        const char *env = getenv("CMF_ROCTX");
        if (env && atoi(env) == 0 && *env) return a; // CMF_ROCTX=0: never
        void *dl = nullptr;
        for (const char *nm : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            dl = dlopen(nm, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
            if (dl) break;
        }
        if (!dl && env && atoi(env) == 1)
            for (const char *nm : {"librocprofiler-sdk-roctx.so.1", "libroctx64.so.4", "/opt/rocm/lib/librocprofiler-sdk-roctx.so.1", "/opt/rocm/lib/libroctx64.so.4"}) {
                dl = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
                if (dl) break;
            }
        if (!dl) return a;
        a.push = reinterpret_cast<int (*)(const char *)>(dlsym(dl, "roctxRangePushA"));
        a.pop = reinterpret_cast<int (*)()>(dlsym(dl, "roctxRangePop"));
        if (!a.push || !a.pop) a.push = nullptr, a.pop = nullptr;
        return a;
    }();
    return api;
}
struct RoctxRange {
    bool on;
    explicit RoctxRange(const char *name) : on(roctx_api().push != nullptr)
    {
        if (on) roctx_api().push(name);
    }
    RoctxRange(const char *fmt, int a) : on(roctx_api().push != nullptr)
    {
        if (!on) return;
        char buf[96];
        snprintf(buf, sizeof(buf), fmt, a);
        roctx_api().push(buf);
    }
    ~RoctxRange()
    {
        if (on) roctx_api().pop();
    }
};

// ------------------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------------------
struct cmf_handle_s {
    int device = 0;
    CmfDims d{};
    int64_t t_offset = 0, T_global = 0;
    int halo_r = 0;       // data / H right halo columns actually present (0 on the last shard)
    bool has_left = false; // a left neighbour exists (t_offset > 0)
    bool sharded = false;

    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    float *H = nullptr, *Ht = nullptr, *Wt = nullptr, *Wn = nullptr;
    float *X = nullptr, *XT = nullptr, *est = nullptr, *estT = nullptr;
    float *wslabs = nullptr; // [nchunks][2][L][K32][Np]
    float *numden = nullptr; // [2][L][K32][Np]   (summed; the all-reduce buffer)
    float *numden_own = nullptr;
    float *hslabs = nullptr; // [S][2][Tl][K32]
    float *halo[4] = {nullptr, nullptr, nullptr, nullptr};
    float *halo_own[4] = {nullptr, nullptr, nullptr, nullptr};
    double *partial = nullptr; // loss partials
    double *d_scalar = nullptr; // device double[4]
    double *d_scalar_own = nullptr;
    double *h_scalar = nullptr; // pinned host double[4]
    double *stage = nullptr;    // fp64 staging for layout conversion
    size_t stage_elems = 0;

    // launch plans
    int hxt_LP = 1, hxt_groups = 1, hxt_nchunks = 1, hxt_chunk_len = 2;
    int tc_LT = 4, tc_S = 1, tc_W = 4;   // fragment slabs and participating waves of the two-source transconv
    int tc_S1 = 1, tc_W1 = 4;            // ... when only one source is contracted
    int tc_S2 = 1, tc_W2 = 4;            // ... two sources with one more 128-column block in FRONT of the own columns (shards that carry the halo of H in
                                         // the W-phase all-reduce update the L-1 columns in front of their own themselves: cmf_groups.hip)
    int halo_ext = 0;                    // that form is possible on this handle: L-1 (a shard with a left neighbour, the one-wave conv tiles, L-1 <= 64), else 0
    bool left_data = false;              // the L-1 columns of data in front of the own block are resident (data' only: the transconv's operand)
    std::vector<int4> tc_tab_host[3];    // per-wave work tables of the variants ([0]: two sources, [1]: one, [2]: two sources + the block in front)
    int4 *tc_tab[3] = {nullptr, nullptr, nullptr};
    int hxt_nchunks1 = 1, hxt_chunk_len1 = 6; // time chunks when only one source is contracted
    int hxt_main = 0, hxt_main1 = 0;          // rows the C2 kernel contracts in the two forms; the < 6*LP rows behind them are added by the slab sum
    // few components (K <= 16): the contractions on the flattened (lag, component) index (cmf_small_k.h)
    bool small_k = false;                   // option "small_k" (default on where the shape allows it)
    bool small_k_ok = false;                // the shape allows it
    bool sk_tc_ok = false, sk_tc = false;   // ... and its C3 form (G GEMM + fold) has enough columns to fill the chip; sk_tc: in use
    int sk_J = 0, sk_JP = 0, sk_MG = 1, sk_MBW = 4, sk_chunk_len = 16, sk_ngroups = 1, sk_TG = 128;
    int sk_RV = 0;                          // C2: the last sk_RV rows j on the VALU instead of in a padded MFMA block (hxt_small_kernel)
    int sk3_MG = 1, sk3_MBW = 4, sk3_Kg = 1, sk3_JP = 128; // C3: whole components per row group (g_gemm_fold_small_kernel)
    int sk3_NS = 1, sk3_RPS = 1;            // short recordings: C3's reduction over n in sk3_NS pieces of sk3_RPS rounds of 8 rows (2 slabs per piece)
    int sk3_GR = 128, sk3_RV = 0;           // rows of a row group in Wj (32-row blocks); the last sk3_RV live rows on the VALU (sk3_MBW then counts the MFMA blocks only)
    float *sk_slabs = nullptr, *sk_Wj = nullptr;
    int *sk_cnt = nullptr;                  // ticket counters of the fused few-component launches (one per 128-column block of H; zero between launches)
    int64_t sk_fused_h = 0;                 // cmf_get_counter "small_k_fused_h_updates"
    int sk_fuse = 1;                        // option "small_k_fuse": 1 = the element-wise update of H runs inside the C3 launch (cmf_small_k.h)
    int64_t sk_wj_gen = -1;                 // est_gen at which sk_Wj was packed from the resident W (w_update_small_kernel, wj_pack_kernel): every writer of W
                                            // passes through set_est, so a stale operand cannot be taken for a fresh one (-1: never packed)
    int tc_S_full = 1, tc_S1_full = 1;      // fragment slabs of the general transconv kernel (tc_S / tc_S1 are 1 while small_k is on)
    int conv_gx = 1, conv_gy = 1, conv_gy_ext = 1;
    int conv_variant = 0;   // K % 32 == 0: 3 = one-wave workgroups (conv3_kernel), 2 = 128 x 128 tiles (conv2_kernel), 0 = per mode
    int conv_partials = 1;  // loss partials written by the last conv launch
    int conv_split = 1;     // option "conv_split": 0 = whole tiles only, 1 = quarter / sixteenth tiles for the thin last round of the
                            // one-wave conv kernel, 4 = quarter tiles only
    int n_cu = 256;

    // HALS scratch (allocated on first use)
    bool hals_ready = false;
    bool gram_ready = false;                // the scratch the Gram form and HALS share (gram_ensure)
    int hals_NpH = 0, hals_NpC = 0, hals_TPp = 0, hals_ne = 0, hals_t_edge0 = 0, hals_corr_R = 4;
    float *hals_HX = nullptr, *hals_cslabs = nullptr, *hals_C = nullptr, *hals_HH = nullptr, *hals_PT = nullptr, *hals_D = nullptr;
    float *hals_PW = nullptr, *hals_GW = nullptr, *hals_GE = nullptr, *hals_GWt = nullptr;
    int hals_seg = 256, hals_nseg = 1;      // column segments of the pipelined H sweep
    int hals_gram = 2;                      // the sweeps' projections as differences of the MU quantities: 2 = P of the H phase only (default:
                                            // one conv launch fewer, H within the residual form's bars), 1 = G of the W phase too (~20x the rounding
                                            // error in W: opt-in), 0 = both contracted from the stored residual
    bool hals_w_general = false, hals_h_general = false; // shapes beyond the on-chip sweeps' limits: the general sweep kernels
    // options "hals_persist" (1 = the persistent pipeline where it fits, 0 = the stage pipeline, n > 1 = at most n pullers per row),
    // "hals_general" (bit 0 / 1: the general W / H sweeps at any shape), "hals_seg" / "hals_lag" (the stage pipeline's segment length
    // and schedule), "hals_debug" (CMF_TEST_HOOKS=1 only: 3 = the pullers leave at once, so that every bounded wait must run out)
    int hals_opt_persist = 1, hals_opt_general = 0, hals_opt_seg = 384, hals_opt_lag = 2, hals_debug = 0;
    // The residual conv CHASING the row pipeline (option "hals_chase" = per cent of its tile rows, 0 = off): the pipeline runs on a
    // stream masked to the CUs its K + (K-1)P workgroups need, the first tile rows of the conv on a stream masked to the other CUs,
    // each tile waiting for the last row's progress flag (conv3_chase_kernel); the rest of the conv follows on the whole chip.
    int hals_opt_chase = -1;             // (-1: the share is estimated from the shape, hals_chase_rows)
    hipStream_t hals_sA = nullptr, hals_sB = nullptr; // CU-masked: pipeline | chasing conv (created at the first chased sweep)
    hipEvent_t hals_ev[3] = {nullptr, nullptr, nullptr}; // fork, pipeline done, chasing part done
    int hals_mask_aper = 0;                 // CUs per XCD the pipeline's stream is masked to (the streams are remade when the plan changes)
    int hals_cuB = 0;                       // CUs of the chasing stream (the launch plans its tail pieces for them)
    int hals_chased_rows = 0, hals_chased_partials = 0; // tile rows / loss partials the chasing launch of the sweep in flight covers (0: none)
    int hals_pullers = 0;                   // persistent H pipeline: puller workgroups per row (0 = stage pipeline)
    int *hals_flags = nullptr;              // its progress flags (device)
    int *hals_status = nullptr;             // pinned host word: 1 = a wait of the persistent pipeline ran out
    float *hals_snap = nullptr;             // [2][TP][K32]: H and Ht as they were when the persistent sweep started
    double hals_l1 = 0.0, hals_l2 = 0.0;    // regularisers of the sweep in flight (for a rerun)
    int64_t hals_reruns = 0;                // H sweeps redone on the stage pipeline after such an expiry (cmf_get_counter)

    // PGD rule state (pgd.jl:139-154)
    double pgd_stepW = 5.0, pgd_stepH = 5.0, pgd_cur_loss = -1.0;
    float *pgd_gradH = nullptr;
    int pgd_loss_abs_now = 0;  // loss kind of the residual conv being launched (set by resid_and_loss / the PGD H phase)
    int pgd_loss_abs = 0;      // 0 SquareLoss (pgd.jl:29-36), 1 AbsoluteLoss (pgd.jl:41-47)
    double *pgd_knorm = nullptr; // [K32] per-component sums of squares of UnitNormConstraint (pgd.jl:100-110)
    float *M = nullptr, *MT = nullptr; // mask of MaskedLoss (pgd.jl:58-70) in the layouts of X and XT; null = no mask

    double data_sumsq = 0.0, data_norm = 0.0;
    bool factors_set = false;
    bool have_data = false;
    bool reuse_est = true;  // option "reuse_est"
    int gram = 0;           // option "gram": 0 off, 1 Gram-form denominators, 2 also the loss from Gram sums
    float *gram_numden_h = nullptr; // [1][2][Tl][K32]: numH | denomH in the h_update slab layout
    // in-loop kernel timing (option "profile"): HIP event pairs around the contraction launches, on the launch stream
    bool prof = false;
    int prof_every = 1;          // bracket every n-th launch of a class (option value n)
    int prof_seen[32] = {0};
    struct ProfRec { hipEvent_t a, b; int cls; };
    unsigned prof_mask = 0; // option "profile_mask"
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> prof_pool;
    int est_kind = 0;       // what est[t][n] holds for the resident W, H: 0 nothing, 1 tensor_conv(W,H), 2 tensor_conv(W,H) - data, 3 mask .* (tensor_conv(W,H) - data),
                            // 4 sign(tensor_conv(W,H) - data), 5 mask .* sign(...)  (the AbsoluteLoss gradient)
    void *arena = nullptr;  // the small buffers of the handle as ONE device allocation (cmf_create): 21 hipFree calls cost 1.3 ms, one 0.16
    size_t arena_bytes = 0;
    bool streams_may_hang = false;  // set on the shards of a FAILED group: their streams are not waited for when they are given back
    int64_t est_gen = 0;    // counts the assignments of est_kind (set_est): whatever changes H, W or est passes through one
    int64_t spec_gen = -1;  // est_gen for which the C2 contraction of the NEXT update_motifs! has already been enqueued (w_speculate); -1: none
    int last_rule_call = 0; // 1: cmf_update_motifs, 2: cmf_update_feature_maps (MU rule, single handle): speculation follows the alternation only
    bool speculate = true;  // option "speculate"
    std::function<int()> after_reduce;  // run once by the next reduce_partials between its launch and its wait for the sum (the HALS rule's speculation)
    int64_t hals_spec_gen = -1; // est_gen for which G (the C2 contraction on the residual) and HH of the NEXT HALS update_motifs! have been enqueued; -1: none
    int64_t spec_hits = 0;  // update_motifs! calls that found their contraction done (cmf_get_counter "speculated_contractions")

    // T-sharded groups (cmf_group.h): the handle the caller holds fronts a group when `group` is set
    struct cmf_group_s *group = nullptr;
    bool root_only = false;               // cmf_create_multi's front handle: no device state of its own
    hipStream_t own_comm_stream = nullptr, comm_stream = nullptr; // overlap form: the numW all-reduce runs here
    hipEvent_t ev_c0 = nullptr, ev_c1 = nullptr;
    // pipelined loss read-back of cmf_iterate (single handle): two pinned slots + events
    double *h_ring = nullptr;
    bool dev_stamps = false;              // set by cmf_fit around its pipelined batch: time_hist from HIP timing events on the stream
    CmfLossCarry carry{};                 // a loss reduction waiting for the next W phase's slab sum (cmf_iterate only)
    CmfWriteback *wb = nullptr;           // cmf_arm_writeback: the factors written into the caller's arrays behind a rule call
};

#define HALS_PMAX 4 // puller workgroups per row of the persistent H pipeline (4 -> 7 measured the same span: profiles/r04_hals_pullers_sweep.txt)

// cmf_group.h -- T-sharded groups behind the C ABI (included by cmf_api.hip).
//
// A *group* is R contiguous column blocks ("shards") of one problem: data / est / H are partitioned along T, W is
// replicated.  The caller holds ONE handle and makes the reference's two calls per iteration
// (update_motifs! / update_feature_maps!, alternating.jl:52,54); the library runs the sharded iteration:
//
//   update_motifs!        per shard: est = conv(W,H) (reused), [numW | denomW] partial sums        (mult.jl:28-34)
//                         ONE all-reduce(sum) of [numW | denomW | tail]   <- the only bulk exchange; the tail carries
//                                                                            the previous loss scalar of every rank
//                         per shard: the identical W update                                        (mult.jl:37-38)
//   update_feature_maps!  per shard: est (new W) on own columns + right lag halo, numH, denomH, H  (mult.jl:44-52)
//                         ONE all-gather of every shard's [first | last] L-1 columns of H (2 x 2.4 KB per shard)
//                         per shard: loss conv, sum((est - data)^2) -> the tail of the next all-reduce (mult.jl:55-57)
//
// Two ways to form a group:
//   cmf_create_multi       one process drives all shards (ndev devices, per-device streams; RCCL communicators from
//                          ncclCommInitAll, collectives inside ncclGroupStart/End) -- what a Julia caller of `fit` gets;
//   cmf_create_shard + cmf_comm_init_rccl / cmf_comm_init_callbacks
//                          one process per shard (torchrun-style launchers; bench.py --gpus N).
// Transports: RCCL over xGMI (the product path), "loopback" (all shards of a cmf_create_multi group on ONE device:
// the collectives are plain kernels -- exercises middle-rank shards on a one-GPU box), and host callbacks (the
// library stages the buffers through pinned host memory and the host performs the collective, e.g. gloo in the tests).
//
// Who enqueues (one-process groups with a stream per shard -- RCCL, peer, loopback-streams): by default one ENQUEUE WORKER
// thread per shard, bound to the shard's device, takes the shard's whole share of an iteration -- its kernels and its
// collective calls -- from a small queue, so the calling thread only posts (and polls the pinned loss words): eight shards
// are enqueued in parallel instead of one after the other.  Each worker calls ncclAllReduce / ncclAllGather on ITS
// communicator from ITS thread, without ncclGroupStart/End: that is RCCL's one-thread-per-device mode, in which no thread
// ever manages two devices -- chosen over "workers meet at a barrier, the caller issues one grouped call" because the
// barrier would put a host round trip back into every collective (two per iteration), which is exactly what the workers
// are there to remove.  Option "enqueue_threads" = 0 (or CMF_ENQUEUE_THREADS=0) restores the single-thread form with
// grouped RCCL calls.  The phases are written once, as lists of per-shard segments and collectives (GroupStep), and run
// either way.
#pragma once
#include <dlfcn.h>
#include <memory>
#include <mutex>
#include "cmf_workers.h"

// The handful of RCCL types this file passes through function pointers, declared here (values as in rccl.h of ROCm 7:
// the NCCL ABI these have had since NCCL 2.0) so that the library builds -- and loads -- on hosts without the RCCL
// development package; RCCL itself is bound with dlopen below.
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;      // non-zero codes are only ever turned into text by ncclGetErrorString
typedef enum { ncclFloat32 = 7 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
}

// ---- RCCL, bound at run time --------------------------------------------------------------------------------------
// librccl is opened with dlopen the first time a communicator is needed: the library then loads on hosts without
// RCCL, and when the process already holds an RCCL (e.g. PyTorch's bundled copy, same SONAME) that copy is reused
// instead of a second one being mapped.
struct RcclApi {
    void *dl = nullptr;
    std::string path;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr; // optional
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr; // optional
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
};
inline RcclApi g_rccl; // (one table for the whole library: C++17 inline variable)

#define RCCLCHK(expr)                                                                                          \
    do {                                                                                                       \
        ncclResult_t r_ = (expr);                                                                              \
        if (r_ != ncclSuccess) return fail(CMF_ERR_COMM, "%s failed: %s", #expr, g_rccl.GetErrorString(r_));   \
    } while (0)

// ---- the group ----------------------------------------------------------------------------------------------------
enum { CMF_TR_LOOPBACK = 0, CMF_TR_RCCL = 1, CMF_TR_CALLBACKS = 2, CMF_TR_PEER = 3 };
enum { CMF_ERR_ECHO = -1000 }; // internal: a worker that gave up because ANOTHER shard's job failed (never crosses the ABI)

// ---- enqueue workers ----------------------------------------------------------------------------------------------
// One thread per local shard of a one-process group (see the file comment); the queue, the meeting point and the abort
// protocol live in cmf_workers.h (free of HIP: stress-tested under ThreadSanitizer on the CPU).
struct cmf_group_s {
    int nranks = 1;
    int transport = CMF_TR_LOOPBACK;
    bool one_process = false;            // cmf_create_multi: all shards are local, H crosses the ABI as the global K x T matrix
    std::vector<cmf_handle_s *> sh;      // local shards
    std::vector<int> rank;               // global rank of each local shard
    std::vector<ncclComm_t> comm;        // RCCL communicators (one per local shard)
    std::vector<ncclComm_t> comm2;       // ... of the communication stream ("lane 1": the overlap form's bulk all-reduce never shares a
                                         // communicator with a collective of the main stream); created when the overlap form is switched on
    cmf_allreduce_fn ar_cb = nullptr;    // host-callback transport
    cmf_allgather_fn ag_cb = nullptr;
    void *cb_user = nullptr;
    float *cb_host = nullptr;            // pinned staging of the callback transport
    size_t cb_host_elems = 0;
    // device buffers, one per local shard
    std::vector<float *> red;            // [LKN2 + tail]: numW | denomW | loss tail  (the shard's numden points here)
    std::vector<float *> halo_send;      // [2 * HC]: own first | last L-1 columns of H
    std::vector<float *> halo_all;       // [nranks * 2 * HC]: every rank's send block (the shard's receive halos point inside)
    std::vector<float *> loss_all;       // [tail]: gathered (hi, lo) loss pairs of the synchronous path
    float *h_tail = nullptr;             // pinned host: 2 ring slots (the late loss read-back: pairs + stamp) + 1 for the synchronous one
    int64_t slot_len = 0;                // floats per slot
    int64_t LKN2 = 0, tail = 0, HC = 0;
    int64_t HHsz = 0;                    // floats of HH = H_unfold H_unfold' ((L*Kpad) x its 128-padded pitch): the Gram form's share
    int gram = 0;                        // option "gram" on a group: the all-reduce carries [numW | HH | tail] instead of [numW | denomW | tail]
    int64_t N = 0, T = 0, K = 0, L = 0;
    std::vector<int64_t> t0, t1;         // column block of every rank
    double data_sumsq = 0.0, data_norm = 0.0;
    bool overlap = false;                // option "allreduce_overlap": numW contracted + all-reduced under the loss conv
    bool num_ready = false;              // overlap form: the numW half belongs to the current H and is reduced (or in flight)
    bool halos_current = false;
    // The halo of H in the tail of the W-phase all-reduce (option "halo_in_allreduce", default on where every shard can: K a multiple of
    // 32, 1 <= L-1 <= 64, the L-1 columns of data in front of every shard resident; the default formulation): no collective between the
    // H update and the loss conv -- ONE collective per iteration.  Every shard with a left neighbour updates the L-1 columns in front
    // of its own itself (h_update_impl, front = true), from an H that is valid 2(L-1) columns out on the left; its own outer columns
    // (last 2(L-1), first L-1) travel in per-rank slots behind the loss tail of the NEXT all-reduce.
    bool halo_opt = true, halo_can = false;
    bool halos_pending = false;          // the H phase has run in that form: the L-1 columns in front are locally valid, everything else of
                                         // the halos is stale until the next all-reduce has brought the slots (halos_current is false meanwhile)
    bool halo_wide = false;              // the left halo is valid 2(L-1) columns out (set by the wide exchanges only)
    int64_t halo_len = 0;                // floats of the slots: nranks * 3 * HC
    std::vector<float *> halo3_send, halo3_all; // the same exchange as an all-gather (set-up, and whenever no all-reduce follows an H phase)
    int64_t n_allreduce = 0, n_allgather = 0;   // collectives built into step lists (cmf_get_counter "allreduce_calls" / "allgather_calls")
    // loopback with one stream PER SHARD (CMF_COMM_LOOPBACK_STREAMS): the collectives keep RCCL's stream semantics -- the
    // operation starts when every shard's stream has reached it and every shard's stream continues when it is done --
    // through events, so a missing dependency between shards cannot hide behind a shared stream (tests on a one-GPU box)
    std::vector<float *> gbuf;           // scratch of group_gather_doubles (one per local shard)
    size_t gbuf_words = 0;
    bool loop_ms = false;
    hipEvent_t ev_in[2][CMF_MAX_LOCAL] = {};  // [main | comm stream][shard]
    hipEvent_t ev_out[2] = {nullptr, nullptr};
    // peer transport: events of the stream fences around its kernels, [lane][before | after the kernel][shard]
    hipEvent_t ev_peer[2][2][CMF_MAX_LOCAL] = {};
    // enqueue workers (empty: the calling thread enqueues every shard itself)
    CmfWorkerPool pool;
    int test_fail_shard = -1;             // test hook CMF_TEST_FAIL_SHARD (group_check_ready)
    int64_t enqueue_ns = 0, enqueue_iters = 0; // cmf_iterate: time the calling thread spent enqueueing / posting, and the iterations it covers
    int force_inline = 0;                // > 0: step lists run on the calling thread although workers exist (GroupInline)
    bool failed = false;                 // a wait for the group ran out (or a collective reported an error): streams and communicators
                                         // may never drain -- destruction aborts the communicators and does not wait for the streams
};


// ---- prototypes of the functions the three translation units share (cmf_api.hip, cmf_rules.hip, cmf_groups.hip) ----
int hals_ensure(cmf_handle_s *h);
void hals_plan(cmf_handle_s *h);
static inline void set_est(cmf_handle_s *h, int kind) // every change of what est holds (and with it: of H, W) passes through here
{
    h->est_kind = kind;
    ++h->est_gen;
}
int wb_after_H(cmf_handle_s *h); // hook: the kernels that make H final have been enqueued (cmf_writeback.h)
int gram_ensure(cmf_handle_s *h);
int hals_w_impl(cmf_handle_s *h, double l1W, double l2W);
int hals_w_speculate(cmf_handle_s *h);
int resid_and_loss(cmf_handle_s *h, double *sumsq, bool masked = false, bool loss_abs = false);
int hals_resid_and_loss(cmf_handle_s *h, double *sumsq);
int gram_w_impl(cmf_handle_s *h, double l1W, double l2W);
int gram_w_partial(cmf_handle_s *h, float *hh_out);
int gram_w_finish(cmf_handle_s *h, const float *HH, double l1W, double l2W, const float *tail_src = nullptr, float *tail_dst = nullptr,
                         int tail_n = 0);
int gram_h_update(cmf_handle_s *h, double l1H, double l2H);
int gram_h_impl(cmf_handle_s *h, double l1H, double l2H, double *loss);
int pgd_w_impl(cmf_handle_s *h, double pen_sq, double pen_abs, int nonneg);
int pgd_h_impl(cmf_handle_s *h, double pen_sq, double pen_abs, int nonneg, double *loss);
int hals_h_impl(cmf_handle_s *h, double l1H, double l2H);
int hals_h_rerun(cmf_handle_s *h);
int gram_denom_h(cmf_handle_s *h, float *out);
int gram_tables(cmf_handle_s *h);
int group_pgd_w(cmf_handle_s *st, struct cmf_group_s *g, double pen_sq, double pen_abs, int nonneg);
int group_pgd_h(cmf_handle_s *st, struct cmf_group_s *g, double pen_sq, double pen_abs, int nonneg, double *loss);
int group_set_mask(struct cmf_group_s *g, const double *mask);
size_t n_partial(const cmf_handle_s *h);
hipError_t stream_acquire(int device, hipStream_t *s);
void plan(cmf_handle_s *h, int n_cu);
int upload_cols(cmf_handle_s *h, const double *src, int64_t tc, int64_t ncols, bool rows_layout, bool accumulate_sumsq,
                       float *rows_dst = nullptr, float *cols_dst = nullptr);
int create_impl(cmf_handle *out, int device, int64_t N, int64_t Tl, int64_t K, int64_t L,
                       const double *data, int64_t t_offset, int64_t T_global, bool sharded);
int launch_hxt_on(cmf_handle_s *h, const float *X0, const float *X1, int NpX, int nsrc, float *slabs, int nchunks, int chunk_len,
                         int main_rows = -1);
int launch_transconv(cmf_handle_s *h, int nsrc, const float *xt0 = nullptr, bool front_block = false);
int launch_slab_sum(cmf_handle_s *h, float *out, const float *in, int nslabs, size_t stride, bool take_carry = false,
                           CmfHxtTail tail = CmfHxtTail{nullptr, nullptr, nullptr, 0, 0, 0, 0, 0});
int hxt_contract(cmf_handle_s *h, const float *X0, const float *X1, int nsrc, float *out, bool take_carry = false, bool slabs_only = false);
int hxt_contract_small(cmf_handle_s *h, const float *X0, const float *X1, int nsrc, float *out, bool take_carry, bool slabs_only); // cmf_small.hip
bool sk_can_fuse_h(const cmf_handle_s *h); // cmf_small.hip
int launch_transconv_small(cmf_handle_s *h, int nsrc, const float *xt0, bool update_h = false, float l1 = 0.f, float two_l2 = 0.f);                                                          // cmf_small.hip
int read_scalar(cmf_handle_s *h, int slot, double *v);
int w_partial_impl(cmf_handle_s *h);
int w_partial_half_impl(cmf_handle_s *h, int den);
int w_apply_impl(cmf_handle_s *h, double l1W, double l2W, const float *tail_src = nullptr, float *tail_dst = nullptr, int tail_n = 0,
                        const float *den = nullptr);
int h_update_impl(cmf_handle_s *h, double l1H, double l2H, bool front = false);
int launch_loss_conv(cmf_handle_s *h);
int loss_partial_impl(cmf_handle_s *h, double *sumsq, bool readback = true, double *host_out = nullptr, bool speculate = false);
int set_factors_impl(cmf_handle_s *h, const double *W, const double *H);
int get_factors_impl(cmf_handle_s *h, double *W, double *H);
int reduce_partials(cmf_handle_s *h, const double *partial, int n, int slot, double *v);
double wait_timeout_s();
int ensure_resid(cmf_handle_s *h, bool masked = false, bool loss_abs = false);
HalsRowParams hals_row_params(cmf_handle_s *h, double l1H, double l2H);
int hals_persist_launch(cmf_handle_s *h, const HalsRowParams &q, int debug = 0, bool clear_flags = true);
int pgd_unit_norm(cmf_handle_s *h, bool is_W);
int rccl_load();
int group_use(cmf_handle_s *s);
int group_join(cmf_group_s *g);
int group_start_workers(cmf_group_s *g);
int group_allreduce(cmf_group_s *g, const std::vector<float *> &bufs, size_t off, size_t count, int lane = 0);
int group_allgather(cmf_group_s *g, const std::vector<float *> &send, const std::vector<float *> &recv, size_t count);
int group_gather_doubles(cmf_group_s *g, const std::vector<double> &vals, std::vector<double> &out, int n = 1);
int group_check_ready(cmf_group_s *g);
int group_sync(cmf_group_s *g);
int group_update_motifs(cmf_group_s *g, double l1W, double l2W);
int group_update_feature_maps(cmf_group_s *g, double l1H, double l2H, double *sumsq);
int group_compute_loss(cmf_group_s *g, double *loss);
int group_iterate(cmf_group_s *g, int64_t n, int eval_mode, double l1W, double l2W, double l1H, double l2H,
                         double *losses, double *stamps);
int group_set_factors(cmf_group_s *g, const double *W, const double *H);
int group_get_factors(cmf_group_s *g, double *W, double *H);
bool group_destroy(cmf_group_s *g);
int group_ensure_lane1(cmf_group_s *g, const void *id128 = nullptr);

// small helpers
static inline size_t group_tail_off(const cmf_group_s *g) { return (size_t)(g->gram ? g->LKN2 / 2 + g->HHsz : g->LKN2); }
static size_t group_stop_workers(cmf_group_s *g) { return cmf_pool_stop(g->pool, g->failed); }

// The entries that are not on the MU hot path (PGD, masks, stand-alone timings, ...) enqueue from the calling thread as they
// always did: inside this scope the workers are idle and every step list runs in line.
struct GroupInline {
    cmf_group_s *g;
    int rc;
    explicit GroupInline(cmf_group_s *g_) : g(g_), rc(group_join(g_)) { ++g->force_inline; }
    ~GroupInline() { --g->force_inline; }
};

template <typename T>
static int dalloc_zero(T **p, size_t n)
{
    HIPCHK(hipMalloc(p, n * sizeof(T)));
    HIPCHK(hipMemset(*p, 0, n * sizeof(T)));
    // hipMemset of device memory runs on the null stream and may return before it has finished; the handle's work runs on
    // non-blocking streams, which the null stream does not order -- a lagging fill would wipe what they wrote meanwhile
    HIPCHK(hipStreamSynchronize(nullptr));
    return CMF_OK;
}

// the C2 kernel adds CG consecutive time chunks inside a workgroup: it writes nchunks / CG slabs
static int hxt_cg(int nchunks) { return nchunks % 4 == 0 ? 4 : (nchunks % 2 == 0 ? 2 : 1); }
static int hxt_nslabs(int nchunks) { return nchunks / hxt_cg(nchunks); }

static const int kHxtLP[] = {1, 2, 3, 4, 5, 6, 8}; // 2*LP*16 accumulator registers must fit the 256 AGPRs

// kernel classes of the "profile" option
enum { PROF_CONV = 0, PROF_CONV_T, PROF_CONV_LOSS, PROF_CONV_LOSS_STORE, PROF_HXT, PROF_TRANSCONV, PROF_HXT_NUM, PROF_HXT_DEN, PROF_OTHER,
       PROF_HALS_PIPE, PROF_HALS_WSWEEP,
       // the contraction launches of the HALS / PGD / Gram rules (one source, or on the stored residual)
       PROF_CONV_RESID, PROF_HXT_RESID, PROF_HXT_HH, PROF_TRANSCONV_1, PROF_GRAM_DENOM_H, PROF_GRAM_TABLES, PROF_GRAM_W, PROF_NCLS };
static const char *kProfNames[PROF_NCLS] = {"conv", "conv_t", "conv_loss", "conv_loss_store", "hxt", "transconv", "hxt_num", "hxt_den", "other",
                                            "hals_h_pipeline", "hals_w_sweep",
                                            "conv_resid", "hxt_resid", "hxt_hh", "transconv_1src", "gram_denom_h", "gram_tables", "gram_w"};
static_assert(PROF_NCLS <= 32, "cmf_handle_s::prof_seen holds 32 classes");

struct ProfScope {
    cmf_handle_s *h;
    hipEvent_t b = nullptr;
    ProfScope(cmf_handle_s *h_, int cls) : h(h_)
    {
        if (!h->prof || h->prof_recs.size() >= 8192) return;
        if (h->prof_mask && !((h->prof_mask >> cls) & 1u)) return;
        if ((h->prof_seen[cls]++ % h->prof_every) != 0) return;
        hipEvent_t ev[2] = {nullptr, nullptr};
        for (int q = 0; q < 2; ++q) {
            if (!h->prof_pool.empty()) { ev[q] = h->prof_pool.back(); h->prof_pool.pop_back(); }
            else if (hipEventCreate(&ev[q]) != hipSuccess) { if (q == 1) h->prof_pool.push_back(ev[0]); return; }
        }
        (void)hipEventRecord(ev[0], h->stream);
        h->prof_recs.push_back({ev[0], ev[1], cls});
        b = ev[1];
    }
    ~ProfScope() { if (b) (void)hipEventRecord(b, h->stream); }
};

template <int MODE>
static int launch_conv(cmf_handle_s *h, float *out, int T_store, int gy, const float *data = nullptr)
{
    ProfScope prof_(h, MODE == 0 ? PROF_CONV : MODE == 1 ? PROF_CONV_T : MODE == 2 ? PROF_CONV_LOSS : MODE == 3 ? PROF_CONV_LOSS_STORE : PROF_CONV_RESID);
    const CmfDims &d = h->d;
    ConvParams p;
    p.Ht = h->Ht; p.Wt = h->Wt; p.out = out; p.data = data ? data : h->X; p.partial = h->partial;
    p.mask = (MODE == 7) ? h->MT : h->M;
    p.Np = d.Np; p.TP = d.TP; p.PADL = d.PADL; p.K = d.K; p.KB = d.KB; p.L = d.L; p.T_store = T_store;
    p.N = d.N; // (n blocks that are all padding are skipped)
    p.loss_abs = (MODE >= 4) ? h->pgd_loss_abs_now : 0;
    dim3 grid(h->conv_gx, gy), block(256);
    // measured at config 2 (tools/time_kernels.py): the one-wave kernel wins for the epilogues that read data
    // (0.924 vs 0.931 ms loss only, 0.936 vs 0.943 ms loss + store), the 128 x 128 tiles for the store-only ones
    // (0.911 vs 0.932 ms est, 0.910 vs 0.924 ms est')
    constexpr bool reads_data = (MODE >= 2); // every mode but the two plain stores loads a data (and mask) tile in its epilogue
    // The one-wave kernel can cut the tiles of its thin last round into quarter tiles (conv3_kernel).  That pays when
    // the remainder is small against the 12 wave slots per CU -- short shards: 3136 tiles on 3072 slots at T/8 -- and
    // then decides the variant for every mode; a remainder above 3 tiles per CU is left as whole tiles.
    const int gx3 = d.Np / 64, tiles3 = gx3 * ((T_store + 63) / 64), slots3 = 12 * h->n_cu;
    const int rem3 = tiles3 % slots3;
    // The tiles at the end of the grid are cut into one-wave pieces: the remainder of the last round when it is thin
    // (at most 3 tiles per CU), and, from `split_min_rounds` rounds on, `split_extra` more -- after several rounds of
    // dynamic dispatch the waves of a SIMD are out of step and the launch ends in a ragged drain one tile long; small
    // pieces at the end of the queue fill it (measured at config 2, 8.1 rounds: -1.5 % in every mode).
    // Below 4 rounds the extra cut costs more than it fills (T/4 and T/8 shards: +1-4 %).
    const int split_min_rounds = 4, split_extra = 3 * h->n_cu; // (profiles/r02*_conv_split*: the sweep these came from)
    int cut = 0;
    if (h->conv_split) {
        if (rem3 > 0 && rem3 <= 3 * h->n_cu) cut = rem3;
        if (tiles3 / slots3 >= split_min_rounds) cut += split_extra;
        cut = std::min(cut, tiles3);
    }
    {
        if (h->small_k) { // few components: one-wave tiles over the ceil(K/2) live k pairs per lag (conv_small_kernel)
            const int nkp = (d.K + 1) / 2;
            // the tiles beyond whole rounds of one tile per SIMD slot-triple (3 per SIMD) go out as quarter pieces at the end of the
            // grid, when they are few (at most one tile per SIMD: otherwise whole tiles balance well enough)
            const int per_round = 4 * h->n_cu;                          // one tile per SIMD
            const int remq = tiles3 % per_round;
            int cutq = (h->conv_split && tiles3 >= per_round && remq > 0 && remq <= per_round / 4) ? remq : 0;
            if (h->conv_split && tiles3 < per_round) cutq = tiles3; // fewer tiles than SIMDs (short recordings): quarter pieces only (configs[0]: 9-12 -> 5-8 us)
            const int n_full = tiles3 - cutq;
            grid = dim3(n_full + 4 * cutq);
            // (loss + store on a short launch: the data tile is requested before the MFMA loop, conv3_tile)
            const bool pre = MODE == 3 && nkp <= 4 && tiles3 <= 4 * per_round;
#define CASE(NKP_) do { if (pre) hipLaunchKernelGGL((conv_small_kernel<MODE, NKP_, (MODE == 3 && NKP_ <= 4)>), grid, dim3(64), 0, h->stream, p, gx3, n_full); \
                        else hipLaunchKernelGGL((conv_small_kernel<MODE, NKP_>), grid, dim3(64), 0, h->stream, p, gx3, n_full); } while (0)
            if (nkp <= 1) CASE(1); else if (nkp == 2) CASE(2); else if (nkp == 3) CASE(3); else if (nkp == 4) CASE(4);
            else if (nkp <= 6) CASE(6); else CASE(8);
#undef CASE
            h->conv_partials = (int)grid.x;
            KCHK("conv_small_kernel");
            return CMF_OK;
        }
    }
    const bool split = cut > 0;
    const int variant = (h->conv_variant && MODE <= 2) ? h->conv_variant : ((reads_data || split) ? 3 : 2);
    if (d.K % 32 == 0 && variant == 3) {
        const int n_full = tiles3 - cut;
        // quarter tiles reach every SIMD only from one tile per CU on; below that, sixteenth tiles
        const int pieces = (split && cut < h->n_cu && h->conv_split != 4) ? 16 : 4;
        grid = dim3(n_full + pieces * (tiles3 - n_full));
        hipLaunchKernelGGL((conv3_kernel<MODE>), grid, dim3(64), 0, h->stream, p, gx3, n_full, pieces);
    } else if (d.K % 32 == 0) {
        // the 128 x 128 kernel exists for the epilogues that only store or only sum (est, est', loss): with a data tile
        // read AND a store in the epilogue (mode 3 and the residual modes) it needs more than the 168 registers three
        // workgroups per CU leave (it spilled to scratch), and the one-wave kernel won those modes anyway
        if constexpr (MODE <= 2) hipLaunchKernelGGL((conv2_kernel<MODE>), grid, block, 0, h->stream, p);
    } else hipLaunchKernelGGL((conv_kernel<MODE, 0>), grid, block, 0, h->stream, p);
    h->conv_partials = (int)(grid.x * grid.y);
    KCHK("conv_kernel");
    return CMF_OK;
}

// The one-wave conv tiles of tile rows [row0, row0 + nrows) on h->stream (conv3_chase_kernel; K a multiple of 32): the tail of the
// grid is cut into pieces for a chip of n_cu CUs like launch_conv does.  gate != NULL: every tile waits for *gate >= its row + 1 (the
// HALS row pipeline's last progress flag) and reads H with agent-scope loads.  Loss partials pidx0 ... pidx0 + *npartials - 1.
template <int MODE>
static int launch_conv_rows(cmf_handle_s *h, float *out, int row0, int nrows, int pidx0, int n_cu, const int *gate, int *abort_word,
                            int *host_status, int *npartials, int T_store = -1)
{
    ProfScope prof_(h, MODE == 1 ? PROF_CONV_T : MODE == 3 ? PROF_CONV_LOSS_STORE : PROF_CONV_RESID);
    const CmfDims &d = h->d;
    ConvParams p;
    p.Ht = h->Ht; p.Wt = h->Wt; p.out = out; p.data = h->X; p.partial = h->partial; p.mask = h->M;
    p.Np = d.Np; p.TP = d.TP; p.PADL = d.PADL; p.K = d.K; p.KB = d.KB; p.L = d.L; p.T_store = T_store >= 0 ? T_store : d.Tl;
    p.N = d.N;
    p.loss_abs = 0;
    if (MODE == 5 || MODE == 7) p.data = h->XT;
    const int gx3 = d.Np / 64, tiles3 = gx3 * nrows, slots3 = 12 * n_cu;
    const int rem3 = tiles3 % slots3;
    int cut = 0;
    if (h->conv_split) {
        if (rem3 > 0 && rem3 <= 3 * n_cu) cut = rem3;
        if (tiles3 / slots3 >= 4) cut += 3 * n_cu;
        cut = std::min(cut, tiles3);
    }
    const int n_full = tiles3 - cut;
    const int pieces = (cut > 0 && cut < n_cu && h->conv_split != 4) ? 16 : 4;
    const int grid = n_full + pieces * cut;
    constexpr bool LOSS = (MODE == 2 || MODE == 3 || MODE == 4 || MODE == 6); // (the other modes write no per-tile sums)
    if (LOSS && (size_t)(pidx0 + grid) > n_partial(h)) return fail(CMF_ERR_STATE, "internal: loss partial buffer too small for a split conv");
    if (gate) hipLaunchKernelGGL((conv3_chase_kernel<MODE, true>), dim3(grid), dim3(64), 0, h->stream, p, gx3, n_full, pieces, row0, pidx0, gate, abort_word, host_status);
    else hipLaunchKernelGGL((conv3_chase_kernel<MODE, false>), dim3(grid), dim3(64), 0, h->stream, p, gx3, n_full, pieces, row0, pidx0, gate, abort_word, host_status);
    KCHK("conv3_chase_kernel");
    *npartials = grid;
    return CMF_OK;
}

inline std::atomic<int64_t> g_liveness_checks{0}; // stream queries / health checks made by waits for a loss (process-wide; cmf_get_counter "liveness_checks")

template <typename U>
static int wait_words(hipStream_t stream, const volatile U *p, int n, U sentinel, const std::function<int()> *health = nullptr,
                      const std::function<bool()> *enqueued = nullptr)
{
    auto all_there = [&]() {
        for (int j = 0; j < n; ++j)
            if (p[j] == sentinel) return false;
        return true;
    };
    // The liveness checks below are made after 50 ms of waiting and every 50 ms from then on, not every few thousand polls:
    // hipStreamQuery makes the runtime put a marker packet (a barrier with a system-scope release) into the stream when the last
    // command carries no signal, and the device paid for it -- 5.9 us between the loss conv and the next iteration's first
    // contraction, every iteration whose loss took longer than the first 4096 polls (profiles/r06_stream_query_gap.txt).
    const auto t_begin = std::chrono::steady_clock::now();
    auto next_check = t_begin + std::chrono::milliseconds(50);
    for (unsigned spins = 1;; ++spins) {
        if (all_there()) {
            std::atomic_thread_fence(std::memory_order_acquire);
            return CMF_OK;
        }
        if ((spins & 0x3FF) == 0 && std::chrono::steady_clock::now() >= next_check) {
            // (enqueue workers that are still posting leave the stream idle: only a stream that has been given all its work
            // and has drained it proves that the words will never come)
            g_liveness_checks.fetch_add(1, std::memory_order_relaxed);
            const hipError_t e = (enqueued && !(*enqueued)()) ? hipErrorNotReady : hipStreamQuery(stream);
            if (e == hipSuccess) { // everything enqueued has run: the words must be there now
                if (all_there()) return CMF_OK;
                return fail(CMF_ERR_HIP, "the stream drained without posting the loss");
            }
            if (e != hipErrorNotReady) return fail(CMF_ERR_HIP, "hipStreamQuery failed: %s", hipGetErrorString(e));
            if (health) CMFTRY((*health)());
            next_check = std::chrono::steady_clock::now() + std::chrono::milliseconds(50);
            if (std::chrono::duration<double>(next_check - t_begin).count() > wait_timeout_s())
                return fail(health ? CMF_ERR_COMM : CMF_ERR_HIP, "no loss arrived within %.0f s (CMF_WAIT_TIMEOUT_S): %s", wait_timeout_s(),
                            health ? "a collective of the group did not complete -- is every rank / device of the group still running?"
                                   : "the device did not finish the iteration");
        }
        __builtin_ia32_pause();
    }
}

// HIP timing events behind the iterations of a pipelined batch (cmf_fit's time_hist): begin() records the start, mark(it) goes
// behind iteration it's loss conv, finish() turns them into seconds since the start.  Inactive unless begin() was called.
struct DevStamps {
    std::vector<hipEvent_t> ev;
    bool active = false;
    int begin(hipStream_t st, int64_t n)
    {
        if (n < 1 || n > 8192) return CMF_OK; // (a batch that long keeps the host stamps)
        ev.assign((size_t)n + 1, nullptr);
        for (auto &e : ev) HIPCHK(hipEventCreate(&e));
        HIPCHK(hipEventRecord(ev[0], st));
        active = true;
        return CMF_OK;
    }
    int mark(hipStream_t st, int64_t it)
    {
        if (active) HIPCHK(hipEventRecord(ev[(size_t)it + 1], st));
        return CMF_OK;
    }
    int finish(double *stamps, int64_t n)
    {
        if (!active) return CMF_OK;
        HIPCHK(hipEventSynchronize(ev[(size_t)n]));
        for (int64_t it = 0; it < n; ++it) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, ev[0], ev[(size_t)it + 1]));
            stamps[it] = 1e-3 * (double)ms;
        }
        return CMF_OK;
    }
    ~DevStamps()
    {
        for (hipEvent_t e : ev)
            if (e) (void)hipEventDestroy(e);
    }
};

