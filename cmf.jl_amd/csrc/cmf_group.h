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
#pragma once
#include <dlfcn.h>
#include <mutex>

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
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr; // optional
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
};
static RcclApi g_rccl;

static int rccl_load()
{
    static std::mutex mu; // handles are per-thread, but this table is per process
    std::lock_guard<std::mutex> lock(mu);
    if (g_rccl.dl) return CMF_OK;
    const char *env = getenv("CMF_RCCL_LIB");
    if (env && std::strcmp(env, "none") == 0) // (tests: a rank without RCCL)
        return fail(CMF_ERR_COMM, "RCCL disabled by CMF_RCCL_LIB=none");
    const char *cands[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *dl = nullptr;
    for (const char *c : {"librccl.so.1", "librccl.so"}) { // a copy this process already mapped wins
        dl = dlopen(c, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        if (dl) break;
    }
    for (size_t i = 0; !dl && i < sizeof(cands) / sizeof(cands[0]); ++i)
        if (cands[i] && *cands[i]) dl = dlopen(cands[i], RTLD_NOW | RTLD_LOCAL);
    if (!dl) return fail(CMF_ERR_COMM, "RCCL not found (librccl.so.1; set CMF_RCCL_LIB): %s", dlerror());
#define RCCL_SYM(field, name)                                                                 \
    do {                                                                                      \
        g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(dl, name));             \
        if (!g_rccl.field) { dlclose(dl); return fail(CMF_ERR_COMM, "RCCL symbol %s missing", name); } \
    } while (0)
    RCCL_SYM(GetVersion, "ncclGetVersion");
    RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
    RCCL_SYM(CommInitRank, "ncclCommInitRank");
    RCCL_SYM(CommInitAll, "ncclCommInitAll");
    RCCL_SYM(CommDestroy, "ncclCommDestroy");
    RCCL_SYM(GetErrorString, "ncclGetErrorString");
    RCCL_SYM(AllReduce, "ncclAllReduce");
    RCCL_SYM(AllGather, "ncclAllGather");
    RCCL_SYM(GroupStart, "ncclGroupStart");
    RCCL_SYM(GroupEnd, "ncclGroupEnd");
#undef RCCL_SYM
    g_rccl.CommGetAsyncError = reinterpret_cast<decltype(g_rccl.CommGetAsyncError)>(dlsym(dl, "ncclCommGetAsyncError"));
    Dl_info info;
    if (dladdr(reinterpret_cast<void *>(g_rccl.AllReduce), &info) && info.dli_fname) g_rccl.path = info.dli_fname;
    g_rccl.dl = dl;
    return CMF_OK;
}

#define RCCLCHK(expr)                                                                                          \
    do {                                                                                                       \
        ncclResult_t r_ = (expr);                                                                              \
        if (r_ != ncclSuccess) return fail(CMF_ERR_COMM, "%s failed: %s", #expr, g_rccl.GetErrorString(r_));   \
    } while (0)

// ---- the group ----------------------------------------------------------------------------------------------------
enum { CMF_TR_LOOPBACK = 0, CMF_TR_RCCL = 1, CMF_TR_CALLBACKS = 2 };

struct cmf_group_s {
    int nranks = 1;
    int transport = CMF_TR_LOOPBACK;
    bool one_process = false;            // cmf_create_multi: all shards are local, H crosses the ABI as the global K x T matrix
    std::vector<cmf_handle_s *> sh;      // local shards
    std::vector<int> rank;               // global rank of each local shard
    std::vector<ncclComm_t> comm;        // RCCL communicators (one per local shard)
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
    // loopback with one stream PER SHARD (CMF_COMM_LOOPBACK_STREAMS): the collectives keep RCCL's stream semantics -- the
    // operation starts when every shard's stream has reached it and every shard's stream continues when it is done --
    // through events, so a missing dependency between shards cannot hide behind a shared stream (tests on a one-GPU box)
    std::vector<float *> gbuf;           // scratch of group_gather_doubles (one per local shard)
    size_t gbuf_words = 0;
    bool loop_ms = false;
    hipEvent_t ev_in[2][CMF_MAX_LOCAL] = {};  // [main | comm stream][shard]
    hipEvent_t ev_out[2] = {nullptr, nullptr};
};

static void group_partition(int64_t T, int R, int64_t L, std::vector<int64_t> &t0, std::vector<int64_t> &t1)
{
    const int64_t base = (T + R - 1) / R;
    t0.resize(R);
    t1.resize(R);
    for (int r = 0; r < R; ++r) {
        t0[r] = std::min<int64_t>((int64_t)r * base, T);
        t1[r] = std::min<int64_t>(t0[r] + base, T);
    }
    (void)L;
}

// Where the loss tail starts in the all-reduce buffer: behind [numW | denomW], or behind [numW | HH] in the Gram form --
// either way the buffer that travels is ONE contiguous range that ends with the tail.
static inline size_t group_tail_off(const cmf_group_s *g) { return (size_t)(g->gram ? g->LKN2 / 2 + g->HHsz : g->LKN2); }

static int group_use(cmf_handle_s *s)
{
    HIPCHK(hipSetDevice(s->device));
    return CMF_OK;
}

static int group_alloc_buffers(cmf_group_s *g)
{
    cmf_handle_s *s0 = g->sh[0];
    const CmfDims &d = s0->d;
    g->LKN2 = (int64_t)2 * d.L * d.K32 * d.Np;
    g->tail = rup(2 * g->nranks, 64);
    g->HC = (int64_t)std::max(1, d.L - 1) * d.K32;
    g->HHsz = (int64_t)d.L * d.K32 * rup((int64_t)d.L * d.K32, 128);
    const size_t red_elems = (size_t)std::max(g->LKN2, g->LKN2 / 2 + g->HHsz) + (size_t)g->tail;
    const size_t nl = g->sh.size();
    g->red.assign(nl, nullptr);
    g->halo_send.assign(nl, nullptr);
    g->halo_all.assign(nl, nullptr);
    g->loss_all.assign(nl, nullptr);
    for (size_t i = 0; i < nl; ++i) {
        cmf_handle_s *s = g->sh[i];
        CMFTRY(group_use(s));
        CMFTRY(dalloc_zero(&g->red[i], red_elems));
        CMFTRY(dalloc_zero(&g->halo_send[i], (size_t)(2 * g->HC)));
        CMFTRY(dalloc_zero(&g->halo_all[i], (size_t)(g->nranks * 2 * g->HC)));
        CMFTRY(dalloc_zero(&g->loss_all[i], (size_t)(2 * g->tail))); // [gathered pairs | send scratch]
        s->numden = g->red[i];
        const int r = g->rank[i];
        s->halo[0] = g->halo_send[i];
        s->halo[1] = g->halo_send[i] + g->HC;
        s->halo[2] = r > 0 ? g->halo_all[i] + (size_t)(2 * (r - 1) + 1) * g->HC : nullptr;          // left neighbour's send-to-right block
        s->halo[3] = r < g->nranks - 1 ? g->halo_all[i] + (size_t)(2 * (r + 1)) * g->HC : nullptr;   // right neighbour's send-to-left block
    }
    CMFTRY(group_use(s0));
    if (2 * g->nranks > 256) return fail(CMF_ERR_UNSUPPORTED, "groups of more than 128 shards are not supported");
    g->slot_len = g->tail;
    HIPCHK(hipHostMalloc(&g->h_tail, (size_t)(3 * g->slot_len) * sizeof(float), hipHostMallocCoherent)); // 2 ring slots + the synchronous read-back
    std::memset(g->h_tail, 0, (size_t)(3 * g->slot_len) * sizeof(float));
    return CMF_OK;
}

static int group_cb_stage(cmf_group_s *g, size_t elems)
{
    if (g->cb_host_elems >= elems) return CMF_OK;
    if (g->cb_host) (void)hipHostFree(g->cb_host);
    g->cb_host = nullptr;
    g->cb_host_elems = 0;
    HIPCHK(hipHostMalloc(&g->cb_host, elems * sizeof(float)));
    g->cb_host_elems = elems;
    return CMF_OK;
}

// loopback with a stream per shard: the collective kernel runs on shard 0's stream once every shard's stream has arrived
// (c = 0: the main streams, 1: the communication streams of the overlap form) ...
static int loopback_arrive(cmf_group_s *g, int c)
{
    for (size_t i = 0; i < g->sh.size(); ++i) {
        if (!g->ev_in[c][i]) HIPCHK(hipEventCreateWithFlags(&g->ev_in[c][i], hipEventDisableTiming));
        HIPCHK(hipEventRecord(g->ev_in[c][i], c ? g->sh[i]->comm_stream : g->sh[i]->stream));
    }
    hipStream_t s0 = c ? g->sh[0]->comm_stream : g->sh[0]->stream;
    for (size_t i = 1; i < g->sh.size(); ++i) HIPCHK(hipStreamWaitEvent(s0, g->ev_in[c][i], 0));
    return CMF_OK;
}
// ... and every other shard's stream goes on when it has finished
static int loopback_depart(cmf_group_s *g, int c)
{
    hipStream_t s0 = c ? g->sh[0]->comm_stream : g->sh[0]->stream;
    if (!g->ev_out[c]) HIPCHK(hipEventCreateWithFlags(&g->ev_out[c], hipEventDisableTiming));
    HIPCHK(hipEventRecord(g->ev_out[c], s0));
    for (size_t i = 1; i < g->sh.size(); ++i) HIPCHK(hipStreamWaitEvent(c ? g->sh[i]->comm_stream : g->sh[i]->stream, g->ev_out[c], 0));
    return CMF_OK;
}

// In-place sum over all ranks of `count` floats at offset `off` of every local shard's buffer `bufs[i]`, ordered on
// `streams[i]`.
static int group_allreduce(cmf_group_s *g, const std::vector<float *> &bufs, size_t off, size_t count, bool on_comm_stream = false)
{
    const size_t nl = g->sh.size();
    if (g->nranks == 1 && g->transport != CMF_TR_RCCL) return CMF_OK;
    auto stream_of = [&](size_t i) { return on_comm_stream ? g->sh[i]->comm_stream : g->sh[i]->stream; };
    switch (g->transport) {
    case CMF_TR_RCCL: {
        if (nl > 1) RCCLCHK(g_rccl.GroupStart());
        for (size_t i = 0; i < nl; ++i) {
            CMFTRY(group_use(g->sh[i]));
            RCCLCHK(g_rccl.AllReduce(bufs[i] + off, bufs[i] + off, count, ncclFloat32, ncclSum, g->comm[i], stream_of(i)));
        }
        if (nl > 1) RCCLCHK(g_rccl.GroupEnd());
        return CMF_OK;
    }
    case CMF_TR_LOOPBACK: { // all shards share one device (and, unless loop_ms, one stream)
        CmfPtrTable tab;
        for (size_t i = 0; i < nl; ++i) tab.p[i] = bufs[i] + off;
        CMFTRY(group_use(g->sh[0]));
        const int c = on_comm_stream ? 1 : 0;
        if (g->loop_ms) CMFTRY(loopback_arrive(g, c));
        const int blocks = (int)std::min<size_t>(2048, (count + 255) / 256);
        hipLaunchKernelGGL(loopback_allreduce_kernel, dim3(blocks), dim3(256), 0, stream_of(0), tab, (int)nl, count);
        KCHK("loopback_allreduce_kernel");
        if (g->loop_ms) CMFTRY(loopback_depart(g, c));
        return CMF_OK;
    }
    default: { // host callbacks: one local shard
        cmf_handle_s *s = g->sh[0];
        CMFTRY(group_use(s));
        CMFTRY(group_cb_stage(g, count));
        HIPCHK(hipMemcpyAsync(g->cb_host, bufs[0] + off, count * sizeof(float), hipMemcpyDeviceToHost, stream_of(0)));
        HIPCHK(hipStreamSynchronize(stream_of(0)));
        const int rc = g->ar_cb(g->cb_user, g->cb_host, (int64_t)count);
        if (rc != 0) return fail(CMF_ERR_COMM, "all-reduce callback returned %d", rc);
        HIPCHK(hipMemcpyAsync(bufs[0] + off, g->cb_host, count * sizeof(float), hipMemcpyHostToDevice, stream_of(0)));
        HIPCHK(hipStreamSynchronize(stream_of(0)));
        return CMF_OK;
    }
    }
}

// recv[i] (nranks * count floats) = every rank's send block (count floats), in rank order
static int group_allgather(cmf_group_s *g, const std::vector<float *> &send, size_t send_off, const std::vector<float *> &recv, size_t count)
{
    const size_t nl = g->sh.size();
    switch (g->transport) {
    case CMF_TR_RCCL: {
        if (nl > 1) RCCLCHK(g_rccl.GroupStart());
        for (size_t i = 0; i < nl; ++i) {
            CMFTRY(group_use(g->sh[i]));
            RCCLCHK(g_rccl.AllGather(send[i] + send_off, recv[i], count, ncclFloat32, g->comm[i], g->sh[i]->stream));
        }
        if (nl > 1) RCCLCHK(g_rccl.GroupEnd());
        return CMF_OK;
    }
    case CMF_TR_LOOPBACK: {
        CmfPtrTable ts, tr;
        for (size_t i = 0; i < nl; ++i) { ts.p[i] = send[i] + send_off; tr.p[i] = recv[i]; }
        CMFTRY(group_use(g->sh[0]));
        if (g->loop_ms) CMFTRY(loopback_arrive(g, 0));
        const int blocks = (int)std::min<size_t>(64, (nl * count + 255) / 256);
        hipLaunchKernelGGL(loopback_allgather_kernel, dim3(blocks), dim3(256), 0, g->sh[0]->stream, ts, tr, (int)nl, (int)count);
        KCHK("loopback_allgather_kernel");
        if (g->loop_ms) CMFTRY(loopback_depart(g, 0));
        return CMF_OK;
    }
    default: {
        cmf_handle_s *s = g->sh[0];
        CMFTRY(group_use(s));
        CMFTRY(group_cb_stage(g, (size_t)(g->nranks + 1) * count));
        float *hs = g->cb_host, *hr = g->cb_host + count;
        HIPCHK(hipMemcpyAsync(hs, send[0] + send_off, count * sizeof(float), hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        const int rc = g->ag_cb(g->cb_user, hs, hr, (int64_t)count);
        if (rc != 0) return fail(CMF_ERR_COMM, "all-gather callback returned %d", rc);
        HIPCHK(hipMemcpyAsync(recv[0], hr, (size_t)g->nranks * count * sizeof(float), hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        return CMF_OK;
    }
    }
}

// sum over ranks (in rank order) of the doubles posted as (hi, lo) float pairs
static double group_decode_tail(const cmf_group_s *g, const float *tail)
{
    double s = 0.0;
    for (int r = 0; r < g->nranks; ++r) s += (double)tail[2 * r] + (double)tail[2 * r + 1];
    return s;
}

// n doubles per rank -> all of them on every rank (exact: the 8 bytes of a double travel through the all-gather as two
// 32-bit words).  Every local shard contributes vals[i*n .. i*n+n); out[r*n + j] for all ranks r.  Synchronises.
static int group_gather_doubles(cmf_group_s *g, const std::vector<double> &vals, std::vector<double> &out, int n = 1)
{
    const size_t nl = g->sh.size();
    out.assign((size_t)g->nranks * n, 0.0);
    if (g->one_process) { // all ranks are local: no transport needed
        for (size_t i = 0; i < nl; ++i)
            for (int j = 0; j < n; ++j) out[(size_t)g->rank[i] * n + j] = vals[i * n + j];
        return CMF_OK;
    }
    const size_t words = (size_t)2 * n; // floats per rank
    if (g->gbuf.size() != nl || g->gbuf_words < words) { // [gathered (nranks * words) | send (words)] per local shard
        for (size_t i = 0; i < g->gbuf.size(); ++i)
            if (g->gbuf[i]) { (void)hipSetDevice(g->sh[i]->device); (void)hipFree(g->gbuf[i]); }
        g->gbuf.assign(nl, nullptr);
        g->gbuf_words = 0;
        for (size_t i = 0; i < nl; ++i) {
            CMFTRY(group_use(g->sh[i]));
            CMFTRY(dalloc_zero(&g->gbuf[i], (size_t)(g->nranks + 1) * words));
        }
        g->gbuf_words = words;
    }
    std::vector<float *> send(nl), recv(nl);
    for (size_t i = 0; i < nl; ++i) {
        cmf_handle_s *s = g->sh[i];
        CMFTRY(group_use(s));
        recv[i] = g->gbuf[i];
        send[i] = g->gbuf[i] + (size_t)g->nranks * g->gbuf_words;
        HIPCHK(hipMemcpyAsync(send[i], vals.data() + i * n, (size_t)n * 8, hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    CMFTRY(group_allgather(g, send, 0, recv, words));
    cmf_handle_s *s = g->sh[0];
    CMFTRY(group_use(s));
    HIPCHK(hipMemcpyAsync(out.data(), recv[0], (size_t)g->nranks * n * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return CMF_OK;
}

static int group_check_ready(cmf_group_s *g)
{
    for (cmf_handle_s *s : g->sh) {
        if (!s->factors_set) return fail(CMF_ERR_STATE, "factors not set: call cmf_set_factors first");
        if (!s->have_data) return fail(CMF_ERR_STATE, "handle was created without data");
    }
    return CMF_OK;
}

static int group_sync(cmf_group_s *g)
{
    for (cmf_handle_s *s : g->sh) {
        CMFTRY(group_use(s));
        HIPCHK(hipStreamSynchronize(s->stream));
        if (s->comm_stream) HIPCHK(hipStreamSynchronize(s->comm_stream));
    }
    return CMF_OK;
}

// Called while the host polls for a loss that only shard 0 posts: a fault on another local shard's stream, or an
// asynchronous RCCL error on any local communicator, must end the wait (CMF_ERR_HIP / CMF_ERR_COMM) instead of hanging it.
static int group_health(cmf_group_s *g)
{
    for (size_t i = 0; i < g->sh.size(); ++i) {
        cmf_handle_s *s = g->sh[i];
        if (i > 0 && s->stream != g->sh[0]->stream) {
            CMFTRY(group_use(s));
            const hipError_t e = hipStreamQuery(s->stream);
            if (e != hipSuccess && e != hipErrorNotReady)
                return fail(CMF_ERR_HIP, "shard %d (device %d) failed: %s", g->rank[i], s->device, hipGetErrorString(e));
        }
        if (g->transport == CMF_TR_RCCL && g_rccl.CommGetAsyncError && i < g->comm.size() && g->comm[i]) {
            ncclResult_t ae = ncclSuccess;
            if (g_rccl.CommGetAsyncError(g->comm[i], &ae) == ncclSuccess && ae != ncclSuccess && (int)ae != 7 /* ncclInProgress */)
                return fail(CMF_ERR_COMM, "RCCL reported an asynchronous error on rank %d: %s", g->rank[i], g_rccl.GetErrorString(ae));
        }
    }
    return group_use(g->sh[0]);
}

// (L-1)-column H halo exchange: pack -> one all-gather -> unpack
static int group_exchange_halos(cmf_group_s *g)
{
    cmf_handle_s *s0 = g->sh[0];
    const int rows = s0->d.L - 1;
    g->halos_current = true;
    if (rows < 1 || g->nranks == 1) return CMF_OK;
    for (size_t i = 0; i < g->sh.size(); ++i) {
        cmf_handle_s *s = g->sh[i];
        const CmfDims &d = s->d;
        CMFTRY(group_use(s));
        hipLaunchKernelGGL(halo_pack2_kernel, dim3(8), dim3(256), 0, s->stream, s->H, g->halo_send[i], d.PADL, d.PADL + d.Tl - rows, rows, d.K32);
        KCHK("halo_pack2_kernel");
    }
    CMFTRY(group_allgather(g, g->halo_send, 0, g->halo_all, (size_t)(2 * g->HC)));
    for (size_t i = 0; i < g->sh.size(); ++i) {
        cmf_handle_s *s = g->sh[i];
        const CmfDims &d = s->d;
        CMFTRY(group_use(s));
        hipLaunchKernelGGL(halo_unpack2_kernel, dim3(8), dim3(256), 0, s->stream, s->H, s->Ht, s->halo[2], s->halo[3],
                           d.PADL - rows, d.PADL + d.Tl, rows, d.K32, d.TP);
        KCHK("halo_unpack2_kernel");
    }
    return CMF_OK;
}

// sum((conv(W,H) - data)^2) of every local shard -> its tail slots of the all-reduce buffer (and d_scalar[0]).
// defer: the per-tile sums are reduced by the next update_motifs!' slab sum instead (CmfLossCarry), right in front of the
// all-reduce their total rides on.
static int group_loss_partials(cmf_group_s *g, bool defer = false)
{
    for (size_t i = 0; i < g->sh.size(); ++i) {
        cmf_handle_s *s = g->sh[i];
        CMFTRY(group_use(s));
        CMFTRY(launch_loss_conv(s)); // mult.jl:55-57
        if (defer) {
            s->carry = CmfLossCarry{s->partial, s->conv_partials, s->d_scalar, nullptr, g->red[i] + group_tail_off(g), (int)g->tail, g->rank[i]};
            continue;
        }
        hipLaunchKernelGGL(loss_tail_kernel, dim3(1), dim3(256), 0, s->stream, s->partial, s->conv_partials, s->d_scalar,
                           g->red[i] + group_tail_off(g), (int)g->tail, g->rank[i]);
        KCHK("loss_tail_kernel");
    }
    return CMF_OK;
}

// the tail of the all-reduce buffer right now (synchronous path): all-gather of the (hi, lo) pairs, read back
static int group_loss_now(cmf_group_s *g, double *sumsq)
{
    const size_t nl = g->sh.size();
    cmf_handle_s *s = g->sh[0];
    if (g->nranks == 1 && g->transport != CMF_TR_RCCL) return read_scalar(s, 0, sumsq);
    std::vector<float *> send(nl);
    for (size_t i = 0; i < nl; ++i) send[i] = g->red[i] + group_tail_off(g) + 2 * g->rank[i];
    CMFTRY(group_allgather(g, send, 0, g->loss_all, 2));
    CMFTRY(group_use(s));
    float *stage = g->h_tail + 2 * g->slot_len; // not a ring slot: a pending one-iteration-late loss may still sit there
    HIPCHK(hipMemcpyAsync(stage, g->loss_all[0], (size_t)(2 * g->nranks) * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    *sumsq = group_decode_tail(g, stage);
    return CMF_OK;
}

// overlap form: numW needs H only -- contract it and start its all-reduce on the communication stream.  In the Gram form
// the whole payload [numW | HH] needs H only, so ALL of the bulk all-reduce runs underneath the loss conv.
static int group_start_num(cmf_group_s *g)
{
    const size_t half = (size_t)g->LKN2 / 2;
    for (cmf_handle_s *s : g->sh) {
        CMFTRY(group_use(s));
        if (g->gram) CMFTRY(gram_w_partial(s, s->numden + half));
        else CMFTRY(w_partial_half_impl(s, 0));
        HIPCHK(hipEventRecord(s->ev_c0, s->stream));
        HIPCHK(hipStreamWaitEvent(s->comm_stream, s->ev_c0, 0));
    }
    CMFTRY(group_allreduce(g, g->red, 0, g->gram ? half + (size_t)g->HHsz : half, true));
    for (cmf_handle_s *s : g->sh) {
        CMFTRY(group_use(s));
        HIPCHK(hipEventRecord(s->ev_c1, s->comm_stream));
    }
    g->num_ready = true;
    return CMF_OK;
}

// update_motifs! on the group (mult.jl:23-39).  ring_slot >= 0: after the all-reduce the tail (the previous
// iteration's loss pairs of every rank) is copied to pinned host slot `ring_slot`, which the host polls.
static int group_update_motifs(cmf_group_s *g, double l1W, double l2W, int ring_slot = -1)
{
    if (!g->halos_current) CMFTRY(group_exchange_halos(g));
    const size_t half = (size_t)g->LKN2 / 2;
    const size_t toff = group_tail_off(g);
    if (g->overlap) {
        if (!g->num_ready) CMFTRY(group_start_num(g));
        if (g->gram) { // the bulk is in flight on the communication stream: only the loss tail is left for this stream
            CMFTRY(group_allreduce(g, g->red, toff, (size_t)g->tail));
        } else {
            for (cmf_handle_s *s : g->sh) {
                CMFTRY(group_use(s));
                CMFTRY(w_partial_half_impl(s, 1));
            }
            CMFTRY(group_allreduce(g, g->red, half, half + (size_t)g->tail));
        }
        for (cmf_handle_s *s : g->sh) {
            CMFTRY(group_use(s));
            HIPCHK(hipStreamWaitEvent(s->stream, s->ev_c1, 0));
        }
        g->num_ready = false;
    } else {
        for (cmf_handle_s *s : g->sh) {
            CMFTRY(group_use(s));
            if (g->gram) CMFTRY(gram_w_partial(s, s->numden + half)); // [numW | this shard's share of HH | tail]
            else CMFTRY(w_partial_impl(s));
        }
        CMFTRY(group_allreduce(g, g->red, 0, toff + (size_t)g->tail));
    }
    for (size_t i = 0; i < g->sh.size(); ++i) {
        cmf_handle_s *s = g->sh[i];
        CMFTRY(group_use(s));
        if (i == 0 && ring_slot >= 0) { // shard 0's W update also drops the reduced loss pairs + a stamp into the pinned ring slot
            float *ring = g->h_tail + (size_t)ring_slot * g->slot_len;
            for (int j = 0; j < 2 * g->nranks; ++j) reinterpret_cast<volatile unsigned *>(ring)[j] = CMF_SENTINEL32; // collected an iteration ago
            if (g->gram) CMFTRY(gram_w_finish(s, s->numden + half, l1W, l2W, g->red[0] + toff, ring, 2 * g->nranks));
            else CMFTRY(w_apply_impl(s, l1W, l2W, g->red[0] + toff, ring, 2 * g->nranks));
        } else {
            if (g->gram) CMFTRY(gram_w_finish(s, s->numden + half, l1W, l2W));
            else CMFTRY(w_apply_impl(s, l1W, l2W));
        }
    }
    return CMF_OK;
}

// update_feature_maps! on the group (mult.jl:42-58).  sumsq != NULL: also reduce the loss now (synchronises);
// NULL: the loss partials stay in the tail of the all-reduce buffer and ride on the next update_motifs!.
static int group_update_feature_maps(cmf_group_s *g, double l1H, double l2H, double *sumsq)
{
    if (!g->halos_current) CMFTRY(group_exchange_halos(g));
    for (cmf_handle_s *s : g->sh) {
        CMFTRY(group_use(s));
        CMFTRY(g->gram ? gram_h_update(s, l1H, l2H) : h_update_impl(s, l1H, l2H));
    }
    g->num_ready = false;
    CMFTRY(group_exchange_halos(g));
    if (g->overlap) CMFTRY(group_start_num(g)); // for the next update_motifs!: H and its halos are final now
    // (Gram + overlap: the next W phase has no slab sum left on this stream for a deferred reduction to ride on)
    CMFTRY(group_loss_partials(g, /*defer=*/sumsq == nullptr && !(g->gram && g->overlap)));
    return sumsq ? group_loss_now(g, sumsq) : CMF_OK;
}

static int group_compute_loss(cmf_group_s *g, double *loss)
{
    if (!g->halos_current) CMFTRY(group_exchange_halos(g));
    CMFTRY(group_loss_partials(g));
    double ss = 0.0;
    CMFTRY(group_loss_now(g, &ss));
    *loss = std::sqrt(ss) / g->data_norm;
    return CMF_OK;
}

// n MU iterations back to back (alternating.jl:51-54 n times).  The loss of iteration i travels in the tail of
// iteration i+1's all-reduce and is read from pinned memory after iteration i+1 has been enqueued, so the host never
// stalls the device between iterations; the last loss is flushed with the small all-gather.  stamps (optional):
// host seconds since entry at which each loss became known.
static int group_iterate(cmf_group_s *g, int64_t n, int eval_mode, double l1W, double l2W, double l1H, double l2H,
                         double *losses, double *stamps)
{
    const auto t_begin = std::chrono::steady_clock::now();
    auto now = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
    if (eval_mode) { // no W update, so no all-reduce to ride on: synchronous losses
        for (int64_t it = 0; it < n; ++it) {
            double ss = 0.0;
            CMFTRY(group_update_feature_maps(g, l1H, l2H, &ss));
            losses[it] = std::sqrt(ss) / g->data_norm;
            if (stamps) stamps[it] = now();
        }
        return CMF_OK;
    }
    cmf_handle_s *s0 = g->sh[0];
    const std::function<int()> health = [g]() { return group_health(g); };
    for (int64_t it = 0; it < n; ++it) {
        CMFTRY(group_update_motifs(g, l1W, l2W, it > 0 ? (int)((it - 1) & 1) : -1));
        const bool last = (it + 1 == n);
        double ss = 0.0;
        CMFTRY(group_update_feature_maps(g, l1H, l2H, last ? &ss : nullptr));
        if (it > 0) {
            const int slot = (int)((it - 1) & 1);
            CMFTRY(group_use(s0));
            const float *ring = g->h_tail + (size_t)slot * g->slot_len;
            CMFTRY(wait_words<unsigned>(s0->stream, reinterpret_cast<const volatile unsigned *>(ring), 2 * g->nranks, CMF_SENTINEL32, &health));
            losses[it - 1] = std::sqrt(group_decode_tail(g, ring)) / g->data_norm;
            if (stamps) stamps[it - 1] = now();
        }
        if (last) {
            losses[it] = std::sqrt(ss) / g->data_norm;
            if (stamps) stamps[it] = now();
        }
    }
    return CMF_OK;
}

static int group_set_factors(cmf_group_s *g, const double *W, const double *H)
{
    for (size_t i = 0; i < g->sh.size(); ++i) {
        cmf_handle_s *s = g->sh[i];
        // one process: H is the global K x T matrix (column-major: a shard's columns are contiguous)
        const double *Hs = g->one_process ? H + (size_t)g->t0[(size_t)g->rank[i]] * g->K : H;
        CMFTRY(set_factors_impl(s, W, Hs));
    }
    g->num_ready = false;
    g->halos_current = false;
    return group_exchange_halos(g);
}

static int group_get_factors(cmf_group_s *g, double *W, double *H)
{
    CMFTRY(group_sync(g));
    for (size_t i = 0; i < g->sh.size(); ++i) {
        cmf_handle_s *s = g->sh[i];
        double *Hs = (H && g->one_process) ? H + (size_t)g->t0[(size_t)g->rank[i]] * g->K : H;
        CMFTRY(get_factors_impl(s, i == 0 ? W : nullptr, Hs));
    }
    return CMF_OK;
}

static void group_destroy(cmf_group_s *g)
{
    if (!g) return;
    for (cmf_handle_s *s : g->sh) {
        (void)hipSetDevice(s->device);
        (void)hipStreamSynchronize(s->stream);
        if (s->comm_stream) (void)hipStreamSynchronize(s->comm_stream);
    }
    if (g->transport == CMF_TR_RCCL && g_rccl.dl)
        for (ncclComm_t c : g->comm)
            if (c) (void)g_rccl.CommDestroy(c);
    for (size_t i = 0; i < g->sh.size(); ++i) {
        cmf_handle_s *s = g->sh[i];
        (void)hipSetDevice(s->device);
        s->numden = s->numden_own;
        for (int w = 0; w < 4; ++w) s->halo[w] = s->halo_own[w];
        if (i < g->red.size() && g->red[i]) (void)hipFree(g->red[i]);
        if (i < g->halo_send.size() && g->halo_send[i]) (void)hipFree(g->halo_send[i]);
        if (i < g->halo_all.size() && g->halo_all[i]) (void)hipFree(g->halo_all[i]);
        if (i < g->loss_all.size() && g->loss_all[i]) (void)hipFree(g->loss_all[i]);
    }
    if (g->h_tail) (void)hipHostFree(g->h_tail);
    if (g->cb_host) (void)hipHostFree(g->cb_host);
    for (size_t i = 0; i < g->gbuf.size() && i < g->sh.size(); ++i)
        if (g->gbuf[i]) { (void)hipSetDevice(g->sh[i]->device); (void)hipFree(g->gbuf[i]); }
    for (int c = 0; c < 2; ++c) {
        for (hipEvent_t e : g->ev_in[c])
            if (e) (void)hipEventDestroy(e);
        if (g->ev_out[c]) (void)hipEventDestroy(g->ev_out[c]);
    }
    delete g;
}

// streams / events of the overlap form and common post-construction steps of a shard that joins a group
static int group_prepare_shard(cmf_handle_s *s)
{
    CMFTRY(group_use(s));
    if (!s->own_comm_stream) HIPCHK(hipStreamCreateWithFlags(&s->own_comm_stream, hipStreamNonBlocking));
    s->comm_stream = s->own_comm_stream;
    if (!s->ev_c0) HIPCHK(hipEventCreateWithFlags(&s->ev_c0, hipEventDisableTiming));
    if (!s->ev_c1) HIPCHK(hipEventCreateWithFlags(&s->ev_c1, hipEventDisableTiming));
    return CMF_OK;
}

static int group_finish_norm(cmf_group_s *g)
{
    std::vector<double> vals(g->sh.size()), all;
    for (size_t i = 0; i < g->sh.size(); ++i) vals[i] = g->sh[i]->data_sumsq;
    CMFTRY(group_gather_doubles(g, vals, all));
    g->data_sumsq = 0.0;
    for (double v : all) g->data_sumsq += v; // rank order: identical on every rank
    g->data_norm = std::sqrt(g->data_sumsq);  // mult.jl:13 over all shards
    for (cmf_handle_s *s : g->sh) s->data_norm = g->data_norm;
    return CMF_OK;
}
