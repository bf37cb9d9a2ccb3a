// cmf_groups.hip -- T-sharded groups behind the C ABI: transports, the sharded MU / PGD iteration, group construction.
#include "cmf_internal.h"

int rccl_load()
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
    g_rccl.CommAbort = reinterpret_cast<decltype(g_rccl.CommAbort)>(dlsym(dl, "ncclCommAbort"));
    Dl_info info;
    if (dladdr(reinterpret_cast<void *>(g_rccl.AllReduce), &info) && info.dli_fname) g_rccl.path = info.dli_fname;
    g_rccl.dl = dl;
    return CMF_OK;
}

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

int group_use(cmf_handle_s *s)
{
    HIPCHK(hipSetDevice(s->device));
    return CMF_OK;
}

// ---- worker pool glue ------------------------------------------------------------------------------------------------
// have the workers taken everything that was posted?  (the device may still be running it)
static bool group_enqueued(const cmf_group_s *g) { return cmf_pool_idle(g->pool); }

// Wait until every worker has run every posted job; the first failure (in shard order) becomes this thread's error.
// Bounded like every wait of the group (CMF_WAIT_TIMEOUT_S): a worker that does not come back from a collective call ends
// in CMF_ERR_COMM and a group marked failed, not in a hang.
int group_join(cmf_group_s *g)
{
    if (g->pool.empty()) return CMF_OK;
    if (!cmf_pool_wait(g->pool, wait_timeout_s())) {
        g->failed = true;
        return fail(CMF_ERR_COMM, "an enqueue worker of the group did not return within %.0f s (CMF_WAIT_TIMEOUT_S): a collective call is blocked on the host", wait_timeout_s());
    }
    std::string err;
    int rc = cmf_pool_collect(g->pool, CMF_ERR_ECHO, &err);
    if (rc != CMF_OK) {
        g_err = err;
        if (rc == CMF_ERR_ECHO) rc = CMF_ERR_STATE;
    }
    if (rc == CMF_ERR_COMM) g->failed = true;
    return rc;
}


// Meeting point of the workers inside a collective of the event-ordered transports (an event must have been recorded before
// another stream is told to wait for it).  Gives up when any job of the group has failed.
static int group_barrier(cmf_group_s *g)
{
    const int rc = cmf_pool_barrier(g->pool, CMF_ERR_ECHO);
    return rc == 0 ? CMF_OK : fail(CMF_ERR_ECHO, "another shard of the group failed");
}


static bool group_wants_workers(const cmf_group_s *g)
{
    // (tests: a worker for a ONE-shard RCCL group too, so that a one-GPU box can show a worker thread issuing the RCCL calls of a
    // communicator another thread created -- honoured only with CMF_TEST_HOOKS=1)
    if (g->one_process && g->sh.size() == 1 && g->transport == CMF_TR_RCCL)
        if (test_hook("CMF_TEST_FORCE_WORKERS", 0) == 1) return true;
    if (!g->one_process || g->sh.size() < 2) return false;
    return g->transport == CMF_TR_RCCL || g->transport == CMF_TR_PEER || (g->transport == CMF_TR_LOOPBACK && g->loop_ms);
}

int group_start_workers(cmf_group_s *g)
{
    if (!g->pool.empty() || !group_wants_workers(g)) return CMF_OK;
    std::vector<int> devs;
    for (cmf_handle_s *s : g->sh) devs.push_back(s->device);
    g->pool.on_start = [devs](size_t i) { (void)hipSetDevice(devs[i]); }; // (by value: the closure lives in state the workers co-own)
    g->pool.last_error = []() { return g_err; };
    cmf_pool_start(g->pool, g->sh.size());
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
    g->halo_len = (int64_t)g->nranks * 3 * g->HC;
    const size_t red_elems = (size_t)std::max(g->LKN2, g->LKN2 / 2 + g->HHsz) + (size_t)g->tail + (size_t)g->halo_len;
    const size_t nl = g->sh.size();
    g->red.assign(nl, nullptr);
    g->halo_send.assign(nl, nullptr);
    g->halo_all.assign(nl, nullptr);
    g->halo3_send.assign(nl, nullptr);
    g->halo3_all.assign(nl, nullptr);
    g->loss_all.assign(nl, nullptr);
    for (size_t i = 0; i < nl; ++i) {
        cmf_handle_s *s = g->sh[i];
        CMFTRY(group_use(s));
        CMFTRY(dalloc_zero(&g->red[i], red_elems));
        CMFTRY(dalloc_zero(&g->halo_send[i], (size_t)(2 * g->HC)));
        CMFTRY(dalloc_zero(&g->halo_all[i], (size_t)(g->nranks * 2 * g->HC)));
        CMFTRY(dalloc_zero(&g->halo3_send[i], (size_t)(3 * g->HC)));
        CMFTRY(dalloc_zero(&g->halo3_all[i], (size_t)g->halo_len));
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

static inline hipStream_t lane_stream(const cmf_group_s *g, size_t i, int lane) { return lane ? g->sh[i]->comm_stream : g->sh[i]->stream; }
static inline ncclComm_t lane_comm(const cmf_group_s *g, size_t i, int lane) { return lane ? g->comm2[i] : g->comm[i]; }

static int lane_event(hipEvent_t *e)
{
    if (!*e) HIPCHK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    return CMF_OK;
}

// loopback with a stream per shard: the collective kernel runs on shard 0's stream once every shard's stream has arrived
// (c = 0: the main streams, 1: the communication streams of the overlap form) ...
static int loopback_arrive(cmf_group_s *g, int c)
{
    for (size_t i = 0; i < g->sh.size(); ++i) {
        CMFTRY(lane_event(&g->ev_in[c][i]));
        HIPCHK(hipEventRecord(g->ev_in[c][i], lane_stream(g, i, c)));
    }
    hipStream_t s0 = lane_stream(g, 0, c);
    for (size_t i = 1; i < g->sh.size(); ++i) HIPCHK(hipStreamWaitEvent(s0, g->ev_in[c][i], 0));
    return CMF_OK;
}
// ... and every other shard's stream goes on when it has finished
static int loopback_depart(cmf_group_s *g, int c)
{
    hipStream_t s0 = lane_stream(g, 0, c);
    CMFTRY(lane_event(&g->ev_out[c]));
    HIPCHK(hipEventRecord(g->ev_out[c], s0));
    for (size_t i = 1; i < g->sh.size(); ++i) HIPCHK(hipStreamWaitEvent(lane_stream(g, i, c), g->ev_out[c], 0));
    return CMF_OK;
}

// peer transport: every shard's stream of the lane waits until every other shard's stream has reached this point
// (site 0: in front of the transport's kernel, 1: behind it).  The calling thread does it for all shards ...
static int peer_fence_all(cmf_group_s *g, int lane, int site)
{
    const size_t nl = g->sh.size();
    for (size_t i = 0; i < nl; ++i) {
        CMFTRY(group_use(g->sh[i]));
        CMFTRY(lane_event(&g->ev_peer[lane][site][i]));
        HIPCHK(hipEventRecord(g->ev_peer[lane][site][i], lane_stream(g, i, lane)));
    }
    for (size_t i = 0; i < nl; ++i) {
        CMFTRY(group_use(g->sh[i]));
        for (size_t j = 0; j < nl; ++j)
            if (j != i) HIPCHK(hipStreamWaitEvent(lane_stream(g, i, lane), g->ev_peer[lane][site][j], 0));
    }
    return CMF_OK;
}
// ... or every enqueue worker for its own shard: record, meet (a wait must find the record made), wait for the others
static int peer_fence_shard(cmf_group_s *g, size_t i, int lane, int site)
{
    CMFTRY(lane_event(&g->ev_peer[lane][site][i]));
    HIPCHK(hipEventRecord(g->ev_peer[lane][site][i], lane_stream(g, i, lane)));
    CMFTRY(group_barrier(g));
    for (size_t j = 0; j < g->sh.size(); ++j)
        if (j != i) HIPCHK(hipStreamWaitEvent(lane_stream(g, i, lane), g->ev_peer[lane][site][j], 0));
    return CMF_OK;
}

static int peer_allreduce_launch(cmf_group_s *g, size_t i, const CmfPtrTable &tab, size_t count, int lane)
{
    const int R = (int)g->sh.size();
    size_t per = (count + R - 1) / R;
    per = (per + 3) / 4 * 4;
    int vec4 = 1;
    for (int r = 0; r < R; ++r)
        if (reinterpret_cast<uintptr_t>(tab.p[r]) % 16) vec4 = 0;
    const int blocks = (int)std::min<size_t>(1024, std::max<size_t>(1, (per / 4 + 255) / 256));
    hipLaunchKernelGGL(peer_allreduce_kernel, dim3(blocks), dim3(256), 0, lane_stream(g, i, lane), tab, R, (int)i, count, per, vec4);
    KCHK("peer_allreduce_kernel");
    return CMF_OK;
}

// ---- collectives, issued by the calling thread for all local shards -------------------------------------------------
// In-place sum over all ranks of `count` floats at offset `off` of every local shard's buffer `bufs[i]`, ordered on the
// shards' main streams (lane 0) or communication streams (lane 1).
int group_allreduce(cmf_group_s *g, const std::vector<float *> &bufs, size_t off, size_t count, int lane)
{
    const size_t nl = g->sh.size();
    if (g->nranks == 1 && g->transport != CMF_TR_RCCL) return CMF_OK;
    switch (g->transport) {
    case CMF_TR_RCCL: {
        if (lane && g->comm2.size() != nl) return fail(CMF_ERR_STATE, "the communication stream has no communicator of its own");
        if (nl > 1) RCCLCHK(g_rccl.GroupStart());
        for (size_t i = 0; i < nl; ++i) {
            CMFTRY(group_use(g->sh[i]));
            RCCLCHK(g_rccl.AllReduce(bufs[i] + off, bufs[i] + off, count, ncclFloat32, ncclSum, lane_comm(g, i, lane), lane_stream(g, i, lane)));
        }
        if (nl > 1) RCCLCHK(g_rccl.GroupEnd());
        return CMF_OK;
    }
    case CMF_TR_LOOPBACK: { // all shards share one device (and, unless loop_ms, one stream)
        CmfPtrTable tab;
        for (size_t i = 0; i < nl; ++i) tab.p[i] = bufs[i] + off;
        CMFTRY(group_use(g->sh[0]));
        if (g->loop_ms) CMFTRY(loopback_arrive(g, lane));
        const int blocks = (int)std::min<size_t>(2048, (count + 255) / 256);
        hipLaunchKernelGGL(loopback_allreduce_kernel, dim3(blocks), dim3(256), 0, lane_stream(g, 0, lane), tab, (int)nl, count);
        KCHK("loopback_allreduce_kernel");
        if (g->loop_ms) CMFTRY(loopback_depart(g, lane));
        return CMF_OK;
    }
    case CMF_TR_PEER: {
        CmfPtrTable tab;
        for (size_t i = 0; i < nl; ++i) tab.p[i] = bufs[i] + off;
        CMFTRY(peer_fence_all(g, lane, 0));
        for (size_t i = 0; i < nl; ++i) {
            CMFTRY(group_use(g->sh[i]));
            CMFTRY(peer_allreduce_launch(g, i, tab, count, lane));
        }
        return peer_fence_all(g, lane, 1);
    }
    default: { // host callbacks: one local shard
        cmf_handle_s *s = g->sh[0];
        CMFTRY(group_use(s));
        CMFTRY(group_cb_stage(g, count));
        HIPCHK(hipMemcpyAsync(g->cb_host, bufs[0] + off, count * sizeof(float), hipMemcpyDeviceToHost, lane_stream(g, 0, lane)));
        HIPCHK(hipStreamSynchronize(lane_stream(g, 0, lane)));
        const int rc = g->ar_cb(g->cb_user, g->cb_host, (int64_t)count);
        if (rc != 0) return fail(CMF_ERR_COMM, "all-reduce callback returned %d", rc);
        HIPCHK(hipMemcpyAsync(bufs[0] + off, g->cb_host, count * sizeof(float), hipMemcpyHostToDevice, lane_stream(g, 0, lane)));
        HIPCHK(hipStreamSynchronize(lane_stream(g, 0, lane)));
        return CMF_OK;
    }
    }
}

// recv[i] (nranks * count floats) = every rank's send block (count floats at send[i]), in rank order
int group_allgather(cmf_group_s *g, const std::vector<float *> &send, const std::vector<float *> &recv, size_t count)
{
    const size_t nl = g->sh.size();
    switch (g->transport) {
    case CMF_TR_RCCL: {
        if (nl > 1) RCCLCHK(g_rccl.GroupStart());
        for (size_t i = 0; i < nl; ++i) {
            CMFTRY(group_use(g->sh[i]));
            RCCLCHK(g_rccl.AllGather(send[i], recv[i], count, ncclFloat32, g->comm[i], g->sh[i]->stream));
        }
        if (nl > 1) RCCLCHK(g_rccl.GroupEnd());
        return CMF_OK;
    }
    case CMF_TR_LOOPBACK: {
        CmfPtrTable ts, tr;
        for (size_t i = 0; i < nl; ++i) { ts.p[i] = send[i]; tr.p[i] = recv[i]; }
        CMFTRY(group_use(g->sh[0]));
        if (g->loop_ms) CMFTRY(loopback_arrive(g, 0));
        const int blocks = (int)std::min<size_t>(64, (nl * count + 255) / 256);
        hipLaunchKernelGGL(loopback_allgather_kernel, dim3(blocks), dim3(256), 0, g->sh[0]->stream, ts, tr, (int)nl, (int)count);
        KCHK("loopback_allgather_kernel");
        if (g->loop_ms) CMFTRY(loopback_depart(g, 0));
        return CMF_OK;
    }
    case CMF_TR_PEER: {
        CmfPtrTable ts;
        for (size_t i = 0; i < nl; ++i) ts.p[i] = send[i];
        CMFTRY(peer_fence_all(g, 0, 0));
        for (size_t i = 0; i < nl; ++i) {
            CMFTRY(group_use(g->sh[i]));
            const int blocks = (int)std::min<size_t>(64, (nl * count + 255) / 256);
            hipLaunchKernelGGL(peer_allgather_kernel, dim3(blocks), dim3(256), 0, g->sh[i]->stream, ts, recv[i], (int)nl, (int)count);
            KCHK("peer_allgather_kernel");
        }
        return peer_fence_all(g, 0, 1);
    }
    default: {
        cmf_handle_s *s = g->sh[0];
        CMFTRY(group_use(s));
        CMFTRY(group_cb_stage(g, (size_t)(g->nranks + 1) * count));
        float *hs = g->cb_host, *hr = g->cb_host + count;
        HIPCHK(hipMemcpyAsync(hs, send[0], count * sizeof(float), hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        const int rc = g->ag_cb(g->cb_user, hs, hr, (int64_t)count);
        if (rc != 0) return fail(CMF_ERR_COMM, "all-gather callback returned %d", rc);
        HIPCHK(hipMemcpyAsync(recv[0], hr, (size_t)g->nranks * count * sizeof(float), hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        return CMF_OK;
    }
    }
}

// ---- the same collectives, issued by shard i's enqueue worker for its own shard ---------------------------------------
// (only the transports that have workers: RCCL -- one thread per device, no group call --, peer, loopback-streams)
static int shard_allreduce(cmf_group_s *g, size_t i, const std::vector<float *> &bufs, size_t off, size_t count, int lane)
{
    const size_t nl = g->sh.size();
    if (g->test_fail_shard >= 0 && (size_t)g->test_fail_shard == i) // tests (read by the calling thread in group_check_ready)
        return fail(CMF_ERR_STATE, "CMF_TEST_FAIL_SHARD: the all-reduce call of shard %zu was made to fail", i);
    switch (g->transport) {
    case CMF_TR_RCCL:
        if (lane && g->comm2.size() != nl) return fail(CMF_ERR_STATE, "the communication stream has no communicator of its own");
        RCCLCHK(g_rccl.AllReduce(bufs[i] + off, bufs[i] + off, count, ncclFloat32, ncclSum, lane_comm(g, i, lane), lane_stream(g, i, lane)));
        return CMF_OK;
    case CMF_TR_PEER: {
        CmfPtrTable tab;
        for (size_t j = 0; j < nl; ++j) tab.p[j] = bufs[j] + off;
        CMFTRY(peer_fence_shard(g, i, lane, 0));
        CMFTRY(peer_allreduce_launch(g, i, tab, count, lane));
        return peer_fence_shard(g, i, lane, 1);
    }
    case CMF_TR_LOOPBACK: { // loop_ms: shard 0's worker runs the kernel between two meetings of the workers
        CMFTRY(lane_event(&g->ev_in[lane][i]));
        HIPCHK(hipEventRecord(g->ev_in[lane][i], lane_stream(g, i, lane)));
        CMFTRY(group_barrier(g));
        if (i == 0) {
            CmfPtrTable tab;
            for (size_t j = 0; j < nl; ++j) tab.p[j] = bufs[j] + off;
            for (size_t j = 1; j < nl; ++j) HIPCHK(hipStreamWaitEvent(lane_stream(g, 0, lane), g->ev_in[lane][j], 0));
            const int blocks = (int)std::min<size_t>(2048, (count + 255) / 256);
            hipLaunchKernelGGL(loopback_allreduce_kernel, dim3(blocks), dim3(256), 0, lane_stream(g, 0, lane), tab, (int)nl, count);
            KCHK("loopback_allreduce_kernel");
            CMFTRY(lane_event(&g->ev_out[lane]));
            HIPCHK(hipEventRecord(g->ev_out[lane], lane_stream(g, 0, lane)));
        }
        CMFTRY(group_barrier(g));
        if (i > 0) HIPCHK(hipStreamWaitEvent(lane_stream(g, i, lane), g->ev_out[lane], 0));
        return CMF_OK;
    }
    default:
        return fail(CMF_ERR_STATE, "this transport has no enqueue workers");
    }
}

static int shard_allgather(cmf_group_s *g, size_t i, const std::vector<float *> &send, const std::vector<float *> &recv, size_t count)
{
    const size_t nl = g->sh.size();
    switch (g->transport) {
    case CMF_TR_RCCL:
        RCCLCHK(g_rccl.AllGather(send[i], recv[i], count, ncclFloat32, g->comm[i], g->sh[i]->stream));
        return CMF_OK;
    case CMF_TR_PEER: {
        CmfPtrTable ts;
        for (size_t j = 0; j < nl; ++j) ts.p[j] = send[j];
        CMFTRY(peer_fence_shard(g, i, 0, 0));
        const int blocks = (int)std::min<size_t>(64, (nl * count + 255) / 256);
        hipLaunchKernelGGL(peer_allgather_kernel, dim3(blocks), dim3(256), 0, g->sh[i]->stream, ts, recv[i], (int)nl, (int)count);
        KCHK("peer_allgather_kernel");
        return peer_fence_shard(g, i, 0, 1);
    }
    case CMF_TR_LOOPBACK: {
        CMFTRY(lane_event(&g->ev_in[0][i]));
        HIPCHK(hipEventRecord(g->ev_in[0][i], g->sh[i]->stream));
        CMFTRY(group_barrier(g));
        if (i == 0) {
            CmfPtrTable ts, tr;
            for (size_t j = 0; j < nl; ++j) { ts.p[j] = send[j]; tr.p[j] = recv[j]; }
            for (size_t j = 1; j < nl; ++j) HIPCHK(hipStreamWaitEvent(g->sh[0]->stream, g->ev_in[0][j], 0));
            const int blocks = (int)std::min<size_t>(64, (nl * count + 255) / 256);
            hipLaunchKernelGGL(loopback_allgather_kernel, dim3(blocks), dim3(256), 0, g->sh[0]->stream, ts, tr, (int)nl, (int)count);
            KCHK("loopback_allgather_kernel");
            CMFTRY(lane_event(&g->ev_out[0]));
            HIPCHK(hipEventRecord(g->ev_out[0], g->sh[0]->stream));
        }
        CMFTRY(group_barrier(g));
        if (i > 0) HIPCHK(hipStreamWaitEvent(g->sh[i]->stream, g->ev_out[0], 0));
        return CMF_OK;
    }
    default:
        return fail(CMF_ERR_STATE, "this transport has no enqueue workers");
    }
}

// ---- phases as step lists ---------------------------------------------------------------------------------------------
// A phase of the sharded iteration is a list of steps: per-shard segments (kernel launches of ONE shard on its own
// streams) and collectives.  Without workers the calling thread walks the list -- a segment for every shard in shard
// order, a collective as one (grouped) call; with workers every shard's worker walks the whole list for its shard.
// Everything a step needs is held by value: the list outlives the call that posted it.
struct GroupStep {
    const char *name = "";            // roctx range of the step (a string literal)
    std::function<int(size_t)> seg;   // per-shard work, or empty
    int coll = 0;                     // 0 none, 1 all-reduce (in place in a[i] + off), 2 all-gather (a[i] -> b[i])
    std::vector<float *> a, b;
    size_t off = 0, count = 0;
    int lane = 0;
};
using StepList = std::vector<GroupStep>;

static void step_seg(StepList &st, const char *name, std::function<int(size_t)> fn)
{
    GroupStep s;
    s.name = name;
    s.seg = std::move(fn);
    st.push_back(std::move(s));
}
// is the halo of H carried in the W-phase all-reduce on this group, in its current formulation?  (cmf_group_s::halo_opt)
static inline bool halo_in_ar(const cmf_group_s *g) { return g->halo_opt && g->halo_can && !g->gram && g->nranks > 1; }

static void step_allreduce(cmf_group_s *g, StepList &st, const char *name, const std::vector<float *> &bufs, size_t off, size_t count, int lane = 0)
{
    if (g->nranks == 1 && g->transport != CMF_TR_RCCL) return;
    g->n_allreduce += 1;
    GroupStep s;
    s.name = name;
    s.coll = 1; s.a = bufs; s.off = off; s.count = count; s.lane = lane;
    st.push_back(std::move(s));
}
static void step_allgather(cmf_group_s *g, StepList &st, const char *name, const std::vector<float *> &send, const std::vector<float *> &recv, size_t count)
{
    g->n_allgather += 1;
    GroupStep s;
    s.name = name;
    s.coll = 2; s.a = send; s.b = recv; s.count = count;
    st.push_back(std::move(s));
}

static int group_run_shard(cmf_group_s *g, const StepList &st, size_t i)
{
    for (const GroupStep &s : st) {
        RoctxRange range(s.name);
        CMFTRY(group_use(g->sh[i]));
        if (s.seg) CMFTRY(s.seg(i));
        if (s.coll == 1) CMFTRY(shard_allreduce(g, i, s.a, s.off, s.count, s.lane));
        else if (s.coll == 2) CMFTRY(shard_allgather(g, i, s.a, s.b, s.count));
    }
    return CMF_OK;
}

// Runs the list: returns when it has been ENQUEUED (no workers) or merely posted (workers: failures then surface at the
// next group_join, and through group_health while the host polls for a loss).
static int group_run(cmf_group_s *g, StepList &&st)
{
    if (st.empty()) return CMF_OK;
    if (g->pool.empty() || g->force_inline > 0) {
        for (const GroupStep &s : st) {
            RoctxRange range(s.name);
            if (s.seg)
                for (size_t i = 0; i < g->sh.size(); ++i) {
                    CMFTRY(group_use(g->sh[i]));
                    CMFTRY(s.seg(i));
                }
            if (s.coll == 1) CMFTRY(group_allreduce(g, s.a, s.off, s.count, s.lane));
            else if (s.coll == 2) CMFTRY(group_allgather(g, s.a, s.b, s.count));
        }
        return CMF_OK;
    }
    auto shared = std::make_shared<StepList>(std::move(st));
    for (size_t i = 0; i < g->pool.size(); ++i)
        cmf_pool_post(g->pool, i, [g, shared, i]() { return group_run_shard(g, *shared, i); });
    return CMF_OK;
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
int group_gather_doubles(cmf_group_s *g, const std::vector<double> &vals, std::vector<double> &out, int n)
{
    const size_t nl = g->sh.size();
    out.assign((size_t)g->nranks * n, 0.0);
    if (g->one_process) { // all ranks are local: no transport needed
        for (size_t i = 0; i < nl; ++i)
            for (int j = 0; j < n; ++j) out[(size_t)g->rank[i] * n + j] = vals[i * n + j];
        return CMF_OK;
    }
    CMFTRY(group_join(g));
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
    CMFTRY(group_allgather(g, send, recv, words));
    cmf_handle_s *s = g->sh[0];
    CMFTRY(group_use(s));
    HIPCHK(hipMemcpyAsync(out.data(), recv[0], (size_t)g->nranks * n * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return CMF_OK;
}

int group_check_ready(cmf_group_s *g)
{
    if (g->failed) return fail(CMF_ERR_COMM, "this group has failed (a collective did not complete, or a communicator reported an error): destroy the handle");
    // test hook (honoured only with CMF_TEST_HOOKS=1): that shard's next all-reduce call fails.  The environment is read HERE, by
    // the calling thread at a public entry -- never by the enqueue workers on the per-iteration path (getenv is not safe against
    // a concurrent setenv of the host program)
    g->test_fail_shard = (int)test_hook("CMF_TEST_FAIL_SHARD", -1);
    for (cmf_handle_s *s : g->sh) {
        if (!s->factors_set) return fail(CMF_ERR_STATE, "factors not set: call cmf_set_factors first");
        if (!s->have_data) return fail(CMF_ERR_STATE, "handle was created without data");
    }
    return CMF_OK;
}

int group_sync(cmf_group_s *g)
{
    CMFTRY(group_join(g));
    for (cmf_handle_s *s : g->sh) {
        CMFTRY(group_use(s));
        HIPCHK(hipStreamSynchronize(s->stream));
        if (s->comm_stream) HIPCHK(hipStreamSynchronize(s->comm_stream));
    }
    return CMF_OK;
}

// Called while the host polls for a loss that only shard 0 posts: a failed enqueue job, a fault on another local shard's
// stream, or an asynchronous RCCL error on any local communicator, must end the wait (CMF_ERR_HIP / CMF_ERR_COMM) instead
// of hanging it.
static int group_health(cmf_group_s *g)
{
    if (g->pool.abort.load(std::memory_order_acquire)) {
        const int rc = group_join(g);
        return rc != CMF_OK ? rc : fail(CMF_ERR_STATE, "an enqueue worker of the group gave up");
    }
    const bool posted = group_enqueued(g); // (a worker that has not enqueued yet leaves its stream idle: that is not a fault)
    for (size_t i = 0; i < g->sh.size(); ++i) {
        cmf_handle_s *s = g->sh[i];
        if (posted && i > 0 && s->stream != g->sh[0]->stream) {
            CMFTRY(group_use(s));
            const hipError_t e = hipStreamQuery(s->stream);
            if (e != hipSuccess && e != hipErrorNotReady)
                return fail(CMF_ERR_HIP, "shard %d (device %d) failed: %s", g->rank[i], s->device, hipGetErrorString(e));
        }
        if (g->transport == CMF_TR_RCCL && g_rccl.CommGetAsyncError)
            for (const std::vector<ncclComm_t> *cs : {&g->comm, &g->comm2})
                if (i < cs->size() && (*cs)[i]) {
                    ncclResult_t ae = ncclSuccess;
                    if (g_rccl.CommGetAsyncError((*cs)[i], &ae) == ncclSuccess && ae != ncclSuccess && (int)ae != 7 /* ncclInProgress */) {
                        g->failed = true;
                        return fail(CMF_ERR_COMM, "RCCL reported an asynchronous error on rank %d: %s", g->rank[i], g_rccl.GetErrorString(ae));
                    }
                }
    }
    return group_use(g->sh[0]);
}

// (L-1)-column H halo exchange: pack -> one all-gather -> unpack
static void build_exchange_halos(cmf_group_s *g, StepList &st)
{
    cmf_handle_s *s0 = g->sh[0];
    const int rows = s0->d.L - 1;
    g->halos_current = true;
    g->halos_pending = false;
    g->halo_wide = false;
    if (rows < 1 || g->nranks == 1) return;
    if (halo_in_ar(g)) { // the wide exchange (last 2(L-1) | first L-1 columns of every rank), here as an all-gather of the slots
        g->halo_wide = true;
        step_seg(st, "cmf:halo pack (wide)", [g, rows](size_t i) {
            cmf_handle_s *s = g->sh[i];
            const CmfDims &d = s->d;
            hipLaunchKernelGGL(halo_pack3_kernel, dim3(8), dim3(256), 0, s->stream, s->H, g->halo3_send[i], d.PADL, d.Tl, rows, d.K32, 0, 1);
            KCHK("halo_pack3_kernel");
            return CMF_OK;
        });
        step_allgather(g, st, "cmf:all-gather of the H halos (wide)", g->halo3_send, g->halo3_all, (size_t)(3 * g->HC));
        step_seg(st, "cmf:halo unpack (wide)", [g, rows](size_t i) {
            cmf_handle_s *s = g->sh[i];
            const CmfDims &d = s->d;
            hipLaunchKernelGGL(halo_unpack3_kernel, dim3(8), dim3(256), 0, s->stream, s->H, s->Ht, g->halo3_all[i], d.PADL, d.Tl, rows, d.K32, d.TP,
                               g->rank[i], g->nranks);
            KCHK("halo_unpack3_kernel");
            return CMF_OK;
        });
        return;
    }
    step_seg(st, "cmf:halo pack", [g, rows](size_t i) {
        cmf_handle_s *s = g->sh[i];
        const CmfDims &d = s->d;
        hipLaunchKernelGGL(halo_pack2_kernel, dim3(8), dim3(256), 0, s->stream, s->H, g->halo_send[i], d.PADL, d.PADL + d.Tl - rows, rows, d.K32);
        KCHK("halo_pack2_kernel");
        return CMF_OK;
    });
    step_allgather(g, st, "cmf:all-gather of the H halos", g->halo_send, g->halo_all, (size_t)(2 * g->HC));
    step_seg(st, "cmf:halo unpack", [g, rows](size_t i) {
        cmf_handle_s *s = g->sh[i];
        const CmfDims &d = s->d;
        hipLaunchKernelGGL(halo_unpack2_kernel, dim3(8), dim3(256), 0, s->stream, s->H, s->Ht, s->halo[2], s->halo[3],
                           d.PADL - rows, d.PADL + d.Tl, rows, d.K32, d.TP);
        KCHK("halo_unpack2_kernel");
        return CMF_OK;
    });
}
static int group_exchange_halos(cmf_group_s *g)
{
    StepList st;
    build_exchange_halos(g, st);
    return group_run(g, std::move(st));
}

// sum((conv(W,H) - data)^2) of every local shard -> its tail slots of the all-reduce buffer (and d_scalar[0]).
// defer: the per-tile sums are reduced by the next update_motifs!' slab sum instead (CmfLossCarry), right in front of the
// all-reduce their total rides on.
static void build_loss_partials(cmf_group_s *g, StepList &st, bool defer = false)
{
    const size_t toff = group_tail_off(g);
    step_seg(st, "cmf:loss conv (mult.jl:55-57)", [g, defer, toff](size_t i) {
        cmf_handle_s *s = g->sh[i];
        CMFTRY(launch_loss_conv(s)); // mult.jl:55-57
        if (defer) {
            s->carry = CmfLossCarry{s->partial, s->conv_partials, s->d_scalar, nullptr, g->red[i] + toff, (int)g->tail, g->rank[i]};
            return CMF_OK;
        }
        hipLaunchKernelGGL(loss_tail_kernel, dim3(1), dim3(256), 0, s->stream, s->partial, s->conv_partials, s->d_scalar,
                           g->red[i] + toff, (int)g->tail, g->rank[i]);
        KCHK("loss_tail_kernel");
        return CMF_OK;
    });
}

// the tail of the all-reduce buffer right now (synchronous path): all-gather of the (hi, lo) pairs, read back
static int group_loss_now(cmf_group_s *g, StepList &&st, double *sumsq)
{
    const size_t nl = g->sh.size();
    cmf_handle_s *s = g->sh[0];
    if (g->nranks == 1 && g->transport != CMF_TR_RCCL) {
        CMFTRY(group_run(g, std::move(st)));
        CMFTRY(group_join(g));
        return read_scalar(s, 0, sumsq);
    }
    std::vector<float *> send(nl);
    for (size_t i = 0; i < nl; ++i) send[i] = g->red[i] + group_tail_off(g) + 2 * g->rank[i];
    step_allgather(g, st, "cmf:all-gather of the loss pairs", send, g->loss_all, 2);
    CMFTRY(group_run(g, std::move(st)));
    CMFTRY(group_join(g));
    CMFTRY(group_use(s));
    float *stage = g->h_tail + 2 * g->slot_len; // not a ring slot: a pending one-iteration-late loss may still sit there
    HIPCHK(hipMemcpyAsync(stage, g->loss_all[0], (size_t)(2 * g->nranks) * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    *sumsq = group_decode_tail(g, stage);
    return CMF_OK;
}

// overlap form: numW needs H only -- contract it and start its all-reduce on the communication stream (lane 1: its own
// stream AND its own communicator, so that it never shares one with the main stream's collectives that are issued while
// it is in flight).  In the Gram form the whole payload [numW | HH] needs H only, so ALL of the bulk all-reduce runs
// underneath the loss conv.
static void build_start_num(cmf_group_s *g, StepList &st)
{
    const size_t half = (size_t)g->LKN2 / 2;
    const int gram = g->gram;
    step_seg(st, "cmf:numW contraction (overlap form)", [g, half, gram](size_t i) {
        cmf_handle_s *s = g->sh[i];
        if (gram) CMFTRY(gram_w_partial(s, s->numden + half));
        else CMFTRY(w_partial_half_impl(s, 0));
        HIPCHK(hipEventRecord(s->ev_c0, s->stream));
        HIPCHK(hipStreamWaitEvent(s->comm_stream, s->ev_c0, 0));
        return CMF_OK;
    });
    step_allreduce(g, st, "cmf:all-reduce of numW on the communication stream", g->red, 0, gram ? half + (size_t)g->HHsz : half, 1);
    step_seg(st, "cmf:overlap join event", [g](size_t i) {
        cmf_handle_s *s = g->sh[i];
        HIPCHK(hipEventRecord(s->ev_c1, s->comm_stream));
        return CMF_OK;
    });
    g->num_ready = true;
}

// update_motifs! on the group (mult.jl:23-39).  ring_slot >= 0: after the all-reduce the tail (the previous
// iteration's loss pairs of every rank) is copied to pinned host slot `ring_slot`, which the host polls.
static void build_update_motifs(cmf_group_s *g, StepList &st, double l1W, double l2W, int ring_slot = -1)
{
    if (!g->halos_current && !g->halos_pending) build_exchange_halos(g, st); // (pending: the L-1 columns in front are valid -- all this phase reads)
    const size_t half = (size_t)g->LKN2 / 2;
    const size_t toff = group_tail_off(g);
    const int gram = g->gram;
    // the halos of the H phase before ride behind the loss tail (they were packed there by that phase)
    const bool carry_halo = halo_in_ar(g) && g->halos_pending;
    const size_t hcount = carry_halo ? (size_t)g->halo_len : 0;
    if (g->overlap) {
        if (!g->num_ready) build_start_num(g, st);
        if (gram) { // the bulk is in flight on the communication stream: only the loss tail is left for this stream
            step_allreduce(g, st, "cmf:all-reduce of the loss tail", g->red, toff, (size_t)g->tail);
        } else {
            step_seg(st, "cmf:denomW contraction", [g](size_t i) { return w_partial_half_impl(g->sh[i], 1); });
            step_allreduce(g, st, "cmf:all-reduce of [denomW | tail | halos]", g->red, half, half + (size_t)g->tail + hcount);
        }
        step_seg(st, "cmf:wait for the communication stream", [g](size_t i) {
            cmf_handle_s *s = g->sh[i];
            HIPCHK(hipStreamWaitEvent(s->stream, s->ev_c1, 0));
            return CMF_OK;
        });
        g->num_ready = false;
    } else {
        step_seg(st, "cmf:W phase contractions (mult.jl:28-34)", [g, half, gram](size_t i) {
            cmf_handle_s *s = g->sh[i];
            if (gram) return gram_w_partial(s, s->numden + half); // [numW | this shard's share of HH | tail]
            return w_partial_impl(s);
        });
        step_allreduce(g, st, "cmf:all-reduce of [numW | denomW | tail | halos]", g->red, 0, toff + (size_t)g->tail + hcount);
    }
    float *ring = nullptr;
    if (ring_slot >= 0) { // shard 0's W update also drops the reduced loss pairs + a stamp into the pinned ring slot
        ring = g->h_tail + (size_t)ring_slot * g->slot_len;
        for (int j = 0; j < 2 * g->nranks; ++j) reinterpret_cast<volatile unsigned *>(ring)[j] = CMF_SENTINEL32; // collected an iteration ago
    }
    const int npairs = 2 * g->nranks;
    step_seg(st, "cmf:W update (mult.jl:37-38)", [g, half, toff, gram, l1W, l2W, ring, npairs](size_t i) {
        cmf_handle_s *s = g->sh[i];
        if (i == 0 && ring) {
            if (gram) return gram_w_finish(s, s->numden + half, l1W, l2W, g->red[0] + toff, ring, npairs);
            return w_apply_impl(s, l1W, l2W, g->red[0] + toff, ring, npairs);
        }
        if (gram) return gram_w_finish(s, s->numden + half, l1W, l2W);
        return w_apply_impl(s, l1W, l2W);
    });
    if (carry_halo) { // every rank's outer columns of the new H have arrived with the sums: the halos are whole again, 2(L-1) columns out on the left
        const int rows = g->sh[0]->d.L - 1;
        step_seg(st, "cmf:halo unpack (from the all-reduce)", [g, rows, toff](size_t i) {
            cmf_handle_s *s = g->sh[i];
            const CmfDims &d = s->d;
            hipLaunchKernelGGL(halo_unpack3_kernel, dim3(8), dim3(256), 0, s->stream, s->H, s->Ht, g->red[i] + toff + (size_t)g->tail, d.PADL, d.Tl, rows,
                               d.K32, d.TP, g->rank[i], g->nranks);
            KCHK("halo_unpack3_kernel");
            return CMF_OK;
        });
        g->halos_current = true;
        g->halos_pending = false;
        g->halo_wide = true;
    }
}
int group_update_motifs(cmf_group_s *g, double l1W, double l2W)
{
    StepList st;
    build_update_motifs(g, st, l1W, l2W);
    return group_run(g, std::move(st));
}

// update_feature_maps! on the group (mult.jl:42-58), without the loss read-back: with `defer` the loss partials stay in the
// tail of the all-reduce buffer and ride on the next update_motifs!.
static void build_update_feature_maps(cmf_group_s *g, StepList &st, double l1H, double l2H, bool defer)
{
    const bool in_ar = halo_in_ar(g);
    if (!g->halos_current || (in_ar && !g->halo_wide)) build_exchange_halos(g, st);
    const int gram = g->gram;
    step_seg(st, "cmf:H phase (mult.jl:44-52)", [g, gram, in_ar, l1H, l2H](size_t i) {
        return gram ? gram_h_update(g->sh[i], l1H, l2H) : h_update_impl(g->sh[i], l1H, l2H, in_ar && g->rank[i] > 0);
    });
    g->num_ready = false;
    if (in_ar) { // no exchange here: the loss conv and the next W phase read the L-1 columns in front, which this shard has just updated itself
        const size_t toff = group_tail_off(g);
        const int rows = g->sh[0]->d.L - 1;
        step_seg(st, "cmf:halo pack (into the all-reduce tail)", [g, rows, toff](size_t i) {
            cmf_handle_s *s = g->sh[i];
            const CmfDims &d = s->d;
            hipLaunchKernelGGL(halo_pack3_kernel, dim3(16), dim3(256), 0, s->stream, s->H, g->red[i] + toff + (size_t)g->tail, d.PADL, d.Tl, rows, d.K32,
                               g->rank[i], g->nranks);
            KCHK("halo_pack3_kernel");
            return CMF_OK;
        });
        g->halos_current = false;
        g->halos_pending = true;
        g->halo_wide = false;
    } else build_exchange_halos(g, st);
    if (g->overlap) build_start_num(g, st); // for the next update_motifs!: H and its halos are final now
    // (Gram + overlap: the next W phase has no slab sum left on this stream for a deferred reduction to ride on)
    build_loss_partials(g, st, defer && !(g->gram && g->overlap));
}
// sumsq != NULL: also reduce the loss now (synchronises)
int group_update_feature_maps(cmf_group_s *g, double l1H, double l2H, double *sumsq)
{
    StepList st;
    build_update_feature_maps(g, st, l1H, l2H, sumsq == nullptr);
    return sumsq ? group_loss_now(g, std::move(st), sumsq) : group_run(g, std::move(st));
}

int group_compute_loss(cmf_group_s *g, double *loss)
{
    StepList st;
    if (!g->halos_current && !g->halos_pending) build_exchange_halos(g, st); // (pending: the L-1 columns in front, all the conv reads, are valid)
    build_loss_partials(g, st);
    double ss = 0.0;
    CMFTRY(group_loss_now(g, std::move(st), &ss));
    *loss = std::sqrt(ss) / g->data_norm;
    return CMF_OK;
}

// n MU iterations back to back (alternating.jl:51-54 n times).  The loss of iteration i travels in the tail of
// iteration i+1's all-reduce and is read from pinned memory after iteration i+1 has been enqueued, so the host never
// stalls the device between iterations; the last loss is flushed with the small all-gather.  stamps (optional):
// host seconds since entry at which each loss became known.
int group_iterate(cmf_group_s *g, int64_t n, int eval_mode, double l1W, double l2W, double l1H, double l2H,
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
    const std::function<bool()> enqueued = [g]() { return group_enqueued(g); };
    // cmf_fit's time_hist: timing events on shard 0's stream behind every iteration's loss conv (DevStamps, cmf_api.hip);
    // whoever enqueues shard 0 records them
    auto ds = std::make_shared<DevStamps>();
    if (s0->dev_stamps && stamps) {
        CMFTRY(group_join(g));
        CMFTRY(group_use(s0));
        CMFTRY(ds->begin(s0->stream, n));
    }
    for (int64_t it = 0; it < n; ++it) {
        const bool last = (it + 1 == n);
        const auto t_enq = std::chrono::steady_clock::now();
        StepList st;
        build_update_motifs(g, st, l1W, l2W, it > 0 ? (int)((it - 1) & 1) : -1);
        build_update_feature_maps(g, st, l1H, l2H, !last);
        if (ds->active) step_seg(st, "cmf:iteration stamp", [g, ds, it](size_t i) { return i == 0 ? ds->mark(g->sh[0]->stream, it) : CMF_OK; });
        double ss = 0.0;
        if (last) CMFTRY(group_loss_now(g, std::move(st), &ss));
        else {
            CMFTRY(group_run(g, std::move(st)));
            g->enqueue_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_enq).count();
            g->enqueue_iters += 1;
        }
        if (it > 0) {
            const int slot = (int)((it - 1) & 1);
            CMFTRY(group_use(s0));
            const float *ring = g->h_tail + (size_t)slot * g->slot_len;
            const int rc = wait_words<unsigned>(s0->stream, reinterpret_cast<const volatile unsigned *>(ring), 2 * g->nranks, CMF_SENTINEL32, &health, &enqueued);
            if (rc != CMF_OK) {
                if (rc == CMF_ERR_COMM) g->failed = true;
                return rc;
            }
            losses[it - 1] = std::sqrt(group_decode_tail(g, ring)) / g->data_norm;
            if (stamps) stamps[it - 1] = now();
        }
        if (last) {
            losses[it] = std::sqrt(ss) / g->data_norm;
            if (stamps) stamps[it] = now();
        }
    }
    CMFTRY(group_join(g));
    CMFTRY(group_use(s0));
    return ds->finish(stamps, n);
}

int group_set_factors(cmf_group_s *g, const double *W, const double *H)
{
    CMFTRY(group_join(g));
    for (size_t i = 0; i < g->sh.size(); ++i) {
        cmf_handle_s *s = g->sh[i];
        // one process: H is the global K x T matrix (column-major: a shard's columns are contiguous)
        const double *Hs = (H && g->one_process) ? H + (size_t)g->t0[(size_t)g->rank[i]] * g->K : H;
        CMFTRY(set_factors_impl(s, W, Hs));
    }
    g->num_ready = false;
    g->halos_current = false;
    g->halos_pending = false;
    g->halo_wide = false;
    CMFTRY(group_exchange_halos(g));
    return group_join(g);
}

int group_get_factors(cmf_group_s *g, double *W, double *H)
{
    CMFTRY(group_sync(g));
    for (size_t i = 0; i < g->sh.size(); ++i) {
        cmf_handle_s *s = g->sh[i];
        double *Hs = (H && g->one_process) ? H + (size_t)g->t0[(size_t)g->rank[i]] * g->K : H;
        CMFTRY(get_factors_impl(s, i == 0 ? W : nullptr, Hs));
    }
    return CMF_OK;
}

// a stream of a failed group may never drain (a collective kernel waiting for a peer that is gone): wait a little, then move on
static void bounded_stream_sync(hipStream_t st, bool failed)
{
    if (!failed) {
        (void)hipStreamSynchronize(st);
        return;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (hipStreamQuery(st) == hipErrorNotReady && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.0)
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
}

// false: a worker had to be abandoned inside a job that never returned -- its job holds pointers to the group and its shards, so
// NOTHING of them is freed (leaked on purpose: a late wake-up must not find freed memory); the caller leaves the handles alone too.
bool group_destroy(cmf_group_s *g)
{
    if (!g) return true;
    if (!g->failed) (void)group_join(g); // (bounded; marks the group failed when a worker is stuck in a collective call)
    if (g->failed) {
        // A failed group: FIRST abort the communicators -- that unblocks collective kernels that wait for a peer and wakes a worker
        // out of a collective call that would never return -- THEN give the workers a short while to come back, and only detach
        // one that still has not (its CmfWorker and the pool's shared state stay alive for it: cmf_workers.h).  (The full
        // CMF_WAIT_TIMEOUT_S was already spent when the group was marked failed: not a second time here.)
        if (g->transport == CMF_TR_RCCL && g_rccl.dl && g_rccl.CommAbort)
            for (std::vector<ncclComm_t> *cs : {&g->comm, &g->comm2})
                for (ncclComm_t &c : *cs)
                    if (c) { (void)g_rccl.CommAbort(c); c = nullptr; }
        g->pool.abort.store(true, std::memory_order_release);
        if (!g->pool.empty()) (void)cmf_pool_wait(g->pool, 5.0);
    }
    if (group_stop_workers(g) != 0) return false; // (joins; abandon_busy only for a worker that is still inside its job)
    for (cmf_handle_s *s : g->sh) {
        (void)hipSetDevice(s->device);
        s->streams_may_hang = g->failed;
        bounded_stream_sync(s->stream, g->failed);
        if (s->comm_stream) bounded_stream_sync(s->comm_stream, g->failed);
    }
    if (g->transport == CMF_TR_RCCL && g_rccl.dl && !g->failed)
        for (std::vector<ncclComm_t> *cs : {&g->comm2, &g->comm})
            for (ncclComm_t c : *cs)
                if (c) (void)g_rccl.CommDestroy(c);
    for (size_t i = 0; i < g->sh.size(); ++i) {
        cmf_handle_s *s = g->sh[i];
        (void)hipSetDevice(s->device);
        s->numden = s->numden_own;
        for (int w = 0; w < 4; ++w) s->halo[w] = s->halo_own[w];
        if (g->failed) continue; // the device may still be reading these: leaked on purpose
        if (i < g->red.size() && g->red[i]) (void)hipFree(g->red[i]);
        if (i < g->halo_send.size() && g->halo_send[i]) (void)hipFree(g->halo_send[i]);
        if (i < g->halo_all.size() && g->halo_all[i]) (void)hipFree(g->halo_all[i]);
        if (i < g->halo3_send.size() && g->halo3_send[i]) (void)hipFree(g->halo3_send[i]);
        if (i < g->halo3_all.size() && g->halo3_all[i]) (void)hipFree(g->halo3_all[i]);
        if (i < g->loss_all.size() && g->loss_all[i]) (void)hipFree(g->loss_all[i]);
    }
    if (g->h_tail && !g->failed) (void)hipHostFree(g->h_tail);
    if (g->cb_host) (void)hipHostFree(g->cb_host);
    for (size_t i = 0; i < g->gbuf.size() && i < g->sh.size(); ++i)
        if (g->gbuf[i] && !g->failed) { (void)hipSetDevice(g->sh[i]->device); (void)hipFree(g->gbuf[i]); }
    for (int c = 0; c < 2; ++c) {
        for (hipEvent_t e : g->ev_in[c])
            if (e) (void)hipEventDestroy(e);
        if (g->ev_out[c]) (void)hipEventDestroy(g->ev_out[c]);
        for (int site = 0; site < 2; ++site)
            for (hipEvent_t e : g->ev_peer[c][site])
                if (e) (void)hipEventDestroy(e);
    }
    delete g;
    return true;
}

// streams / events of the overlap form and common post-construction steps of a shard that joins a group
static int group_prepare_shard(cmf_handle_s *s)
{
    CMFTRY(group_use(s));
    if (!s->own_comm_stream) HIPCHK(stream_acquire(s->device, &s->own_comm_stream));
    s->comm_stream = s->own_comm_stream;
    if (!s->ev_c0) HIPCHK(hipEventCreateWithFlags(&s->ev_c0, hipEventDisableTiming));
    if (!s->ev_c1) HIPCHK(hipEventCreateWithFlags(&s->ev_c1, hipEventDisableTiming));
    return CMF_OK;
}

// The communication stream's own communicators (lane 1).  One process: a second ncclCommInitAll over the same devices;
// one process per shard: cmf_comm_init_overlap hands in a second ncclUniqueId.  Other transports have nothing to create.
int group_ensure_lane1(cmf_group_s *g, const void *id128)
{
    if (g->transport != CMF_TR_RCCL || g->comm2.size() == g->sh.size()) return CMF_OK;
    CMFTRY(group_join(g));
    const size_t nl = g->sh.size();
    std::vector<ncclComm_t> c2(nl, nullptr);
    (void)hipGetLastError();
    if (g->one_process) {
        std::vector<int> devs(nl);
        for (size_t i = 0; i < nl; ++i) devs[i] = g->sh[i]->device;
        RCCLCHK(g_rccl.CommInitAll(c2.data(), (int)nl, devs.data()));
    } else {
        if (!id128) return fail(CMF_ERR_STATE, "the overlap form needs a communicator of its own for the communication stream: call cmf_comm_init_overlap (every rank, with a second id from cmf_comm_unique_id) before switching allreduce_overlap on");
        ncclUniqueId id;
        std::memcpy(&id, id128, sizeof(id));
        CMFTRY(group_use(g->sh[0]));
        RCCLCHK(g_rccl.CommInitRank(&c2[0], g->nranks, id, g->rank[0]));
    }
    g->comm2 = c2;
    return CMF_OK;
}

static int group_finish_norm(cmf_group_s *g)
{
    // ... and, in the same exchange, whether EVERY shard can run the H phase on the L-1 columns in front of its own (the halo of H then
    // travels in the W-phase all-reduce: cmf_group_s::halo_opt): K a multiple of 32, 1 <= L-1 <= 64, at least 2(L-1) own columns, and the
    // L-1 columns of data in front resident (cmf_create_multi uploads them; one process per shard: cmf_shard_set_left_data)
    std::vector<double> vals(2 * g->sh.size()), all;
    for (size_t i = 0; i < g->sh.size(); ++i) {
        const cmf_handle_s *s = g->sh[i];
        const int hx = s->d.L - 1;
        const bool shape_ok = s->d.K % 32 == 0 && hx >= 1 && hx <= 64 && s->d.Tl >= 2 * hx;
        vals[2 * i] = s->data_sumsq;
        vals[2 * i + 1] = (shape_ok && (s->t_offset == 0 || (s->halo_ext == hx && s->left_data))) ? 1.0 : 0.0;
    }
    CMFTRY(group_gather_doubles(g, vals, all, 2));
    g->data_sumsq = 0.0;
    g->halo_can = g->nranks > 1;
    for (int r = 0; r < g->nranks; ++r) {
        g->data_sumsq += all[(size_t)2 * r]; // rank order: identical on every rank
        if (all[(size_t)2 * r + 1] != 1.0) g->halo_can = false;
    }
    g->data_norm = std::sqrt(g->data_sumsq);  // mult.jl:13 over all shards
    for (cmf_handle_s *s : g->sh) s->data_norm = g->data_norm;
    return CMF_OK;
}

// ---- groups: construction ---------------------------------------------------------------------------------------
int cmf_create_multi(cmf_handle *out, int ndev, const int *devices, int transport,
                     int64_t N, int64_t T, int64_t K, int64_t L, const double *data)
{
    if (!out) return fail(CMF_ERR_ARG, "handle pointer is NULL");
    *out = nullptr;
    if (!devices || !data) return fail(CMF_ERR_ARG, "NULL argument");
    if (ndev < 1 || ndev > CMF_MAX_LOCAL) return fail(CMF_ERR_ARG, "ndev must be 1..%d (got %d)", CMF_MAX_LOCAL, ndev);
    if (N < 1 || T < 1 || K < 1 || L < 1) return fail(CMF_ERR_ARG, "N, T, K, L must all be >= 1");
    bool all_same = true, distinct = true;
    for (int i = 0; i < ndev; ++i)
        for (int j = 0; j < i; ++j) {
            if (devices[i] != devices[j]) all_same = false;
            else distinct = false;
        }
    int tr;
    if (transport == CMF_COMM_AUTO) tr = (ndev > 1 && distinct) ? CMF_TR_RCCL : CMF_TR_LOOPBACK;
    else if (transport == CMF_COMM_RCCL) tr = CMF_TR_RCCL;
    else if (transport == CMF_COMM_LOOPBACK || transport == CMF_COMM_LOOPBACK_STREAMS) tr = CMF_TR_LOOPBACK;
    else if (transport == CMF_COMM_PEER) tr = CMF_TR_PEER;
    else return fail(CMF_ERR_ARG, "unknown transport %d", transport);
    if (tr == CMF_TR_RCCL && !distinct) return fail(CMF_ERR_ARG, "RCCL needs distinct devices (a device is listed twice)");
    if (tr == CMF_TR_LOOPBACK && !all_same) return fail(CMF_ERR_ARG, "the loopback transport needs all shards on one device; list distinct devices for RCCL");
    if (tr == CMF_TR_PEER && !(distinct || all_same)) return fail(CMF_ERR_ARG, "the peer transport takes distinct devices, or one device for every shard (rehearsal)");
    if (tr == CMF_TR_PEER && ndev > 1 && distinct) {
        // every device maps every other one's memory (xGMI): the transport's kernels read and write the peers' buffers directly
        for (int i = 0; i < ndev; ++i) {
            HIPCHK(hipSetDevice(devices[i]));
            for (int j = 0; j < ndev; ++j) {
                if (i == j) continue;
                int can = 0;
                HIPCHK(hipDeviceCanAccessPeer(&can, devices[i], devices[j]));
                if (!can) return fail(CMF_ERR_COMM, "device %d cannot access device %d's memory: the peer transport needs peer access between all devices of the group", devices[i], devices[j]);
                const hipError_t e = hipDeviceEnablePeerAccess(devices[j], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                    return fail(CMF_ERR_HIP, "hipDeviceEnablePeerAccess(%d) on device %d failed: %s", devices[j], devices[i], hipGetErrorString(e));
                (void)hipGetLastError();
            }
        }
    }
    std::vector<int64_t> t0, t1;
    group_partition(T, ndev, L, t0, t1);
    for (int r = 0; r < ndev; ++r)
        if (ndev > 1 && t1[r] - t0[r] < std::max<int64_t>(L - 1, 1))
            return fail(CMF_ERR_UNSUPPORTED, "T=%lld is too short to shard over %d devices with L=%lld (every shard needs >= L-1 columns)",
                        (long long)T, ndev, (long long)L);

    cmf_group_s *g = new cmf_group_s();
    cmf_handle_s *root = new cmf_handle_s();
    root->root_only = true;
    root->group = g;
    root->device = devices[0];
    g->nranks = ndev;
    g->transport = tr;
    g->one_process = true;
    g->N = N; g->T = T; g->K = K; g->L = L;
    g->t0 = t0; g->t1 = t1;
    auto bail = [&](int rc) { cmf_destroy(root); return rc; };
    for (int r = 0; r < ndev; ++r) {
        cmf_handle_s *s = nullptr;
        // column-major N x T: the block [t0, t1 + halo_r) is contiguous
        int rc = create_impl(&s, devices[r], N, t1[r] - t0[r], K, L, data + (size_t)t0[r] * N, t0[r], T, ndev > 1);
        if (rc == CMF_OK && s->halo_ext) { // the L-1 columns of data in front of the shard (data' only: the operand of its transconv's front block)
            rc = upload_cols(s, data + (size_t)(t0[r] - s->halo_ext) * N, -s->halo_ext, s->halo_ext, false, false);
            s->left_data = rc == CMF_OK;
        }
        if (rc != CMF_OK) return bail(rc);
        g->sh.push_back(s);
        g->rank.push_back(r);
        s->group = g;
        rc = group_prepare_shard(s);
        if (rc != CMF_OK) return bail(rc);
    }
    // CMF_LOOPBACK_STREAMS=1 turns every loopback group of the process into the stream-per-shard form (tests)
    g->loop_ms = tr == CMF_TR_LOOPBACK && transport == CMF_COMM_LOOPBACK_STREAMS;
    if (tr == CMF_TR_LOOPBACK && !g->loop_ms) // one device: every shard works on shard 0's streams, so the kernels of the loopback collectives are ordered
        for (cmf_handle_s *s : g->sh) { s->stream = g->sh[0]->stream; s->comm_stream = g->sh[0]->comm_stream; }
    if (tr == CMF_TR_RCCL) {
        int rc = rccl_load();
        if (rc != CMF_OK) return bail(rc);
        g->comm.assign((size_t)ndev, nullptr);
        (void)hipGetLastError(); // RCCL reports a stale (already handled) HIP error of this thread as its own
        ncclResult_t r_ = g_rccl.CommInitAll(g->comm.data(), ndev, devices);
        if (r_ != ncclSuccess) return bail(fail(CMF_ERR_COMM, "ncclCommInitAll failed: %s (RCCL from %s; if the process holds two HIP runtimes -- e.g. PyTorch imported after this library -- import torch first)", g_rccl.GetErrorString(r_), g_rccl.path.c_str()));
    }
    int rc = group_alloc_buffers(g);
    if (rc == CMF_OK) rc = group_finish_norm(g);
    // an enqueue worker per shard wherever every shard has its own stream (cmf_group.h); CMF_ENQUEUE_THREADS=0: the calling
    // thread enqueues all shards (option "enqueue_threads" switches later)
    const char *et = getenv("CMF_ENQUEUE_THREADS");
    if (rc == CMF_OK && !(et && atoi(et) == 0)) rc = group_start_workers(g);
    if (rc != CMF_OK) return bail(rc);
    *out = root;
    return CMF_OK;
}

// The overlap form's communication stream gets a communicator of its own (cmf_group.h, lane 1).  One process per shard:
// every rank calls this with the SAME second id (rank 0's cmf_comm_unique_id, handed over like the first); groups from
// cmf_create_multi create theirs themselves when the option is switched on and need not call it (id128 may be NULL).
int cmf_comm_init_overlap(cmf_handle h, const void *id128)
{
    if (!h) return fail(CMF_ERR_ARG, "handle is NULL");
    if (!h->group) return fail(CMF_ERR_STATE, "the handle belongs to no group");
    if (h->group->transport != CMF_TR_RCCL) return CMF_OK; // the other transports order their lanes with events
    if (!h->group->one_process && !id128) return fail(CMF_ERR_ARG, "id128 is NULL");
    return group_ensure_lane1(h->group, id128);
}

int cmf_comm_unique_id(void *id128)
{
    if (!id128) return fail(CMF_ERR_ARG, "id128 is NULL");
    CMFTRY(rccl_load());
    ncclUniqueId id;
    RCCLCHK(g_rccl.GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof(id));
    return CMF_OK;
}

static int comm_attach(cmf_handle_s *h, int nranks, int rank, int transport, const void *id128,
                       cmf_allreduce_fn ar, cmf_allgather_fn ag, void *user)
{
    if (!h) return fail(CMF_ERR_ARG, "handle is NULL");
    if (h->group) return fail(CMF_ERR_STATE, "the handle already belongs to a group");
    if (!h->sharded) return fail(CMF_ERR_STATE, "cmf_comm_init_* needs a handle from cmf_create_shard");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(CMF_ERR_ARG, "bad rank %d of %d", rank, nranks);
    const CmfDims &d = h->d;
    // the shard's place in the global partition must match what the other ranks assume
    std::vector<int64_t> t0, t1;
    group_partition(h->T_global, nranks, d.L, t0, t1);
    if (t0[(size_t)rank] != h->t_offset || t1[(size_t)rank] - t0[(size_t)rank] != d.Tl)
        return fail(CMF_ERR_ARG, "rank %d of %d must own columns [%lld, %lld) of T=%lld, the handle owns [%lld, %lld)", rank, nranks,
                    (long long)t0[(size_t)rank], (long long)t1[(size_t)rank], (long long)h->T_global,
                    (long long)h->t_offset, (long long)(h->t_offset + d.Tl));
    CMFTRY(group_use(h));
    cmf_group_s *g = new cmf_group_s();
    g->nranks = nranks;
    g->transport = transport;
    g->one_process = false;
    g->N = d.N; g->T = h->T_global; g->K = d.K; g->L = d.L;
    g->t0 = t0; g->t1 = t1;
    g->sh.push_back(h);
    g->rank.push_back(rank);
    g->ar_cb = ar; g->ag_cb = ag; g->cb_user = user;
    auto bail = [&](int rc) { (void)group_destroy(g); h->group = nullptr; return rc; };
    int rc = group_prepare_shard(h);
    if (rc != CMF_OK) return bail(rc);
    if (transport == CMF_TR_RCCL) {
        rc = rccl_load();
        if (rc != CMF_OK) return bail(rc);
        ncclUniqueId id;
        std::memcpy(&id, id128, sizeof(id));
        g->comm.assign(1, nullptr);
        (void)hipGetLastError(); // RCCL reports a stale (already handled) HIP error of this thread as its own
        ncclResult_t r_ = g_rccl.CommInitRank(&g->comm[0], nranks, id, rank);
        if (r_ != ncclSuccess) return bail(fail(CMF_ERR_COMM, "ncclCommInitRank failed: %s (RCCL from %s; if the process holds two HIP runtimes -- e.g. PyTorch imported after this library -- import torch first)", g_rccl.GetErrorString(r_), g_rccl.path.c_str()));
    }
    rc = group_alloc_buffers(g);
    if (rc == CMF_OK) rc = group_finish_norm(g);
    if (rc != CMF_OK) return bail(rc);
    h->group = g;
    if (h->factors_set) { // factors were set before the communicator existed: the neighbours' halos are still missing
        g->halos_current = false;
        g->halos_pending = false;
        g->halo_wide = false;
        set_est(h, 0);
    }
    return CMF_OK;
}

int cmf_comm_init_rccl(cmf_handle h, int nranks, int rank, const void *id128)
{
    if (!id128) return fail(CMF_ERR_ARG, "id128 is NULL");
    return comm_attach(h, nranks, rank, CMF_TR_RCCL, id128, nullptr, nullptr, nullptr);
}

int cmf_comm_init_callbacks(cmf_handle h, int nranks, int rank, cmf_allreduce_fn allreduce, cmf_allgather_fn allgather, void *user)
{
    if (!allreduce || !allgather) return fail(CMF_ERR_ARG, "NULL callback");
    return comm_attach(h, nranks, rank, CMF_TR_CALLBACKS, nullptr, allreduce, allgather, user);
}

int cmf_comm_info(cmf_handle h, char *buf, int64_t len)
{
    if (!h || !buf || len < 1) return fail(CMF_ERR_ARG, "bad argument");
    char tmp[1024];
    if (!h->group) {
        snprintf(tmp, sizeof(tmp), "transport=none nranks=1");
    } else {
        const cmf_group_s *g = h->group;
        std::string ranks;
        for (size_t i = 0; i < g->rank.size(); ++i) ranks += (i ? "," : "") + std::to_string(g->rank[i]) + "@dev" + std::to_string(g->sh[i]->device);
        const char *enq = g->pool.empty() ? "caller" : "threads"; // who enqueues the shards (cmf_group.h)
        const int lanes = (g->transport == CMF_TR_RCCL && g->comm2.size() == g->sh.size()) ? 2 : 1; // communicators per shard
        if (g->transport == CMF_TR_RCCL) {
            int v = 0;
            (void)g_rccl.GetVersion(&v);
            snprintf(tmp, sizeof(tmp), "transport=rccl version=%d lib=%s nranks=%d local=%zu ranks=%s overlap=%d enqueue=%s lanes=%d failed=%d", v,
                     g_rccl.path.c_str(), g->nranks, g->sh.size(), ranks.c_str(), (int)g->overlap, enq, lanes, (int)g->failed);
        } else {
            snprintf(tmp, sizeof(tmp), "transport=%s nranks=%d local=%zu ranks=%s overlap=%d enqueue=%s lanes=%d failed=%d",
                     g->transport == CMF_TR_LOOPBACK ? (g->loop_ms ? "loopback-streams" : "loopback") : (g->transport == CMF_TR_PEER ? "peer" : "callbacks"),
                     g->nranks, g->sh.size(), ranks.c_str(), (int)g->overlap, enq, lanes, (int)g->failed);
        }
    }
    snprintf(buf, (size_t)len, "%s", tmp);
    return CMF_OK;
}

int cmf_shard_bounds(cmf_handle h, int rank, int64_t *t0, int64_t *t1)
{
    if (!h || !t0 || !t1) return fail(CMF_ERR_ARG, "NULL argument");
    if (!h->group) {
        *t0 = h->t_offset;
        *t1 = h->t_offset + h->d.Tl;
        return CMF_OK;
    }
    if (rank < 0 || rank >= h->group->nranks) return fail(CMF_ERR_ARG, "rank %d out of range", rank);
    *t0 = h->group->t0[(size_t)rank];
    *t1 = h->group->t1[(size_t)rank];
    return CMF_OK;
}

// ---- PGD on T-sharded groups (pgd.jl:158-255 with data / est / H cut along T, W replicated) ------------------------------
// compute_gradW! (pgd.jl:206-214) is the C2 contraction: every shard contracts its own residual columns and ONE all-reduce
// sums the K x N x L partial gradients; the penalty, norm(grad) and the step are then identical replicated arithmetic.
// compute_gradH! (:218-221) is tensor_transconv! on the shard's columns plus its right residual halo (the conv covers
// Tl + halo_r columns, as in the MU rule); norm(gradH)^2, the per-component norms of UnitNormConstraint and the loss are
// sums over shards, combined in rank order on the host (exact: doubles travel as two 32-bit words).  The step-size state
// machine (stepW, stepH, cur_loss; :139-154, :248-253) is replicated: every rank takes the same decisions from the same sums.

// n doubles at device address ptr[i] of every local shard -> their sum over ALL ranks (rank order), written back to every shard
static int group_sum_doubles(cmf_group_s *g, const std::vector<double *> &ptr, int n, std::vector<double> *host_out = nullptr)
{
    const size_t nl = g->sh.size();
    std::vector<std::vector<double>> local(nl, std::vector<double>((size_t)n));
    for (size_t i = 0; i < nl; ++i) {
        cmf_handle_s *s = g->sh[i];
        CMFTRY(group_use(s));
        HIPCHK(hipMemcpyAsync(local[i].data(), ptr[i], (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    std::vector<double> total((size_t)n, 0.0);
    if (g->one_process) {
        for (int r = 0; r < g->nranks; ++r)
            for (size_t i = 0; i < nl; ++i)
                if (g->rank[i] == r)
                    for (int j = 0; j < n; ++j) total[(size_t)j] += local[i][(size_t)j];
    } else { // one all-gather of n doubles per rank, summed in rank order
        std::vector<double> all;
        CMFTRY(group_gather_doubles(g, local[0], all, n));
        for (int r = 0; r < g->nranks; ++r)
            for (int j = 0; j < n; ++j) total[(size_t)j] += all[(size_t)r * n + j];
    }
    for (size_t i = 0; i < nl; ++i) {
        cmf_handle_s *s = g->sh[i];
        CMFTRY(group_use(s));
        HIPCHK(hipMemcpyAsync(ptr[i], total.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream)); // `total` is pageable host memory: finish before it goes out of scope
    }
    if (host_out) *host_out = total;
    return CMF_OK;
}

static bool group_masked(const cmf_group_s *g) { return g->sh[0]->M != nullptr; }

// pgd.jl:245-253 on the group: the residual with the new factors on every shard, loss = sum over shards
static int group_pgd_finish(cmf_handle_s *st, cmf_group_s *g, double *step)
{
    std::vector<double *> ptr;
    for (cmf_handle_s *s : g->sh) {
        CMFTRY(group_use(s));
        CMFTRY(resid_and_loss(s, nullptr, group_masked(g), st->pgd_loss_abs != 0));
        ptr.push_back(s->d_scalar);
    }
    std::vector<double> tot;
    CMFTRY(group_sum_doubles(g, ptr, 1, &tot));
    const double loss = tot[0];
    *step *= (loss < st->pgd_cur_loss) ? 1.05 : 0.70;
    st->pgd_cur_loss = loss;
    return CMF_OK;
}

static int group_pgd_prepare(cmf_handle_s *st, cmf_group_s *g, int nonneg)
{
    if (nonneg < 0 || nonneg > 2) return fail(CMF_ERR_ARG, "constraint must be 0 (none), 1 (NonnegConstraint) or 2 (UnitNormConstraint)");
    if (g->gram || g->overlap) return fail(CMF_ERR_STATE, "the PGD rule runs on a group with the options gram and allreduce_overlap off");
    if (st->pgd_cur_loss < 0.0) st->pgd_cur_loss = g->data_norm; // pgd.jl:151 (the norm, not its square)
    for (cmf_handle_s *s : g->sh) {
        s->pgd_loss_abs = st->pgd_loss_abs;
        s->carry = CmfLossCarry{};
    }
    if (!g->halos_current) CMFTRY(group_exchange_halos(g));
    return CMF_OK;
}

int group_pgd_w(cmf_handle_s *st, cmf_group_s *g, double pen_sq, double pen_abs, int nonneg)
{
    GroupInline scope(g);
    CMFTRY(scope.rc);
    CMFTRY(group_pgd_prepare(st, g, nonneg));
    const CmfDims &d0 = g->sh[0]->d;
    const float gscale = st->pgd_loss_abs ? 1.f : 2.f; // pgd.jl:31-33 vs :42-44
    const size_t LKN = (size_t)d0.L * d0.K32 * d0.Np;
    for (cmf_handle_s *s : g->sh) {
        CMFTRY(group_use(s));
        CMFTRY(ensure_resid(s, group_masked(g), st->pgd_loss_abs != 0));                                 // pgd.jl:230 on the shard's columns
        CMFTRY(hxt_contract(s, s->est, s->est, 1, s->numden)); // pgd.jl:206-214, partial over t
    }
    CMFTRY(group_allreduce(g, g->red, 0, LKN)); // the one bulk exchange: K x N x L partial gradients
    for (cmf_handle_s *s : g->sh) {
        const CmfDims &d = s->d;
        CMFTRY(group_use(s));
        dim3 grid(d.Np / 64, d.KB, d.L);
        const int nblk = (d.Np / 64) * d.KB * d.L;
        if ((size_t)nblk > n_partial(s)) return fail(CMF_ERR_UNSUPPORTED, "PGD: partial buffer too small");
        hipLaunchKernelGGL(pgd_w_grad_kernel, grid, dim3(256), 0, s->stream, s->Wt, s->numden, s->numden + LKN, s->partial,
                           d.N, d.K, d.Np, d.K32, (float)pen_sq, (float)pen_abs, gscale);                // pgd.jl:231-234 (replicated)
        KCHK("pgd_w_grad_kernel");
        hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, s->stream, s->partial, nblk, s->d_scalar + 1, (double *)nullptr);
        KCHK("loss_reduce_kernel");
        hipLaunchKernelGGL(pgd_w_apply_kernel, grid, dim3(256), 0, s->stream, s->Wt, s->Wn, s->numden + LKN, s->d_scalar + 1,
                           d.N, d.K, d.Np, d.K32, (float)st->pgd_stepW, nonneg == 1);                    // pgd.jl:237-241
        KCHK("pgd_w_apply_kernel");
        if (nonneg == 2) CMFTRY(pgd_unit_norm(s, true));                                                 // pgd.jl:100-110 (W is replicated)
        set_est(s, 0);
    }
    return group_pgd_finish(st, g, &st->pgd_stepW);
}

int group_pgd_h(cmf_handle_s *st, cmf_group_s *g, double pen_sq, double pen_abs, int nonneg, double *loss)
{
    GroupInline scope(g);
    CMFTRY(scope.rc);
    CMFTRY(group_pgd_prepare(st, g, nonneg));
    const float gscale = st->pgd_loss_abs ? 1.f : 2.f;
    std::vector<double *> nrm;
    for (cmf_handle_s *s : g->sh) {
        const CmfDims &d = s->d;
        CMFTRY(group_use(s));
        if (!s->pgd_gradH) CMFTRY(dalloc_zero(&s->pgd_gradH, (size_t)d.Tl * d.K32));
        // the transposed residual on the shard's columns AND its right lag halo (transconv reads est[:, t .. t+L-1])
        if (s->est_kind == 2 + (s->M ? 1 : 0) + (st->pgd_loss_abs ? 2 : 0)) {
            // the shard's own columns are in est already (the conv that closed the W phase): transposed, not convolved again;
            // the <= L-1 halo columns are formed directly (resid_halo_kernel)
            hipLaunchKernelGGL(transpose_rows_kernel, dim3(d.Np / 64, (d.Tl + 63) / 64), dim3(256), 0, s->stream, s->est, s->estT, d.Tl, d.Np, d.TP, d.PADL);
            KCHK("transpose_rows_kernel");
            if (s->halo_r > 0) {
                hipLaunchKernelGGL(resid_halo_kernel, dim3(d.Np / 128, s->halo_r), dim3(128), 0, s->stream, s->Wt, s->H, s->XT, s->MT, s->estT,
                                   d.Tl, d.K, d.L, d.K32, d.Np, d.TP, d.PADL, st->pgd_loss_abs);
                KCHK("resid_halo_kernel");
            }
        } else {
            s->pgd_loss_abs_now = st->pgd_loss_abs;
            int rc_conv = s->MT ? launch_conv<7>(s, s->estT, d.Tl + s->halo_r, s->conv_gy_ext, s->XT)
                                : launch_conv<5>(s, s->estT, d.Tl + s->halo_r, s->conv_gy_ext, s->XT);
            s->pgd_loss_abs_now = 0;
            CMFTRY(rc_conv);
        }
        CMFTRY(launch_transconv(s, 1, s->estT));                                                         // pgd.jl:218-221
        dim3 grid((d.Tl + 63) / 64, d.KB);
        const int nblk = ((d.Tl + 63) / 64) * d.KB;
        if ((size_t)nblk > n_partial(s)) return fail(CMF_ERR_UNSUPPORTED, "PGD: partial buffer too small");
        hipLaunchKernelGGL(pgd_h_grad_kernel, grid, dim3(256), 0, s->stream, s->H, s->hslabs, s->tc_S1, s->pgd_gradH, s->partial,
                           d.Tl, d.K, d.K32, d.PADL, (float)pen_sq, (float)pen_abs, gscale);
        KCHK("pgd_h_grad_kernel");
        hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, s->stream, s->partial, nblk, s->d_scalar + 1, (double *)nullptr);
        KCHK("loss_reduce_kernel");
        nrm.push_back(s->d_scalar + 1);
    }
    CMFTRY(group_sum_doubles(g, nrm, 1)); // norm(gradH)^2 over all shards (pgd.jl:236)
    std::vector<double *> kn;
    for (cmf_handle_s *s : g->sh) {
        const CmfDims &d = s->d;
        CMFTRY(group_use(s));
        dim3 grid((d.Tl + 63) / 64, d.KB);
        hipLaunchKernelGGL(pgd_h_apply_kernel, grid, dim3(256), 0, s->stream, s->H, s->Ht, s->pgd_gradH, s->d_scalar + 1,
                           d.Tl, d.K, d.K32, d.PADL, d.TP, (float)st->pgd_stepH, nonneg == 1);
        KCHK("pgd_h_apply_kernel");
        if (nonneg == 2) { // pgd.jl:100-110: the norm of a component runs over all of T
            if (!s->pgd_knorm) CMFTRY(dalloc_zero(&s->pgd_knorm, (size_t)d.K32));
            hipLaunchKernelGGL(pgd_h_knorm_kernel, dim3(d.K), dim3(256), 0, s->stream, s->Ht, s->pgd_knorm, d.Tl, d.TP, d.PADL);
            KCHK("pgd_h_knorm_kernel");
            kn.push_back(s->pgd_knorm);
        }
        set_est(s, 0);
    }
    if (nonneg == 2) {
        CMFTRY(group_sum_doubles(g, kn, g->sh[0]->d.K));
        for (cmf_handle_s *s : g->sh) {
            const CmfDims &d = s->d;
            CMFTRY(group_use(s));
            hipLaunchKernelGGL(pgd_h_kscale_kernel, dim3(1024), dim3(256), 0, s->stream, s->H, s->Ht, s->pgd_knorm, d.Tl, d.K, d.K32, d.TP, d.PADL);
            KCHK("pgd_h_kscale_kernel");
        }
    }
    for (cmf_handle_s *s : g->sh) { // (an armed write-back: every shard's block of H is final here)
        CMFTRY(group_use(s));
        CMFTRY(wb_after_H(s));
    }
    CMFTRY(group_exchange_halos(g));
    CMFTRY(group_pgd_finish(st, g, &st->pgd_stepH));
    *loss = std::sqrt(st->pgd_cur_loss / (g->data_norm * g->data_norm)); // pgd.jl:201
    return CMF_OK;
}

// MaskedLoss on a group: the mask is cut like data -- shard r holds its own columns in both layouts and the right lag halo
// in the transposed one.  One process: `mask` is the whole N x T matrix; one process per shard: the block of data_local.
int group_set_mask(cmf_group_s *g, const double *mask)
{
    CMFTRY(group_sync(g));
    for (size_t i = 0; i < g->sh.size(); ++i) {
        cmf_handle_s *s = g->sh[i];
        const CmfDims &d = s->d;
        CMFTRY(group_use(s));
        set_est(s, 0);
        if (!mask) {
            if (s->M) (void)hipFree(s->M);
            if (s->MT) (void)hipFree(s->MT);
            s->M = s->MT = nullptr;
            continue;
        }
        const size_t TPNp = (size_t)d.TP * d.Np;
        if (!s->M) CMFTRY(dalloc_zero(&s->M, TPNp));
        if (!s->MT) CMFTRY(dalloc_zero(&s->MT, TPNp));
        const double *m = g->one_process ? mask + (size_t)g->t0[(size_t)g->rank[i]] * d.N : mask;
        CMFTRY(upload_cols(s, m, 0, d.Tl, true, false, s->M, s->MT));
        if (s->halo_r > 0) CMFTRY(upload_cols(s, m + (size_t)d.Tl * d.N, d.Tl, s->halo_r, false, false, s->M, s->MT));
    }
    return CMF_OK;
}

