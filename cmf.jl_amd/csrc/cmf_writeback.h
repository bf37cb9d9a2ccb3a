// cmf_writeback.h -- the factors written back into the caller's arrays behind a rule call (cmf_arm_writeback).
//
// The reference's rules mutate W and H in place (src/algs/mult.jl:37-38,51-52; alternating.jl:51-54 passes the same arrays to
// every call), so a caller that drives the rule call by call -- CMF.jl's own `fit` -- expects its arrays to hold the new
// factors when update_feature_maps! returns.  cmf_get_factors after every call does that with a synchronous fp64 download
// (23 MB at config 2: about 2 ms on top of a 5.3 ms iteration).  Here the download hides under the rule's own kernels:
//   * W is final when update_motifs! has run: at arm time the main stream's position is marked with an event and a COPY
//     STREAM brings W (fp32) into pinned host memory while the H phase's first contraction runs;
//   * H is final behind the H update kernel: the same, underneath the loss conv (mult.jl:55-57), 0.9 ms at config 2;
//   * a few helper threads widen fp32 -> fp64 into the caller's arrays as each copy lands (the conversion is exact, so the
//     arrays are bit for bit what cmf_get_factors returns), while the calling thread waits for the loss scalar.
// When K is a multiple of 32 the device layouts ARE Julia's order row by row (H[PADL + t][k], Wn[l][n][k]) and the copies are
// plain DMA transfers (no kernel shares the chip with the statically dealt contraction kernels); other K go through a small
// pack kernel on the copy stream.  The caller's pointers are used only between the hook behind the H update and the return
// of that same rule call.
// T-sharded groups (the reference's `fit` driving 8 GPUs through one handle): every shard copies its own column block of H the
// same way on its own device (shard 0 also W), and the helpers of the front handle widen block after block -- a synchronous
// cmf_get_factors per call (1.6 MB + a stream synchronisation per shard, one after the other, and 10 MB of W) would cost more
// than the 0.9 ms iteration it follows.
#pragma once
#include "cmf_workers.h"

#if defined(__SSE2__)
#include <emmintrin.h>
#endif

struct CmfWriteback {
    hipStream_t stream = nullptr;                    // the copy stream
    hipEvent_t ev_w_ready = nullptr, ev_h_ready = nullptr; // positions of the main stream behind which W / H are final
    hipEvent_t ev_w_done = nullptr, ev_h_done = nullptr;   // the copies have landed in pinned memory
    float *pin_W = nullptr, *pin_H = nullptr;        // pinned fp32 staging (Julia's element order)
    float *dev_stage = nullptr;                      // device staging of the pack kernels (K not a multiple of 32)
    size_t nW = 0, nH = 0;
    double *dst_W = nullptr, *dst_H = nullptr;       // the caller's arrays while armed
    bool armed = false, w_started = false, h_posted = false;
    bool has_copy = false;                           // stream / events / staging exist (a handle with device state of its own)
    // shards of a group: the hook behind a shard's H update runs on whoever enqueues that shard (its worker thread); it flags the
    // copy as issued, and the helpers of the group's front handle -- posted before the phase is enqueued -- wait for the flag
    // before they wait for the event (an event that has not been recorded yet reads as complete)
    std::atomic<bool> h_issued{false}, cancel{false};
    std::atomic<bool> poisoned{false};               // a drain ran out with a helper still inside a device wait: the helpers leave without widening, no further arms
    CmfWorkerPool pool;                              // the widening helpers (of a single handle, or of a group's front handle)
    int64_t armed_calls = 0, hooked_calls = 0;       // (cmf_get_counter: "writeback_calls", "writeback_overlapped")
};

// out[i] = (double)in[i]; streaming stores where the destination allows it (the caller reads the arrays later, the helper never)
static void cmf_widen(const float *in, double *out, size_t n)
{
    size_t i = 0;
#if defined(__SSE2__)
    while (i < n && (reinterpret_cast<uintptr_t>(out + i) & 15u)) { out[i] = (double)in[i]; ++i; }
    for (; i + 4 <= n; i += 4) {
        const __m128 v = _mm_loadu_ps(in + i);
        _mm_stream_pd(out + i, _mm_cvtps_pd(v));
        _mm_stream_pd(out + i + 2, _mm_cvtps_pd(_mm_movehl_ps(v, v)));
    }
    _mm_sfence();
#endif
    for (; i < n; ++i) out[i] = (double)in[i];
}
