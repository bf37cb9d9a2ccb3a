// cmf_api.hip -- the C ABI of include/cmf_hip.h for the MU rule, handles, launchers of the MU kernels, write-back, measurement.
// (round 6: one of three translation units -- cmf_rules.hip holds the HALS / Gram / PGD rules, cmf_groups.hip the T-sharded groups;
// cmf_internal.h what they share.)
#include "cmf_internal.h"

static void wb_free(cmf_handle_s *h);
static void wb_disarm(cmf_handle_s *h);

size_t n_partial(const cmf_handle_s *h)
{
    const CmfDims &d = h->d;
    size_t n = (size_t)(64 * h->conv_gx) * (size_t)std::max(h->conv_gy, h->conv_gy_ext); // conv loss partials (64 x 64 tiles, their quarters or sixteenths)
    n = std::max(n, (size_t)(d.Np / 64) * d.KB * d.L);                                 // PGD gradW norm partials
    n = std::max(n, (size_t)((d.Tl + 63) / 64) * d.KB);                                // PGD gradH norm partials
    n = std::max(n, 2 * (((size_t)d.Tl * d.K32 + 1023) / 1024));                       // Gram-form loss partials
    return n;
}

static int ensure_stage(cmf_handle_s *h, size_t elems)
{
    if (h->stage_elems >= elems) return CMF_OK;
    if (h->stage) HIPCHK(hipFree(h->stage));
    h->stage = nullptr;
    h->stage_elems = 0;
    HIPCHK(hipMalloc(&h->stage, elems * sizeof(double)));
    h->stage_elems = elems;
    return CMF_OK;
}

// A pool of non-blocking streams per device, shared by the handles of the process: creating a stream costs 1.4 ms and destroying one
// 1.1 ms on this runtime (tools/alloc_cost.hip) -- more than a whole 100-iteration fit of the reference's README example takes -- and
// init_rand, parameter_sweep and every fit_cnmf make and drop handles.  A released stream has been synchronised; at most 32 are kept.
static std::mutex g_stream_mu;
static std::map<int, std::vector<hipStream_t>> g_stream_pool;
hipError_t stream_acquire(int device, hipStream_t *s)
{
    {
        std::lock_guard<std::mutex> lock(g_stream_mu);
        auto &v = g_stream_pool[device];
        if (!v.empty()) { *s = v.back(); v.pop_back(); return hipSuccess; }
    }
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}
static void stream_release(int device, hipStream_t s, bool may_wait = true)
{
    if (!s) return;
    // (may_wait = false: a shard of a failed group -- its stream may never drain; hipStreamDestroy does not wait for it)
    if ((may_wait ? hipStreamSynchronize(s) : hipStreamQuery(s)) == hipSuccess) {
        std::lock_guard<std::mutex> lock(g_stream_mu);
        auto &v = g_stream_pool[device];
        if (v.size() < 32) { v.push_back(s); return; }
    }
    (void)hipStreamDestroy(s);
}


void plan(cmf_handle_s *h, int n_cu)
{
    const CmfDims &d = h->d;
    // C2 (hxt): lags per launch group = 2*LP; pick the LP that wastes the fewest padded lags
    int best = 1;
    int64_t bestP = 1 << 30;
    for (int lp : kHxtLP) {
        int64_t P = rup(d.L, 2 * lp);
        if (P < bestP || (P == bestP && lp > best)) { bestP = P; best = lp; }
    }
    h->hxt_LP = best;
    h->hxt_groups = (int)(bestP / (2 * best));
    const int wave_slots = 4 * n_cu * (best <= 5 ? 2 : 1); // resident waves (register-limited)
    // Time chunks: as many as fill the resident wave slots, each a whole number of ring rotations (6*LP rows).  When
    // leaving the < 6*LP rows behind the last whole rotation of Tl to the slab sum (CmfHxtTail) lets the chunks get one
    // rotation shorter, that is done: T = 6250 (the T/8 shard of config 2) over 8 chunks is 26 rotations + 10 rows, not 27.
    auto chunks = [&](int64_t waves_per_chunk, int *nchunks, int *chunk_len, int *main_rows) {
        const int unit = 6 * best;
        int nch = (int)std::max<int64_t>(1, (wave_slots + waves_per_chunk / 2) / waves_per_chunk);
        const int64_t units_all = (d.Tl + unit - 1) / unit, units_whole = d.Tl / unit;
        int64_t per = (units_all + nch - 1) / nch;
        *main_rows = d.Tl;
        if (units_whole > 0 && (units_whole + nch - 1) / nch < per) {
            per = (units_whole + nch - 1) / nch;
            *main_rows = (int)(units_whole * unit);
        }
        *chunk_len = (int)(per * unit);
        *nchunks = (int)((*main_rows + *chunk_len - 1) / *chunk_len);
    };
    chunks((int64_t)(d.Np / 32) * 2 * d.KB * h->hxt_groups, &h->hxt_nchunks, &h->hxt_chunk_len, &h->hxt_main);
    chunks((int64_t)(d.Np / 32) * d.KB * h->hxt_groups, &h->hxt_nchunks1, &h->hxt_chunk_len1, &h->hxt_main1);
    // C3 (transconv): the chunk units of all (t block, k block, source) pairs are dealt out evenly to the resident
    // waves (2 workgroups of 4 waves per CU); F = the largest number of waves that share one pair
    h->tc_LT = d.L <= 32 ? (int)rup(d.L, 4) : 32;
    auto tc_plan = [&](int nsrc, int *Wout, int *Fout, std::vector<int4> &tab, int extra_blocks) {
        const long long C = rup(d.N, 8) / 8;
        const long long pairs = (long long)((d.Tl + 127) / 128 + extra_blocks) * d.KB * nsrc;
        const long long U = pairs * C;
        // 2 workgroups of 4 waves per CU; 12 (the register limit with the LDS-DMA staging) measured the same
        // 1.73 ms at config 2 and costs more fragment slabs
        const long long W = std::min<long long>(8LL * n_cu, U);
        int F = 1;
        tab.resize((size_t)W);
        for (long long w = 0; w < W; ++w) {
            const long long u0 = w * U / W, u1 = (w + 1) * U / W;
            const long long pr = u0 / C;
            const long long first = tc_wave_of(pr * C, U, W); // first wave that touches this pair
            tab[(size_t)w] = make_int4((int)pr, (int)(u0 - pr * C), (int)(u1 - u0), (int)(w - first));
            F = std::max<int>(F, (int)(w - first) + 1);
        }
        // Placement (speed only): workgroup b lands on XCD b % 8 under round-robin dispatch.  Give each XCD the ranges
        // whose first chunk lies in one eighth of the n axis, so that the W rows it streams (the n range of its
        // waves, all lags) stay within its 4 MB L2 instead of every XCD cycling through all of W.
        if (W >= 64 && W % 32 == 0) {
            std::vector<int> order((size_t)W);
            for (long long w = 0; w < W; ++w) order[(size_t)w] = (int)w;
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return tab[(size_t)a].y < tab[(size_t)b].y; });
            std::vector<int4> placed((size_t)W);
            const long long per_xcd = W / 8, wgs_per_xcd = per_xcd / 4;
            for (long long x = 0; x < 8; ++x)
                for (long long m = 0; m < wgs_per_xcd; ++m)
                    for (long long j = 0; j < 4; ++j)
                        placed[(size_t)(4 * (x + 8 * m) + j)] = tab[(size_t)order[(size_t)(x * per_xcd + 4 * m + j)]];
            tab.swap(placed);
        }
        *Wout = (int)W;
        *Fout = F;
    };
    tc_plan(2, &h->tc_W, &h->tc_S, h->tc_tab_host[0], 0);
    tc_plan(1, &h->tc_W1, &h->tc_S1, h->tc_tab_host[1], 0);
    if (h->halo_ext) tc_plan(2, &h->tc_W2, &h->tc_S2, h->tc_tab_host[2], 1); // (one more 128-column block, in front of the own columns)
    h->tc_S_full = h->tc_S;
    h->tc_S1_full = h->tc_S1;
    // few components: J = L*K rows on the MFMA axes (cmf_small_k.h)
    h->small_k_ok = d.K <= 16 && d.L <= SK_MAXL;
    if (h->small_k_ok) {
        h->sk_J = d.L * d.K;
        const int mblocks = (h->sk_J + 31) / 32;              // 32-row blocks that hold rows j
        const int c2max = SK_MAXMBW_C2;
        h->sk_MG = (mblocks + c2max - 1) / c2max;             // groups of at most SK_MAXMBW_C2 blocks ...
        h->sk_MBW = (mblocks + h->sk_MG - 1) / h->sk_MG;      // ... as even as possible: the least padding
        h->sk_JP = 32 * h->sk_MBW * h->sk_MG;
        // a last block of at most SK_RVT live rows goes to the VALU (one row group of 2-4 blocks: the kernel variants that exist)
        h->sk_RV = 0;
        if (h->sk_MG == 1 && h->sk_MBW >= 2 && h->sk_MBW <= 4 && h->sk_J % 32 >= 1 && h->sk_J % 32 <= SK_RVT) {
            h->sk_RV = h->sk_J % 32;
            h->sk_MBW -= 1; // (sk_JP keeps the padded row count: the slabs' rows)
        }
        h->sk_TG = (int)rup(d.Tl + d.L - 1, 128);
        // C2: a wave = (n block, m group, source, time chunk).  Chunks of about 512 rows (four strips): many short waves, so that
        // the rounds of the launch are short and its last one costs little -- but at least as many waves as are resident; chunks are whole
        // 16-row rounds, 4 chunks per workgroup.
        const int64_t per_chunk = (int64_t)(d.Np / 32) * h->sk_MG * 2;
        // two waves per SIMD, or one when the launch is short (under ~800 MFMAs per SIMD: the per-wave prologue -- strip, X ring -- and the
        // chunk reduction weigh more than the second wave's latency hiding; measured 250 x 6250 .. 50000, K = 5 and 12:
        // profiles/r05_c2_dma_strips.txt).  More, shorter chunks (12, 16 per CU) lose 3-15 % everywhere.
        const int64_t mfma_per_simd = (int64_t)(d.Np / 32) * h->sk_MG * 2 * ((d.Tl + 15) / 16) * 8 * h->sk_MBW / (4LL * n_cu);
        const int64_t waves_per_cu = mfma_per_simd < 800 ? 4 : 8;
        int64_t nch = std::max<int64_t>({(int64_t)4, (waves_per_cu * n_cu + per_chunk - 1) / per_chunk, (int64_t)(d.Tl + 511) / 512});
        h->sk_chunk_len = (int)std::max<int64_t>(16, rup((d.Tl + nch - 1) / nch, 16));
        nch = (d.Tl + h->sk_chunk_len - 1) / h->sk_chunk_len;
        h->sk_ngroups = (int)((nch + 3) / 4);
        // C3: the G GEMM runs one wave per 32 columns, row group and source; with fewer than two waves per CU (short
        // recordings: BASELINE configs[0] has T = 2000) the general kernel, which splits the reduction over n, is faster
        // (measured at configs[0]: 20 us against 39)
        // Its row groups hold whole components (the lag sum of an output is folded inside one workgroup): Kg components of L rows
        // each in at most SK_MAXMBW blocks of 32 rows.
        h->sk3_MG = (d.K * d.L + 32 * SK_MAXMBW - 1) / (32 * SK_MAXMBW);
        for (;; ++h->sk3_MG) {
            h->sk3_Kg = (d.K + h->sk3_MG - 1) / h->sk3_MG;
            if (h->sk3_Kg * d.L <= 32 * SK_MAXMBW) break;
        }
        h->sk3_MG = (d.K + h->sk3_Kg - 1) / h->sk3_Kg;
        h->sk3_MBW = (h->sk3_Kg * d.L + 31) / 32;
        h->sk3_GR = 32 * h->sk3_MBW;
        h->sk3_JP = h->sk3_GR * h->sk3_MG;
        // (as in C2: a last block of at most SK_RVT live rows goes to the VALU; their Wj columns lie in the fold tile's LDS during the main loop)
        h->sk3_RV = 0;
        if (h->sk3_MG == 1 && h->sk3_MBW >= 2 && h->sk3_MBW <= 4 && (h->sk3_Kg * d.L) % 32 >= 1 && (h->sk3_Kg * d.L) % 32 <= SK_RVT &&
            (8 * ((rup(d.N, 2) + 7) / 8) + 2) * 4 <= 32 * SK_TILE_STRIDE) {
            h->sk3_RV = (h->sk3_Kg * d.L) % 32;
            h->sk3_MBW -= 1;
        }
        // Short recordings (BASELINE configs[0]: T = 2000 is 128 waves for 1024 SIMDs): the reduction over n is cut into up to 8 pieces,
        // each a workgroup row of its own writing its own pair of slabs (h_update adds the slabs in order: deterministic), until there
        // are two waves per SIMD; a piece keeps at least 8 rounds of 8 rows (measured over T = 1000 .. 25000: profiles/r05_small_k_n_split.txt).
        {
            const int64_t waves = (int64_t)(h->sk_TG / 32) * h->sk3_MG * 2;
            const int rounds = (int)((rup(d.N, 2) + 7) / 8);
            int ns = (int)std::min<int64_t>({(int64_t)8, (8LL * n_cu + waves - 1) / waves, (int64_t)std::max(1, rounds / 8)});
            ns = std::max(ns, 1);
            h->sk3_RPS = (rounds + ns - 1) / ns;
            h->sk3_NS = (rounds + h->sk3_RPS - 1) / h->sk3_RPS; // (no empty piece)
            h->sk_tc_ok = waves * h->sk3_NS >= 2LL * n_cu;
        }
    }
    // C1 (conv)
    h->conv_gx = d.Np / 128;
    h->conv_gy = (d.Tl + 127) / 128;
    h->conv_gy_ext = (d.Tl + h->halo_r + 127) / 128;
}

static void destroy_impl(cmf_handle_s *h)
{
    if (!h) return;
    if (!h->root_only) (void)hipSetDevice(h->device);
    wb_free(h);
    if (h->root_only) { delete h; return; }
    (void)hipSetDevice(h->device);
    auto mine = [&](const void *q) { // not a piece of the arena
        return q && !(h->arena && static_cast<const char *>(q) >= static_cast<const char *>(h->arena) &&
                      static_cast<const char *>(q) < static_cast<const char *>(h->arena) + h->arena_bytes);
    };
    for (float *q : {h->sk_slabs, h->sk_Wj})
        if (mine(q)) (void)hipFree(q);
    if (mine(h->sk_cnt)) (void)hipFree(h->sk_cnt);
    (void)hipSetDevice(h->device);
    float *fbufs[] = {h->H, h->Ht, h->Wt, h->Wn, h->X, h->XT, h->est, h->estT, h->wslabs, h->numden_own, h->hslabs,
                      h->halo_own[0], h->halo_own[1], h->halo_own[2], h->halo_own[3],
                      h->gram_numden_h, h->pgd_gradH, h->M, h->MT, h->hals_snap, h->hals_HX, h->hals_cslabs, h->hals_C, h->hals_HH, h->hals_PT, h->hals_D, h->hals_PW, h->hals_GW, h->hals_GE, h->hals_GWt};
    for (float *p : fbufs)
        if (mine(p)) (void)hipFree(p);
    for (int v = 0; v < 3; ++v)
        if (h->tc_tab[v]) (void)hipFree(h->tc_tab[v]);
    if (mine(h->partial)) (void)hipFree(h->partial);
    for (hipStream_t st : {h->hals_sA, h->hals_sB})
        if (st) { if (!h->streams_may_hang) (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (hipEvent_t e : h->hals_ev)
        if (e) (void)hipEventDestroy(e);
    if (h->hals_flags) (void)hipFree(h->hals_flags);
    if (h->hals_status) (void)hipHostFree(h->hals_status);
    if (h->pgd_knorm) (void)hipFree(h->pgd_knorm);
    if (mine(h->d_scalar_own)) (void)hipFree(h->d_scalar_own);
    if (h->arena) (void)hipFree(h->arena);
    if (h->h_scalar) (void)hipHostFree(h->h_scalar);
    if (h->stage) (void)hipFree(h->stage);
    for (auto &r : h->prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (hipEvent_t e : h->prof_pool) (void)hipEventDestroy(e);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev_c0) (void)hipEventDestroy(h->ev_c0);
    if (h->ev_c1) (void)hipEventDestroy(h->ev_c1);
    if (h->h_ring) (void)hipHostFree(h->h_ring);
    stream_release(h->device, h->own_comm_stream, !h->streams_may_hang);
    stream_release(h->device, h->own_stream, !h->streams_may_hang);
    delete h;
}

// upload `ncols` columns (fp64, N x ncols column-major) starting at local column tc
int upload_cols(cmf_handle_s *h, const double *src, int64_t tc, int64_t ncols, bool rows_layout, bool accumulate_sumsq,
                       float *rows_dst, float *cols_dst)
{
    if (!rows_dst) rows_dst = h->X;
    if (!cols_dst) cols_dst = h->XT;
    const CmfDims &d = h->d;
    const int64_t chunk = std::max<int64_t>(32, (int64_t)(32u << 20) / std::max<int64_t>(1, d.N) / 8 / 32 * 32); // ~32 MiB
    CMFTRY(ensure_stage(h, (size_t)std::min(chunk, rup(ncols, 32)) * d.N + 1024));
    for (int64_t c0 = 0; c0 < ncols; c0 += chunk) {
        int64_t nc = std::min(chunk, ncols - c0);
        HIPCHK(hipMemcpyAsync(h->stage, src + (size_t)c0 * d.N, (size_t)nc * d.N * sizeof(double), hipMemcpyHostToDevice, h->stream));
        dim3 grid((d.N + 31) / 32, (unsigned)((nc + 31) / 32)), block(32, 8);
        hipLaunchKernelGGL(pack_cols_kernel, grid, block, 0, h->stream, h->stage, d.N, (int)(tc + c0), (int)nc,
                           rows_layout ? rows_dst : nullptr, cols_dst, d.Np, d.TP, d.PADL);
        KCHK("pack_cols_kernel");
        if (accumulate_sumsq) {
            double *acc = h->stage + (size_t)nc * d.N; // 256 spare doubles
            hipLaunchKernelGGL(sumsq_f64_kernel, dim3(256), dim3(256), 0, h->stream, h->stage, (size_t)nc * d.N, acc);
            KCHK("sumsq_f64_kernel");
            std::vector<double> part(256);
            HIPCHK(hipMemcpyAsync(part.data(), acc, 256 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            for (double v : part) h->data_sumsq += v;
        } else {
            HIPCHK(hipStreamSynchronize(h->stream));
        }
    }
    return CMF_OK;
}

int create_impl(cmf_handle *out, int device, int64_t N, int64_t Tl, int64_t K, int64_t L,
                       const double *data, int64_t t_offset, int64_t T_global, bool sharded)
{
    if (!out) return fail(CMF_ERR_ARG, "handle pointer is NULL");
    *out = nullptr;
    if (N < 1 || Tl < 1 || K < 1 || L < 1) return fail(CMF_ERR_ARG, "N, T, K, L must all be >= 1 (got N=%lld T=%lld K=%lld L=%lld)",
                                                       (long long)N, (long long)Tl, (long long)K, (long long)L);
    if (t_offset < 0 || t_offset + Tl > T_global) return fail(CMF_ERR_ARG, "shard [%lld, %lld) is outside [0, %lld)",
                                                               (long long)t_offset, (long long)(t_offset + Tl), (long long)T_global);
    if (N > (1 << 24) || Tl > (1 << 28) || K > 4096 || L > 4096) return fail(CMF_ERR_UNSUPPORTED, "problem size out of range");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(CMF_ERR_HIP, "no HIP device available (libcmf_hip has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(CMF_ERR_ARG, "device %d out of range (0..%d)", device, ndev - 1);
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));

    cmf_handle_s *h = new cmf_handle_s();
    h->device = device;
    h->t_offset = t_offset;
    h->T_global = T_global;
    h->sharded = sharded;
    h->has_left = t_offset > 0;
    h->halo_r = (int)std::min<int64_t>(L - 1, T_global - t_offset - Tl);
    CmfDims &d = h->d;
    d.N = (int)N; d.Tl = (int)Tl; d.K = (int)K; d.L = (int)L;
    d.Np = (int)rup(N, 128);
    d.KB = (int)((K + 31) / 32);
    d.K32 = 32 * d.KB;
    d.PADL = (int)rup(L - 1, 32) + 32;
    // A shard with a left neighbour may carry the halo of H in the W-phase all-reduce (cmf_groups.hip): it then updates the L-1 columns
    // in front of its own itself -- one conv tile row at columns [-64, 0) and one transconv block at [-128, 0), from an H that is valid
    // 2(L-1) columns out -- which needs 128 columns of padding in front of every time axis (K a multiple of 32: the one-wave conv tiles)
    h->halo_ext = (sharded && t_offset > 0 && K % 32 == 0 && L - 1 >= 1 && L - 1 <= 64) ? (int)(L - 1) : 0;
    if (h->halo_ext) d.PADL += 64;
    d.TP = d.PADL + (int)rup(Tl + L, 512) + 256;
    d.Lp = L <= 32 ? (int)rup(L, 4) : (int)rup(L, 32);
    if ((double)d.Lp * d.Np * d.K32 * 4.0 >= 2147483648.0 || (double)d.TP * d.K32 * 4.0 >= 2147483648.0 ||
        64.0 * d.TP * 4.0 >= 2147483648.0 || 64.0 * d.Np * 4.0 >= 2147483648.0) {
        delete h;
        return fail(CMF_ERR_UNSUPPORTED, "W (L*N*K), H (T*K) or a 64-row block of est (64*T or 64*N) exceeds the 2 GiB the "
                                         "kernels' 32-bit buffer offsets address");
    }
    h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    plan(h, h->n_cu);
    // the C2 kernel addresses a time chunk of X with 32-bit byte offsets: keep chunks below 2 GiB
    while ((double)(h->hxt_chunk_len + 16 * h->hxt_LP + 8) * d.Np * 4.0 >= 2147483648.0) {
        h->hxt_chunk_len = (int)rup(h->hxt_chunk_len / 2, 6 * h->hxt_LP);
        h->hxt_main = d.Tl;
        h->hxt_nchunks = (d.Tl + h->hxt_chunk_len - 1) / h->hxt_chunk_len;
    }
    while ((double)(h->hxt_chunk_len1 + 16 * h->hxt_LP + 8) * d.Np * 4.0 >= 2147483648.0) {
        h->hxt_chunk_len1 = (int)rup(h->hxt_chunk_len1 / 2, 6 * h->hxt_LP);
        h->hxt_main1 = d.Tl;
        h->hxt_nchunks1 = (d.Tl + h->hxt_chunk_len1 - 1) / h->hxt_chunk_len1;
    }

    auto bail = [&](int rc) { destroy_impl(h); return rc; };
#define TRYB(expr) do { int rc__ = (expr); if (rc__ != CMF_OK) return bail(rc__); } while (0)
#define HIPB(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return bail(fail(CMF_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e__))); } while (0)
    HIPB(stream_acquire(h->device, &h->own_stream));
    h->stream = h->own_stream;
    HIPB(hipEventCreate(&h->ev0));
    HIPB(hipEventCreate(&h->ev1));
    // the buffers a handle has, zero-filled: the small ones (under 32 MB each) as ONE allocation in 256-byte granules -- on a small
    // problem that is all of them, and creating and destroying the handle costs less than a millisecond instead of seven --, the large
    // ones on their own (one 2.5 GB allocation for config 2 took 10 ms longer than its parts)
    struct Want { void **pp; size_t bytes; };
    std::vector<Want> wants;
    int want_rc = CMF_OK;
    // (test hook CMF_TEST_NO_ARENA=1: every buffer an allocation of its own, so that an overrun of one cannot land silently in a live
    // neighbour -- tests/test_small_k.py runs once in that form)
    const bool no_arena = test_hook("CMF_TEST_NO_ARENA", 0) == 1;
    auto want = [&](auto **pp, size_t n) {
        if (no_arena || n * sizeof(**pp) >= ((size_t)32 << 20)) { if (want_rc == CMF_OK) want_rc = dalloc_zero(pp, std::max<size_t>(n, 1)); }
        else wants.push_back({reinterpret_cast<void **>(pp), n * sizeof(**pp)});
    };
    const size_t TPNp = (size_t)d.TP * d.Np;
    want(&h->H, (size_t)d.TP * d.K32);
    want(&h->Ht, (size_t)d.K32 * d.TP);
    want(&h->Wt, (size_t)d.Lp * d.K32 * d.Np);
    want(&h->Wn, (size_t)d.Lp * d.Np * d.K32);
    want(&h->X, TPNp);
    want(&h->XT, TPNp);
    want(&h->est, TPNp);
    want(&h->estT, TPNp);
    want(&h->wslabs, (size_t)std::max(2 * hxt_nslabs(h->hxt_nchunks), hxt_nslabs(h->hxt_nchunks1)) * d.L * d.K32 * d.Np);
    want(&h->numden_own, (size_t)2 * d.L * d.K32 * d.Np);
    want(&h->hslabs, std::max((size_t)std::max(2 * std::max(h->tc_S, 2 * h->sk3_NS), std::max(h->tc_S1, 2 * h->sk3_NS)) * d.Tl * d.K32,
                              h->halo_ext ? (size_t)2 * h->tc_S2 * (d.Tl + 128) * d.K32 : (size_t)0)); // (the few-component C3 writes 2 slabs per piece of its reduction)
    if (h->small_k_ok) {
        want(&h->sk_slabs, (size_t)h->sk_ngroups * 2 * h->sk_JP * d.Np);
        want(&h->sk_Wj, (size_t)d.Np * h->sk3_JP);
        want(&h->sk_cnt, (size_t)h->sk_TG / 128 + 4);
        h->small_k = true; // (option "small_k": 0 = the general kernels for every K)
        h->sk_tc = h->small_k && h->sk_tc_ok;
        if (h->sk_tc) h->tc_S = h->tc_S1 = 2 * h->sk3_NS;
    }
    for (int v = 0; v < 3; ++v) {
        if (h->tc_tab_host[v].empty()) continue;
        HIPB(hipMalloc(&h->tc_tab[v], h->tc_tab_host[v].size() * sizeof(int4)));
        HIPB(hipMemcpy(h->tc_tab[v], h->tc_tab_host[v].data(), h->tc_tab_host[v].size() * sizeof(int4), hipMemcpyHostToDevice));
    }
    for (int w = 0; w < 4; ++w) {
        want(&h->halo_own[w], (size_t)std::max(1, d.L - 1) * d.K32);
    }
    want(&h->partial, n_partial(h));
    want(&h->d_scalar_own, 4);
    TRYB(want_rc);
    {
        size_t total = 0;
        for (const Want &w : wants) total += (w.bytes + 255) & ~(size_t)255;
        HIPB(hipMalloc(&h->arena, total));
        h->arena_bytes = total;
        // (hipMemset of device memory runs on the null stream and may return before it has finished; the handle's work runs on
        // non-blocking streams, which the null stream does not order)
        HIPB(hipMemset(h->arena, 0, total));
        HIPB(hipStreamSynchronize(nullptr));
        size_t off = 0;
        for (const Want &w : wants) {
            *w.pp = static_cast<char *>(h->arena) + off;
            off += (w.bytes + 255) & ~(size_t)255;
        }
        h->numden = h->numden_own;
        for (int w = 0; w < 4; ++w) h->halo[w] = h->halo_own[w];
        h->d_scalar = h->d_scalar_own;
    }
    HIPB(hipHostMalloc(&h->h_scalar, 4 * sizeof(double)));
    if (data) {
        TRYB(upload_cols(h, data, 0, Tl, true, true));
        if (h->halo_r > 0) TRYB(upload_cols(h, data + (size_t)Tl * N, Tl, h->halo_r, false, false));
        h->data_norm = std::sqrt(h->data_sumsq);
        h->have_data = true;
    }
#undef TRYB
#undef HIPB
    *out = h;
    return CMF_OK;
}

// ------------------------------------------------------------------------------------------
// kernel launchers
// ------------------------------------------------------------------------------------------

int launch_hxt_on(cmf_handle_s *h, const float *X0, const float *X1, int NpX, int nsrc, float *slabs, int nchunks, int chunk_len,
                         int main_rows)
{
    ProfScope prof_(h, (nsrc == 2 && X0 == h->X) ? PROF_HXT : (nsrc == 1 && X0 == h->X) ? PROF_HXT_NUM : (nsrc == 1 && X0 == h->est && h->est_kind == 1) ? PROF_HXT_DEN
                          : (nsrc == 1 && X0 == h->est) ? PROF_HXT_RESID : (X0 == h->hals_HX && h->hals_HX) ? PROF_HXT_HH : PROF_OTHER);
    const CmfDims &d = h->d;
    h->spec_gen = -1; // (the slabs of a speculated contraction are being overwritten)
    HxtParams p;
    p.H = h->H; p.X0 = X0; p.X1 = X1; p.slabs = slabs;
    p.Np = NpX; p.K32 = d.K32; p.KB = d.KB; p.PADL = d.PADL; p.L = d.L; p.Tl = main_rows >= 0 ? main_rows : d.Tl; p.chunk_len = chunk_len;
    p.G = h->hxt_groups; p.nsrc = nsrc;
    p.CG = hxt_cg(nchunks);
    dim3 grid((NpX / 128) * p.CG * h->hxt_groups, nchunks / p.CG, nsrc * d.KB), block(256);
    switch (h->hxt_LP) {
#define CASE(LP_) case LP_: hipLaunchKernelGGL((hxt_kernel<LP_>), grid, block, 0, h->stream, p); break;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(8)
#undef CASE
    default: return fail(CMF_ERR_STATE, "internal: bad hxt LP %d", h->hxt_LP);
    }
    KCHK("hxt_kernel");
    return CMF_OK;
}

static int launch_hxt(cmf_handle_s *h)
{
    return launch_hxt_on(h, h->X, h->est, h->d.Np, 2, h->wslabs, h->hxt_nchunks, h->hxt_chunk_len, h->hxt_main);
}

int launch_transconv(cmf_handle_s *h, int nsrc, const float *xt0, bool front_block)
{
    if (h->sk_tc) return launch_transconv_small(h, nsrc, xt0);
    ProfScope prof_(h, nsrc == 2 ? PROF_TRANSCONV : PROF_TRANSCONV_1);
    const CmfDims &d = h->d;
    TcParams p;
    p.Wn = h->Wn; p.XT0 = xt0 ? xt0 : h->XT; p.XT1 = h->estT; p.slabs = h->hslabs;
    p.NpW = d.Np;
    p.TP = d.TP; p.PADL = d.PADL; p.K32 = d.K32; p.KB = d.KB; p.L = d.L; p.Tl = d.Tl;
    p.nsrc = nsrc;
    p.C = (int)(rup(d.N, 8) / 8); // rows >= roundup(N, 8) are zero: stop at the last 8-row chunk that holds data
    p.W = nsrc == 2 ? h->tc_W : h->tc_W1;
    p.F = nsrc == 2 ? h->tc_S : h->tc_S1;
    p.wtab = h->tc_tab[nsrc == 2 ? 0 : 1];
    p.t_first = 0;
    p.slab_rows = d.Tl;
    if (front_block) { // (two sources; a shard that also updates the columns in front of its own: h_update_impl)
        p.W = h->tc_W2; p.F = h->tc_S2; p.wtab = h->tc_tab[2];
        p.t_first = -128;
        p.slab_rows = d.Tl + 128;
    }
    dim3 grid((p.W + 3) / 4), block(256);
    switch (h->tc_LT) {
#define CASE(LT_) case LT_: hipLaunchKernelGGL((transconv_kernel<LT_>), grid, block, 0, h->stream, p); break;
        CASE(4) CASE(8) CASE(12) CASE(16) CASE(20) CASE(24) CASE(28) CASE(32)
#undef CASE
    default: return fail(CMF_ERR_STATE, "internal: bad transconv LT %d", h->tc_LT);
    }
    KCHK("transconv_kernel");
    return CMF_OK;
}

int launch_slab_sum(cmf_handle_s *h, float *out, const float *in, int nslabs, size_t stride, bool take_carry,
                           CmfHxtTail tail)
{
    size_t n4 = stride / 4;
    int blocks = (int)std::min<size_t>(2048, (n4 + 255) / 256);
    CmfLossCarry carry{};
    if (take_carry && h->carry.partial) { // a loss reduction deferred by cmf_iterate rides on this launch
        carry = h->carry;
        h->carry = CmfLossCarry{};
    }
    hipLaunchKernelGGL(slab_sum_kernel, dim3(blocks + (carry.partial ? 1 : 0)), dim3(256), 0, h->stream, out, in, nslabs, stride, n4, carry, tail);
    KCHK("slab_sum_kernel");
    return CMF_OK;
}

// The C2 contraction of one (nsrc = 1: X0) or two sources with H_shift, complete: kernel, slab sum, and the rows the kernel
// leaves to the slab sum.   out: [nsrc][L][K32][Np]
int hxt_contract(cmf_handle_s *h, const float *X0, const float *X1, int nsrc, float *out, bool take_carry, bool slabs_only)
{
    const CmfDims &d = h->d;
    h->spec_gen = -1; // (the slabs / sums of a speculated contraction are being overwritten)
    if (h->small_k) return hxt_contract_small(h, X0, X1, nsrc, out, take_carry, slabs_only); // few components (cmf_small.hip)
    const int nch = nsrc == 2 ? h->hxt_nchunks : h->hxt_nchunks1, clen = nsrc == 2 ? h->hxt_chunk_len : h->hxt_chunk_len1;
    const int main_rows = nsrc == 2 ? h->hxt_main : h->hxt_main1;
    CMFTRY(launch_hxt_on(h, X0, X1, d.Np, nsrc, h->wslabs, nch, clen, main_rows));
    const CmfHxtTail tail{h->H, X0, X1, d.Tl - main_rows, d.PADL + main_rows, d.L, d.K32, d.Np};
    return launch_slab_sum(h, out, h->wslabs, hxt_nslabs(nch), (size_t)nsrc * d.L * d.K32 * d.Np, take_carry, tail);
}

int read_scalar(cmf_handle_s *h, int slot, double *v)
{
    HIPCHK(hipMemcpyAsync(h->h_scalar + slot, h->d_scalar + slot, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *v = h->h_scalar[slot];
    return CMF_OK;
}

// ------------------------------------------------------------------------------------------
// phases
// ------------------------------------------------------------------------------------------
static int check_ready(cmf_handle_s *h, bool need_data)
{
    if (!h) return fail(CMF_ERR_ARG, "handle is NULL");
    if (h->root_only) return fail(CMF_ERR_STATE, "this entry needs a single-GPU handle, not the front handle of a cmf_create_multi group");
    HIPCHK(hipSetDevice(h->device));
    if (!h->factors_set) return fail(CMF_ERR_STATE, "factors not set: call cmf_set_factors first");
    if (need_data && !h->have_data) return fail(CMF_ERR_STATE, "handle was created without data");
    return CMF_OK;
}

// The C2 contraction of update_motifs! has already been enqueued for exactly this state (w_speculate, behind the loss conv of the
// update_feature_maps! before): est is current and nothing has touched H, W, est or the slabs since.
static bool w_speculated(cmf_handle_s *h)
{
    const bool hit = h->spec_gen >= 0 && h->spec_gen == h->est_gen && h->reuse_est && h->est_kind == 1 && !h->carry.partial;
    h->spec_gen = -1;
    if (hit) h->spec_hits += 1;
    return hit;
}

int w_partial_impl(cmf_handle_s *h)
{
    const CmfDims &d = h->d;
    if (w_speculated(h)) return CMF_OK;
    if (!(h->reuse_est && h->est_kind == 1))
        CMFTRY(launch_conv<0>(h, h->est, d.Tl, h->conv_gy)); // mult.jl:28 (skipped when est is still current)
    set_est(h, 1);
    return hxt_contract(h, h->X, h->est, 2, h->numden, true); // mult.jl:31-34
}

// The two halves of w_partial_impl as separate steps (same arithmetic, the sources contracted one at a time): the
// numerator needs H only, so a sharded host can compute and all-reduce it while the loss conv and the denominator
// contraction are still running.
int w_partial_half_impl(cmf_handle_s *h, int den)
{
    const CmfDims &d = h->d;
    const size_t LKN = (size_t)d.L * d.K32 * d.Np;
    if (den) {
        if (!(h->reuse_est && h->est_kind == 1))
            CMFTRY(launch_conv<0>(h, h->est, d.Tl, h->conv_gy)); // mult.jl:28 (skipped when est is still current)
        set_est(h, 1);
    }
    const float *src = den ? h->est : h->X;
    return hxt_contract(h, src, src, 1, h->numden + (den ? LKN : 0), den != 0); // mult.jl:31-34, one source
}

// den == NULL: denomW lies behind numW in h->numden (the layout of the [numW | denomW] all-reduce buffer)
static int w_apply_impl_(cmf_handle_s *h, double l1W, double l2W, const float *tail_src, float *tail_dst, int tail_n, const float *den);
int w_apply_impl(cmf_handle_s *h, double l1W, double l2W, const float *tail_src, float *tail_dst, int tail_n,
                        const float *den)
{
    return w_apply_impl_(h, l1W, l2W, tail_src, tail_dst, tail_n, den);
}

// update_motifs! of the MU rule on a single handle (mult.jl:23-39).  Few components: conv (unless est is current), the C2
// kernel, and ONE launch that sums its slabs, updates W and packs the C3 operand (w_update_small_kernel).
static int w_phase_impl(cmf_handle_s *h, double l1W, double l2W)
{
    const CmfDims &d = h->d;
    if (!h->small_k) {
        CMFTRY(w_partial_impl(h));
        return w_apply_impl(h, l1W, l2W);
    }
    if (!w_speculated(h)) {
        if (!(h->reuse_est && h->est_kind == 1))
            CMFTRY(launch_conv<0>(h, h->est, d.Tl, h->conv_gy)); // mult.jl:28 (skipped when est is still current)
        set_est(h, 1);
        CMFTRY(hxt_contract(h, h->X, h->est, 2, nullptr, false, true)); // mult.jl:31-34: the slabs only
    }
    CmfLossCarry carry{};
    if (h->carry.partial) { // a loss reduction deferred by cmf_iterate rides on this launch
        carry = h->carry;
        h->carry = CmfLossCarry{};
    }
    hipLaunchKernelGGL(w_update_small_kernel, dim3(d.Np / 64 + 1, d.L, d.K), dim3(256), 0, h->stream, h->Wt, h->Wn, h->sk_tc ? h->sk_Wj : nullptr, h->sk_slabs,
                       h->sk_ngroups, d.N, d.K, d.L, d.Np, d.K32, h->sk_JP, h->sk3_Kg, h->sk3_GR, h->sk3_JP,
                       (float)l1W, (float)(2.0 * l2W), carry); // mult.jl:37-38
    KCHK("w_update_small_kernel");
    set_est(h, 0);
    h->sk_wj_gen = h->sk_tc ? h->est_gen : -1;
    return CMF_OK;
}

static int w_apply_impl_(cmf_handle_s *h, double l1W, double l2W, const float *tail_src, float *tail_dst, int tail_n, const float *den)
{
    const CmfDims &d = h->d;
    dim3 grid(d.Np / 64, d.KB, d.L);
    if (!den) den = h->numden + (size_t)d.L * d.K32 * d.Np;
    hipLaunchKernelGGL(w_update_kernel, grid, dim3(256), 0, h->stream, h->Wt, h->Wn, h->numden, den,
                       d.N, d.K, d.L, d.Np, d.K32, (float)l1W, (float)(2.0 * l2W), tail_src, tail_dst, tail_n); // mult.jl:37-38
    KCHK("w_update_kernel");
    set_est(h, 0);
    return CMF_OK;
}

// front: a shard whose group carries the halo of H in the W-phase all-reduce (cmf_groups.hip) also updates the hx = L-1 columns in
// FRONT of its own -- its left neighbour's last ones, which its loss conv and its next W phase read -- from an H that is valid 2 hx
// columns out: one more conv tile row (columns [-64, 0)), one more transconv block ([-128, 0)), the element-wise update from -hx on.
int h_update_impl(cmf_handle_s *h, double l1H, double l2H, bool front)
{
    const CmfDims &d = h->d;
    if (front) {
        if (!h->halo_ext || !h->left_data) return fail(CMF_ERR_STATE, "internal: this shard cannot update the columns in front of its own");
        // mult.jl:44 on columns [-64, Tl + halo_r): ONE launch of the one-wave tiles from tile row -1 on (a launch of its own for the
        // row in front cost 10 us per shard and iteration: profiles/r06_halo_in_allreduce_cost.txt)
        int np = 0;
        CMFTRY(launch_conv_rows<1>(h, h->estT, -1, 1 + (d.Tl + h->halo_r + 63) / 64, 0, h->n_cu, nullptr, nullptr, nullptr, &np, d.Tl + h->halo_r));
        CMFTRY(launch_transconv(h, 2, nullptr, true));                    // mult.jl:47-48 on columns [-128, Tl)
        const int hx = h->halo_ext, R = d.Tl + 128;
        dim3 gridx((d.Tl + hx + HUPD_T - 1) / HUPD_T, d.KB);
        const float *num = h->hslabs + (size_t)(128 - hx) * d.K32;        // slab row r holds column r - 128: the update starts at column -hx
        hipLaunchKernelGGL(h_update_kernel, gridx, dim3(256), 0, h->stream, h->H, h->Ht, num, (size_t)2 * R * d.K32, h->tc_S2, num + (size_t)R * d.K32,
                           (size_t)2 * R * d.K32, h->tc_S2, d.Tl + hx, d.K, d.K32, d.PADL - hx, d.TP, (float)l1H, (float)(2.0 * l2H)); // mult.jl:51-52
        KCHK("h_update_kernel");
        set_est(h, 0);
        return wb_after_H(h);
    }
    CMFTRY(launch_conv<1>(h, h->estT, d.Tl + h->halo_r, h->conv_gy_ext)); // mult.jl:44 (est with the new W)
    if (sk_can_fuse_h(h)) { // few components: mult.jl:47-48 and :51-52 in ONE launch (whoever completes a block's slabs updates the block)
        CMFTRY(launch_transconv_small(h, 2, nullptr, true, (float)l1H, (float)(2.0 * l2H)));
        ++h->sk_fused_h;
        set_est(h, 0);
        return wb_after_H(h);
    }
    CMFTRY(launch_transconv(h, 2));                                         // mult.jl:47-48
    dim3 grid((d.Tl + HUPD_T - 1) / HUPD_T, d.KB);
    const size_t TK = (size_t)d.Tl * d.K32;
    hipLaunchKernelGGL(h_update_kernel, grid, dim3(256), 0, h->stream, h->H, h->Ht, h->hslabs, 2 * TK, h->tc_S, h->hslabs + TK, 2 * TK, h->tc_S,
                       d.Tl, d.K, d.K32, d.PADL, d.TP, (float)l1H, (float)(2.0 * l2H)); // mult.jl:51-52
    KCHK("h_update_kernel");
    set_est(h, 0);
    return wb_after_H(h);
}

// the conv of mult.jl:55-57 with the loss fused: per-tile sums of (est - data)^2 -> h->partial
int launch_loss_conv(cmf_handle_s *h)
{
    const CmfDims &d = h->d;
    if (h->reuse_est && !h->gram) { // (the Gram form never reads est: nothing to keep)
        CMFTRY(launch_conv<3>(h, h->est, d.Tl, h->conv_gy)); // est kept for the next update_motifs!
        set_est(h, 1);
        return CMF_OK;
    }
    return launch_conv<2>(h, nullptr, d.Tl, h->conv_gy);
}

// Behind the loss conv of update_feature_maps! (est is current, H final): enqueue the C2 contraction the NEXT update_motifs! starts
// with (mult.jl:31-34 needs H, data and est only -- l1W, l2W enter the element-wise update), so that the device works on it while the
// loss travels to the host, the caller's loop comes round and the next call is made: on the reference's own problem sizes an
// iteration is a few launches long and that round trip was an eighth of it.  Used only when the caller alternates the two rule calls
// (the call before this one was update_motifs!); a caller that stops pays one contraction nobody reads.
static int w_speculate(cmf_handle_s *h)
{
    if (!(h->reuse_est && h->est_kind == 1) || h->gram || h->group || h->carry.partial) return CMF_OK;
    if (h->small_k) CMFTRY(hxt_contract(h, h->X, h->est, 2, nullptr, false, true));
    else CMFTRY(hxt_contract(h, h->X, h->est, 2, h->numden, true));
    h->spec_gen = h->est_gen;
    return CMF_OK;
}

static int ensure_ring(cmf_handle_s *h) // pinned words a loss reduction stores into and the host polls (0, 1: cmf_iterate's pipeline; 2: a synchronous read)
{
    if (!h->h_ring) HIPCHK(hipHostMalloc(&h->h_ring, 3 * sizeof(double), hipHostMallocCoherent));
    return CMF_OK;
}


// The loss conv (mult.jl:55-57) and the reduction of its per-tile sums.  readback: the calling thread returns with the sum -- the
// reduction stores it into a pinned word that the host polls (no copy operation and no stream synchronisation behind the last
// kernel: a rule call on a small problem is a few launches long, and the reference's loop makes one such read per iteration).
int loss_partial_impl(cmf_handle_s *h, double *sumsq, bool readback, double *host_out, bool speculate)
{
    volatile unsigned long long *word = nullptr;
    if (readback && !host_out) {
        CMFTRY(ensure_ring(h));
        host_out = h->h_ring + 2;
        word = reinterpret_cast<volatile unsigned long long *>(host_out);
        *word = CMF_SENTINEL64;
    }
    CMFTRY(launch_loss_conv(h));
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, h->stream, h->partial, h->conv_partials, h->d_scalar, host_out);
    KCHK("loss_reduce_kernel");
    if (speculate) CMFTRY(w_speculate(h)); // (enqueued behind the reduction: the loss does not wait for it)
    if (!readback) return CMF_OK;
    if (!word) return read_scalar(h, 0, sumsq);
    CMFTRY(wait_words<unsigned long long>(h->stream, word, 1, CMF_SENTINEL64, nullptr, nullptr));
    *sumsq = h->h_ring[2];
    return CMF_OK;
}

int set_factors_impl(cmf_handle_s *h, const double *W, const double *H)
{
    if (!h || (!W && !H)) return fail(CMF_ERR_ARG, "NULL argument");
    if ((!W || !H) && !h->factors_set) return fail(CMF_ERR_STATE, "the first cmf_set_factors needs both W and H");
    HIPCHK(hipSetDevice(h->device));
    const CmfDims &d = h->d;
    const size_t nW = (size_t)d.L * d.N * d.K, nH = (size_t)d.Tl * d.K;
    CMFTRY(ensure_stage(h, std::max(nW, nH)));
    // padding must be zero: clear, then scatter the valid entries (a NULL factor keeps its resident value)
    if (W) {
        HIPCHK(hipMemsetAsync(h->Wt, 0, (size_t)d.Lp * d.K32 * d.Np * sizeof(float), h->stream));
        HIPCHK(hipMemsetAsync(h->Wn, 0, (size_t)d.Lp * d.Np * d.K32 * sizeof(float), h->stream));
        HIPCHK(hipMemcpyAsync(h->stage, W, nW * sizeof(double), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(pack_W_kernel, dim3(1024), dim3(256), 0, h->stream, h->stage, d.N, d.K, d.L, h->Wt, h->Wn, d.Np, d.K32);
        KCHK("pack_W_kernel");
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    if (H) {
        HIPCHK(hipMemsetAsync(h->H, 0, (size_t)d.TP * d.K32 * sizeof(float), h->stream));
        HIPCHK(hipMemsetAsync(h->Ht, 0, (size_t)d.K32 * d.TP * sizeof(float), h->stream));
        HIPCHK(hipMemcpyAsync(h->stage, H, nH * sizeof(double), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(pack_H_kernel, dim3(1024), dim3(256), 0, h->stream, h->stage, d.Tl, d.K, h->H, h->Ht, d.K32, d.TP, d.PADL);
        KCHK("pack_H_kernel");
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    h->factors_set = true;
    set_est(h, 0);
    return CMF_OK;
}

int get_factors_impl(cmf_handle_s *h, double *W, double *H)
{
    if (!h) return fail(CMF_ERR_ARG, "handle is NULL");
    CMFTRY(check_ready(h, false));
    const CmfDims &d = h->d;
    const size_t nW = (size_t)d.L * d.N * d.K, nH = (size_t)d.Tl * d.K;
    CMFTRY(ensure_stage(h, std::max(nW, nH)));
    if (W) {
        hipLaunchKernelGGL(unpack_W_kernel<double>, dim3(1024), dim3(256), 0, h->stream, h->stage, d.N, d.K, d.L, h->Wn, d.Np, d.K32);
        KCHK("unpack_W_kernel");
        HIPCHK(hipMemcpyAsync(W, h->stage, nW * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    if (H) {
        hipLaunchKernelGGL(unpack_H_kernel<double>, dim3(1024), dim3(256), 0, h->stream, h->stage, d.Tl, d.K, h->H, d.K32, d.PADL);
        KCHK("unpack_H_kernel");
        HIPCHK(hipMemcpyAsync(H, h->stage, nH * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return CMF_OK;
}


// The sum of n per-tile partials -> d_scalar[slot], and to the calling thread when v is given (polled pinned word, like loss_partial_impl).
int reduce_partials(cmf_handle_s *h, const double *partial, int n, int slot, double *v)
{
    double *host_out = nullptr;
    if (v) {
        CMFTRY(ensure_ring(h));
        host_out = h->h_ring + 2;
        *reinterpret_cast<volatile unsigned long long *>(host_out) = CMF_SENTINEL64;
    }
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, h->stream, partial, n, h->d_scalar + slot, host_out);
    KCHK("loss_reduce_kernel");
    if (h->after_reduce) { // (enqueued behind the reduction: the sum does not wait for it -- the HALS rule's speculated W-phase contraction)
        std::function<int()> f = std::move(h->after_reduce);
        h->after_reduce = nullptr;
        if (v) CMFTRY(f());
    }
    if (!v) return CMF_OK;
    if (!host_out) return read_scalar(h, slot, v);
    CMFTRY(wait_words<unsigned long long>(h->stream, reinterpret_cast<volatile unsigned long long *>(host_out), 1, CMF_SENTINEL64, nullptr, nullptr));
    *v = h->h_ring[2];
    return CMF_OK;
}

// Wait until a kernel has replaced the sentinel pattern in `n` consecutive words of pinned host memory (a loss
// read-back).  Polling instead of an event keeps barrier packets and cache write-backs out of the stream; the stream is
// queried now and then so that a failed launch surfaces as an error instead of a hang.  `health` (optional) is called at
// the same cadence: a group passes a check of its other shards' streams and communicators, because the word is posted
// by shard 0 only and a peer that faulted would otherwise leave the host spinning here.  `enqueued` (optional): false while
// enqueue workers of the group have not yet handed all posted work to the streams.  The wait is bounded
// (CMF_WAIT_TIMEOUT_S seconds, default 300): a collective that can never complete ends in CMF_ERR_COMM, not in a hang.
double wait_timeout_s()
{
    static const double t = [] {
        const char *e = getenv("CMF_WAIT_TIMEOUT_S");
        const double v = e ? atof(e) : 0.0;
        return v > 0.0 ? v : 300.0;
    }();
    return t;
}

// ------------------------------------------------------------------------------------------
// write-back of the factors behind a rule call (cmf_writeback.h; cmf_arm_writeback)
// ------------------------------------------------------------------------------------------
static void wb_free(cmf_handle_s *h)
{
    CmfWriteback *wb = h->wb;
    if (!wb) return;
    if (!wb->pool.empty()) {
        const bool idle = cmf_pool_wait(wb->pool, 5.0);
        cmf_pool_stop(wb->pool, true);
        if (!idle) { // a helper is stuck in a wait on the device: it still reads this record and its staging -- leaked, never freed under it
            h->wb = nullptr;
            return;
        }
    }
    for (hipEvent_t e : {wb->ev_w_ready, wb->ev_h_ready, wb->ev_w_done, wb->ev_h_done})
        if (e) (void)hipEventDestroy(e);
    if (wb->pin_W) (void)hipHostFree(wb->pin_W);
    if (wb->pin_H) (void)hipHostFree(wb->pin_H);
    if (wb->dev_stage) (void)hipFree(wb->dev_stage);
    stream_release(h->device, wb->stream, !h->streams_may_hang);
    delete wb;
    h->wb = nullptr;
}

static int wb_alloc_copy(cmf_handle_s *h, CmfWriteback *wb, bool with_W) // the device side: copy stream, events, pinned staging
{
    const CmfDims &d = h->d;
    HIPCHK(hipSetDevice(h->device));
    wb->nW = (size_t)d.L * d.N * d.K;
    wb->nH = (size_t)d.Tl * d.K;
    HIPCHK(stream_acquire(h->device, &wb->stream));
    for (hipEvent_t *e : {&wb->ev_w_ready, &wb->ev_h_ready, &wb->ev_w_done, &wb->ev_h_done})
        HIPCHK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    if (with_W) HIPCHK(hipHostMalloc((void **)&wb->pin_W, wb->nW * sizeof(float), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&wb->pin_H, wb->nH * sizeof(float), hipHostMallocDefault));
    if (d.K != d.K32) HIPCHK(hipMalloc((void **)&wb->dev_stage, std::max(with_W ? wb->nW : 0, wb->nH) * sizeof(float)));
    wb->has_copy = true;
    return CMF_OK;
}

static void wb_start_pool(int device, CmfWriteback *wb) // the host side: the widening helpers
{
    if (!wb->pool.empty()) return;
    wb->pool.on_start = [device](size_t) { (void)hipSetDevice(device); };
    wb->pool.last_error = [] { return g_err; };
    static const int nthr_env = [] { const char *e = getenv("CMF_WRITEBACK_THREADS"); return e ? atoi(e) : 0; }();
    cmf_pool_start(wb->pool, (size_t)std::min(16, std::max(1, nthr_env > 0 ? nthr_env : 4)));
}

// what == 1: copy resources (a handle with device state), 2: helpers (the handle the caller holds), 3: both
static int wb_ensure(cmf_handle_s *h, int what, bool with_W = true)
{
    if (!h->wb) h->wb = new CmfWriteback();
    if ((what & 1) && !h->wb->has_copy) {
        const int rc = wb_alloc_copy(h, h->wb, with_W);
        if (rc != CMF_OK) { wb_free(h); return rc; }
    }
    if (what & 2) wb_start_pool(h->device, h->wb);
    return CMF_OK;
}

// W is final at the main stream's current position (update_motifs! has been enqueued): bring it to pinned memory on the copy stream
static int wb_start_W(cmf_handle_s *h)
{
    CmfWriteback *wb = h->wb;
    const CmfDims &d = h->d;
    HIPCHK(hipEventRecord(wb->ev_w_ready, h->stream));
    HIPCHK(hipStreamWaitEvent(wb->stream, wb->ev_w_ready, 0));
    if (d.K == d.K32) { // Wn[l][n][k] with K32 = K: every lag is N * K contiguous floats in Julia's order
        const size_t row = (size_t)d.N * d.K * sizeof(float);
        HIPCHK(hipMemcpy2DAsync(wb->pin_W, row, h->Wn, (size_t)d.Np * d.K32 * sizeof(float), row, (size_t)d.L, hipMemcpyDeviceToHost, wb->stream));
    } else {
        hipLaunchKernelGGL(unpack_W_kernel<float>, dim3(256), dim3(256), 0, wb->stream, wb->dev_stage, d.N, d.K, d.L, h->Wn, d.Np, d.K32);
        KCHK("unpack_W_kernel");
        HIPCHK(hipMemcpyAsync(wb->pin_W, wb->dev_stage, wb->nW * sizeof(float), hipMemcpyDeviceToHost, wb->stream));
    }
    HIPCHK(hipEventRecord(wb->ev_w_done, wb->stream));
    wb->w_started = true;
    return CMF_OK;
}

static int wb_drain(cmf_handle_s *h)
{
    CmfWriteback *wb = h->wb;
    if (wb->poisoned.load(std::memory_order_acquire)) return fail(CMF_ERR_HIP, "write-back: a helper thread of this handle never returned");
    if (!cmf_pool_wait(wb->pool, wait_timeout_s())) {
        // A helper is stuck in a wait on the device.  When it wakes up it must neither widen into arrays the caller has taken back nor
        // into the arrays of a later arm: the record is poisoned (helpers look at the flag behind every wait and leave), keeps its
        // pointers -- a helper reads only the copies in its own closure -- and takes no further arms (wb_arm).
        wb->poisoned.store(true, std::memory_order_release);
        return fail(CMF_ERR_HIP, "write-back: the helper threads did not finish within %.0f s", wait_timeout_s());
    }
    std::string err;
    const int rc = cmf_pool_collect(wb->pool, -1, &err);
    return rc == 0 ? CMF_OK : fail(rc, "write-back: %s", err.c_str());
}

// Hook of the rules' H phases: the kernels that make H final have just been enqueued on the main stream.
int wb_after_H(cmf_handle_s *h)
{
    CmfWriteback *wb = h->wb;
    if (!wb || !wb->armed || !wb->has_copy) return CMF_OK;
    const CmfDims &d = h->d;
    const bool shard = h->group != nullptr; // a shard of a group: issue the copy and say so; the front handle's helpers do the rest
    const bool again = !shard && wb->h_posted; // a second pass over H inside the same call (the HALS rerun): the helpers must be done with the staging
    if (again) CMFTRY(wb_drain(h));
    if (wb->dst_H) {
        HIPCHK(hipEventRecord(wb->ev_h_ready, h->stream));
        HIPCHK(hipStreamWaitEvent(wb->stream, wb->ev_h_ready, 0));
        if (d.K == d.K32) { // H[PADL + t][k] with K32 = K is Julia's K x T order
            HIPCHK(hipMemcpyAsync(wb->pin_H, h->H + (size_t)d.PADL * d.K32, wb->nH * sizeof(float), hipMemcpyDeviceToHost, wb->stream));
        } else {
            hipLaunchKernelGGL(unpack_H_kernel<float>, dim3(256), dim3(256), 0, wb->stream, wb->dev_stage, d.Tl, d.K, h->H, d.K32, d.PADL);
            KCHK("unpack_H_kernel");
            HIPCHK(hipMemcpyAsync(wb->pin_H, wb->dev_stage, wb->nH * sizeof(float), hipMemcpyDeviceToHost, wb->stream));
        }
        HIPCHK(hipEventRecord(wb->ev_h_done, wb->stream));
    }
    if (shard) {
        wb->h_issued.store(true, std::memory_order_release);
        return CMF_OK;
    }
    const size_t nthr = wb->pool.size();
    const bool want_W = !again && wb->dst_W && wb->w_started;
    double *const dst_W = wb->dst_W, *const dst_H = wb->dst_H; // (the helpers use these copies: the record's fields belong to the calling thread)
    for (size_t i = 0; i < nthr; ++i)
        cmf_pool_post(wb->pool, i, [wb, i, nthr, want_W, dst_W, dst_H]() -> int {
            auto slice = [&](size_t n, size_t *a, size_t *b) { // 16-element granules, the last helper takes the remainder
                const size_t per = (n / nthr) & ~(size_t)15;
                *a = i * per;
                *b = i + 1 == nthr ? n : (i + 1) * per;
            };
            size_t a, b;
            if (want_W) {
                if (hipEventSynchronize(wb->ev_w_done) != hipSuccess) return fail(CMF_ERR_HIP, "the download of W failed");
                if (wb->poisoned.load(std::memory_order_acquire)) return CMF_OK; // (the call gave up on this helper: the arrays are the caller's again)
                slice(wb->nW, &a, &b);
                cmf_widen(wb->pin_W + a, dst_W + a, b - a);
            }
            if (dst_H) {
                if (hipEventSynchronize(wb->ev_h_done) != hipSuccess) return fail(CMF_ERR_HIP, "the download of H failed");
                if (wb->poisoned.load(std::memory_order_acquire)) return CMF_OK;
                slice(wb->nH, &a, &b);
                cmf_widen(wb->pin_H + a, dst_H + a, b - a);
            }
            return CMF_OK;
        });
    if (!again) wb->hooked_calls += 1;
    wb->h_posted = true;
    return CMF_OK;
}


// A group's MU H phase is about to be enqueued: post the widening of W (shard 0's copy was started when the write-back was armed)
// and of every shard's column block of H to the front handle's helpers.  Helper j takes slice j of every block.
static int wb_group_post(cmf_handle_s *h)
{
    CmfWriteback *wb = h->wb;
    if (!wb || !wb->armed || !h->group) return CMF_OK;
    cmf_group_s *g = h->group;
    const size_t nthr = wb->pool.size();
    const double tmo = wait_timeout_s();
    struct Dst { double *W, *H; };
    std::vector<Dst> dst; // (the helpers use these copies of the shards' pointers: the records' fields belong to the calling thread)
    for (cmf_handle_s *s : g->sh) dst.push_back(s->wb && s->wb->armed ? Dst{s->wb->w_started ? s->wb->dst_W : nullptr, s->wb->dst_H} : Dst{nullptr, nullptr});
    for (size_t j = 0; j < nthr; ++j)
        cmf_pool_post(wb->pool, j, [g, wb, j, nthr, tmo, dst]() -> int {
            auto slice = [&](size_t n, size_t *a, size_t *b) {
                const size_t per = (n / nthr) & ~(size_t)15;
                *a = j * per;
                *b = j + 1 == nthr ? n : (j + 1) * per;
            };
            for (size_t i = 0; i < g->sh.size(); ++i) {
                CmfWriteback *sw = g->sh[i]->wb;
                if (!sw || (!dst[i].W && !dst[i].H)) continue;
                (void)hipSetDevice(g->sh[i]->device);
                size_t a, b;
                if (dst[i].W) {
                    if (hipEventSynchronize(sw->ev_w_done) != hipSuccess) return fail(CMF_ERR_HIP, "the download of W failed");
                    if (wb->poisoned.load(std::memory_order_acquire)) return CMF_OK;
                    slice(sw->nW, &a, &b);
                    cmf_widen(sw->pin_W + a, dst[i].W + a, b - a);
                }
                if (dst[i].H) {
                    const auto t0 = std::chrono::steady_clock::now();
                    for (unsigned spins = 1; !sw->h_issued.load(std::memory_order_acquire); ++spins) { // (the copy is issued by whoever enqueues the shard)
                        if (sw->cancel.load(std::memory_order_acquire)) return CMF_OK;
                        if ((spins & 0xFFFF) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > tmo)
                            return fail(CMF_ERR_HIP, "write-back: shard %zu never issued its copy of H", i);
                        CMF_CPU_PAUSE();
                    }
                    if (hipEventSynchronize(sw->ev_h_done) != hipSuccess) return fail(CMF_ERR_HIP, "the download of H failed");
                    if (wb->poisoned.load(std::memory_order_acquire)) return CMF_OK;
                    slice(sw->nH, &a, &b);
                    cmf_widen(sw->pin_H + a, dst[i].H + a, b - a);
                }
            }
            return CMF_OK;
        });
    wb->h_posted = true;
    wb->hooked_calls += 1;
    return CMF_OK;
}

// End of a rule call that may carry an armed write-back: the caller's arrays are complete (or the call fails) and are not
// touched after this returns.
static int wb_finish(cmf_handle_s *h, int rc)
{
    CmfWriteback *wb = h ? h->wb : nullptr;
    if (!wb || !wb->armed) return rc;
    double *W = wb->dst_W, *H = wb->dst_H;
    int rc2 = CMF_OK;
    if (wb->h_posted) {
        if (rc != CMF_OK && h->group) // the phase failed somewhere: a shard's copy may never be issued -- release the helpers that wait for it
            for (cmf_handle_s *s : h->group->sh)
                if (s->wb) s->wb->cancel.store(true, std::memory_order_release);
        rc2 = wb_drain(h);
    } else if (rc == CMF_OK) { // no hook on this path (the PGD rule on a group): the synchronous download
        wb->armed = false;
        rc2 = cmf_get_factors(h, W, H);
    }
    if (h->group)
        for (cmf_handle_s *s : h->group->sh)
            if (s->wb) {
                s->wb->armed = s->wb->w_started = false;
                s->wb->dst_W = s->wb->dst_H = nullptr;
            }
    wb->armed = wb->h_posted = wb->w_started = false; // (a poisoned record stays unarmed for good: wb_arm refuses it)
    wb->dst_W = wb->dst_H = nullptr;
    return rc != CMF_OK ? rc : rc2;
}


// A loss reduction deferred by cmf_iterate (CmfLossCarry) only lives between two phases of that call.  If the call
// failed in between, the record would be consumed by the next W phase and reduce partial sums that have been
// overwritten since: every public entry that starts a phase drops whatever an aborted batch left behind.
static void drop_carry(cmf_handle_s *h)
{
    if (!h) return;
    if (h->group) {
        for (cmf_handle_s *s : h->group->sh) s->carry = CmfLossCarry{};
        return;
    }
    h->carry = CmfLossCarry{};
}
struct CarryGuard { // error exits of cmf_iterate: leave no deferred reduction (and no half-reduced numerator) behind
    cmf_handle_s *h;
    bool armed = true;
    ~CarryGuard()
    {
        if (!armed) return;
        if (h && h->group) (void)group_join(h->group); // (enqueue workers may still be walking the batch: they read the carry records)
        drop_carry(h);
        if (h && h->group) h->group->num_ready = false;
    }
};

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------

#ifndef CMF_SRC_DIGEST
#define CMF_SRC_DIGEST "unknown" // cmf.jl_amd/build.py passes -DCMF_SRC_DIGEST=... (cmf_source_digest, include/cmf_hip.h)
#endif
#define CMF_STR2(x) #x
#define CMF_STR(x) CMF_STR2(x)
const char *cmf_version(void) { return "cmf_hip gfx950 0.6.0 abi=" CMF_STR(CMF_ABI_VERSION) " src=" CMF_SRC_DIGEST; }
const char *cmf_source_digest(void) { return CMF_SRC_DIGEST; }
int cmf_abi_version(void) { return CMF_ABI_VERSION; }
const char *cmf_last_error(void) { return g_err.c_str(); }

int cmf_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// columns one handle can hold: a 64-row block of est' (64 * Tpad floats) must stay below the 2 GiB a 32-bit buffer offset
// addresses, and so must H (Tpad * Kpad floats)
static int64_t max_columns_per_handle(int64_t K)
{
    // tests take the long-recording paths at small sizes; the knob is honoured only together with CMF_TEST_HOOKS=1, so that
    // a stray variable in a production environment cannot move the cut point
    if (const long long forced = test_hook("CMF_MAX_COLUMNS", 0); forced > 0) return forced;
    const int64_t K32 = 32 * ((K + 31) / 32);
    const int64_t by_est = ((int64_t)1 << 31) / (64 * 4), by_h = ((int64_t)1 << 31) / (K32 * 4);
    return std::min(by_est, by_h) - 4096; // (left lag halo and tile padding)
}

int cmf_create(cmf_handle *h, int device, int64_t N, int64_t T, int64_t K, int64_t L, const double *data)
{
    if (!data) return fail(CMF_ERR_ARG, "data is NULL");
    if (K >= 1 && L >= 1 && T > max_columns_per_handle(K)) {
        // A recording longer than one handle's 32-bit offsets reach (8.3 M columns at K <= 64; the reference's own long
        // recordings have 20 M: notebooks/test_mouse.ipynb cell 5) is cut along T into shards on the SAME device: a loopback
        // group behind the same handle type, so the rule constructor of the reference (mult.jl:11-20, pgd.jl:139-154) just works.
        const int64_t per = max_columns_per_handle(K) - L;
        const int R = (int)((T + per - 1) / per);
        if (R > CMF_MAX_LOCAL) return fail(CMF_ERR_UNSUPPORTED, "T=%lld needs %d shards on one device (at most %d)", (long long)T, R, CMF_MAX_LOCAL);
        std::vector<int> devs((size_t)R, device);
        return cmf_create_multi(h, R, devs.data(), CMF_COMM_LOOPBACK, N, T, K, L, data);
    }
    return create_impl(h, device, N, T, K, L, data, 0, T, false);
}

int cmf_create_shard(cmf_handle *h, int device, int64_t N, int64_t T_local, int64_t K, int64_t L,
                     const double *data_local, int64_t t_offset, int64_t T_global)
{
    if (!data_local) return fail(CMF_ERR_ARG, "data_local is NULL");
    if (T_local < L - 1 && T_local != T_global)
        return fail(CMF_ERR_UNSUPPORTED, "a shard must hold at least L-1 = %lld columns (got %lld)", (long long)(L - 1), (long long)T_local);
    return create_impl(h, device, N, T_local, K, L, data_local, t_offset, T_global, true);
}

int cmf_shard_set_left_data(cmf_handle h, const double *cols)
{
    if (!h || !cols) return fail(CMF_ERR_ARG, "NULL argument");
    if (h->group) return fail(CMF_ERR_STATE, "call cmf_shard_set_left_data before the shard joins its group (cmf_comm_init_*)");
    if (!h->sharded) return fail(CMF_ERR_STATE, "cmf_shard_set_left_data needs a handle from cmf_create_shard");
    if (!h->halo_ext) return CMF_OK; // (the first shard, or a shape on which the halo keeps its own all-gather: nothing to keep)
    HIPCHK(hipSetDevice(h->device));
    CMFTRY(upload_cols(h, cols, -h->halo_ext, h->halo_ext, false, false)); // data' only: the operand of the transconv's front block
    h->left_data = true;
    return CMF_OK;
}

int cmf_destroy(cmf_handle h)
{
    if (h && h->group) {
        cmf_group_s *g = h->group;
        if (h->wb && !h->wb->pool.empty()) { // write-back helpers read the group's shards: none may outlive the group
            for (cmf_handle_s *s : g->sh)
                if (s->wb) s->wb->cancel.store(true, std::memory_order_release);
            if (!cmf_pool_wait(h->wb->pool, 5.0)) // (stuck in a wait on a device: the group is leaked rather than freed under the helper)
                return fail(CMF_ERR_HIP, "cmf_destroy: a write-back helper is stuck in a device wait; the group was not freed");
        }
        std::vector<cmf_handle_s *> shards = g->sh;
        if (!group_destroy(g)) // detaches the shards from the group's buffers
            return fail(CMF_ERR_COMM, "cmf_destroy: an enqueue worker is stuck inside a call that does not return; the group and its shards were not freed");
        for (cmf_handle_s *s : shards) {
            s->group = nullptr;
            if (s != h) destroy_impl(s);
        }
        h->group = nullptr;
    }
    destroy_impl(h);
    return CMF_OK;
}

int cmf_synchronize(cmf_handle h)
{
    if (!h) return fail(CMF_ERR_ARG, "handle is NULL");
    if (h->group) return group_sync(h->group);
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    return CMF_OK;
}

int cmf_rccl_version(int *version, char *path, int64_t path_len)
{
    if (!version) return fail(CMF_ERR_ARG, "version is NULL");
    CMFTRY(rccl_load());
    RCCLCHK(g_rccl.GetVersion(version));
    if (path && path_len > 0) snprintf(path, (size_t)path_len, "%s", g_rccl.path.c_str());
    return CMF_OK;
}

int cmf_get_counter(cmf_handle h, const char *name, int64_t *value)
{
    if (!h || !name || !value) return fail(CMF_ERR_ARG, "NULL argument");
    if (std::strcmp(name, "hals_pipeline_reruns") == 0) { *value = h->hals_reruns; return CMF_OK; }
    if (std::strcmp(name, "writeback_calls") == 0) { *value = h->wb ? h->wb->armed_calls : 0; return CMF_OK; }            // cmf_arm_writeback calls
    if (std::strcmp(name, "speculated_contractions") == 0) { *value = h->spec_hits; return CMF_OK; }                         // update_motifs! calls whose C2 contraction was already enqueued
    if (std::strcmp(name, "liveness_checks") == 0) { *value = g_liveness_checks.load(); return CMF_OK; } // (process-wide) stream queries made while waiting for a loss
    if (std::strcmp(name, "small_k_fused_h_updates") == 0) { // (a group: over its shards)
        *value = h->sk_fused_h;
        if (h->group)
            for (const cmf_handle_s *s : h->group->sh) *value += (s != h) ? s->sk_fused_h : 0;
        return CMF_OK;
    }                       // H updates that ran inside the few-component C3 launch
    if (std::strcmp(name, "writeback_overlapped") == 0) { *value = h->wb ? h->wb->hooked_calls : 0; return CMF_OK; }     // ... served by the copy stream behind the H update
    if (h->group) { // host cost of the pipelined iterations of a group (reading a counter resets nothing)
        cmf_group_s *g = h->group;
        if (std::strcmp(name, "allreduce_calls") == 0) { *value = g->n_allreduce; return CMF_OK; } // collectives this handle has issued since it was made
        if (std::strcmp(name, "allgather_calls") == 0) { *value = g->n_allgather; return CMF_OK; } // (per shard: every rank makes each of them once)
        if (std::strcmp(name, "halo_in_allreduce") == 0) { *value = (g->halo_opt && g->halo_can && !g->gram && g->nranks > 1) ? 1 : 0; return CMF_OK; }
        if (std::strcmp(name, "enqueue_ns") == 0) { *value = g->enqueue_ns; return CMF_OK; }       // calling thread: enqueueing / posting
        if (std::strcmp(name, "enqueue_iters") == 0) { *value = g->enqueue_iters; return CMF_OK; } // ... over this many iterations
        if (std::strcmp(name, "worker_ns") == 0) {                                                  // busiest enqueue worker: time inside its jobs
            int64_t m = 0;
            for (const auto &w : g->pool.w) m = std::max<int64_t>(m, w->busy_ns.load());
            *value = m;
            return CMF_OK;
        }
    }
    return fail(CMF_ERR_ARG, "unknown counter '%s'", name);
}

int cmf_set_stream(cmf_handle h, void *hip_stream)
{
    if (!h) return fail(CMF_ERR_ARG, "handle is NULL");
    if (h->group) return fail(CMF_ERR_STATE, "a group handle runs on its own per-device streams");
    if ((hipStream_t)hip_stream != h->stream) {
        // what is still running on the stream so far -- since round 5 a rule call may return with a speculated contraction in flight
        // (option "speculate") -- is not ordered with the new stream: wait for it, and let nothing speculated carry over
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->spec_gen = -1;
        h->hals_spec_gen = -1;
    }
    h->stream = (hipStream_t)hip_stream;
    return CMF_OK;
}

// every name cmf_set_option knows (cmf_option_names; tests/test_library_abi.py walks the table)
static const char *const kOptionNames[] = {"reuse_est", "speculate", "gram", "conv_kernel", "conv_split", "small_k", "small_k_fuse", "hals_prepare", "hals_gram",
                                           "hals_persist", "hals_general", "hals_seg", "hals_lag", "hals_debug", "hals_chase", "profile", "profile_mask",
                                           "allreduce_overlap", "enqueue_threads", "halo_in_allreduce"};
int cmf_option_names(char *buf, int64_t len)
{
    if (!buf || len < 1) return fail(CMF_ERR_ARG, "bad buffer");
    std::string all;
    for (const char *n : kOptionNames) all += (all.empty() ? "" : ",") + std::string(n);
    if ((int64_t)all.size() + 1 > len) return fail(CMF_ERR_ARG, "buffer too small: %zu bytes needed", all.size() + 1);
    std::memcpy(buf, all.c_str(), all.size() + 1);
    return CMF_OK;
}

int cmf_set_option(cmf_handle h, const char *name, int value)
{
    if (!h || !name) return fail(CMF_ERR_ARG, "NULL argument");
    if (h->group) {
        cmf_group_s *g = h->group;
        CMFTRY(group_join(g));
        if (std::strcmp(name, "allreduce_overlap") == 0) {
            CMFTRY(group_sync(g));
            if (value) CMFTRY(group_ensure_lane1(g)); // the communication stream gets a communicator of its own
            g->overlap = value != 0;
            g->num_ready = false;
            return CMF_OK;
        }
        if (std::strcmp(name, "halo_in_allreduce") == 0) { // 1 (default): the halo of H travels in the W-phase all-reduce where every shard can (cmf_groups.hip)
            CMFTRY(group_sync(g));
            g->halo_opt = value != 0;
            g->halos_current = false; // (the next phase starts from a whole exchange in the form that is now in force)
            g->halos_pending = false;
            g->halo_wide = false;
            g->num_ready = false;
            return CMF_OK;
        }
        if (std::strcmp(name, "enqueue_threads") == 0) { // 1: an enqueue worker per shard (cmf_group.h), 0: the calling thread enqueues every shard
            CMFTRY(group_sync(g));
            if (value) return group_start_workers(g);
            (void)group_stop_workers(g); // (the group has just been synchronised: nobody is inside a job)
            return CMF_OK;
        }
        if (std::strcmp(name, "gram") == 0) {
            // Gram form on a T-sharded group: every shard contracts numW and its additive share of HH = H_unfold H_unfold'
            // from its own columns, the all-reduce carries [numW | HH | tail] (numW + (L*Kpad)^2 floats instead of 2 numW),
            // denomW = HH * W is formed on every shard; denomH = lag-Gram taps of W applied to H with the H halos.
            if (value != 0 && value != 1) return fail(CMF_ERR_UNSUPPORTED, "sharded handles take gram = 0 or 1 (the loss of gram = 2 is not exact to 1e-4)");
            if (value && g->nranks > 1 && g->T < 4 * g->L)
                return fail(CMF_ERR_UNSUPPORTED, "the Gram form on a sharded handle needs T >= 4 L (got T=%lld, L=%lld)", (long long)g->T, (long long)g->L);
            CMFTRY(group_sync(g));
            for (size_t i = 0; i < g->sh.size(); ++i) {
                cmf_handle_s *s = g->sh[i];
                CMFTRY(group_use(s));
                if (value) CMFTRY(gram_ensure(s));
                // the HH block shares its place with denomW of the default form; hals_hh_kernel writes the first L*K columns
                // of each 128-padded row only, and the all-reduce sums the whole block in place: whatever the pad columns
                // held would be multiplied by the rank count every iteration until it overflowed
                if (value && i < g->red.size() && g->red[i])
                    HIPCHK(hipMemsetAsync(g->red[i] + g->LKN2 / 2, 0, (size_t)g->HHsz * sizeof(float), s->stream));
                s->gram = value;
                set_est(s, 0);
                s->carry = CmfLossCarry{};
            }
            g->gram = value;
            g->num_ready = false;
            return CMF_OK;
        }
        for (cmf_handle_s *s : g->sh) {
            cmf_group_s *keep = s->group;
            s->group = nullptr;
            const int rc = cmf_set_option(s, name, value);
            s->group = keep;
            if (rc != CMF_OK) return rc;
        }
        g->num_ready = false;
        return CMF_OK;
    }
    if (std::strcmp(name, "allreduce_overlap") == 0 || std::strcmp(name, "enqueue_threads") == 0 || std::strcmp(name, "halo_in_allreduce") == 0) return CMF_OK; // single GPU: nothing to overlap, nobody else to enqueue
    if (std::strcmp(name, "gram") == 0) {
        if (value < 0 || value > 2) return fail(CMF_ERR_ARG, "gram must be 0, 1 or 2");
        if (value && h->sharded && h->T_global != h->d.Tl) return fail(CMF_ERR_STATE, "the Gram form is not available on sharded handles");
        h->gram = value;
        set_est(h, 0);
        return CMF_OK;
    }
    if (std::strcmp(name, "conv_kernel") == 0) { // K % 32 == 0 only: 0 = chosen per mode (default), 3 = one-wave workgroups, 2 = 128 x 128 tiles
        if (value != 0 && value != 2 && value != 3) return fail(CMF_ERR_ARG, "conv_kernel must be 0 (per mode), 2 or 3");
        h->conv_variant = value;
        set_est(h, 0);
        return CMF_OK;
    }
    if (std::strcmp(name, "hals_gram") == 0) { // HALS projections as differences of the MU quantities: 2 (default) = H phase, 1 = both phases, 0 = neither
        h->hals_gram = (value == 1 || value == 2) ? value : 0; // 2 = the H phase only
        set_est(h, 0);
        return CMF_OK;
    }
    if (std::strcmp(name, "hals_persist") == 0 || std::strcmp(name, "hals_general") == 0 || std::strcmp(name, "hals_seg") == 0 ||
        std::strcmp(name, "hals_lag") == 0 || std::strcmp(name, "hals_debug") == 0 || std::strcmp(name, "hals_chase") == 0) {
        if (value < 0 && !(name[5] == 'c' && value == -1)) return fail(CMF_ERR_ARG, "%s must be >= 0", name);
        if (name[5] == 'd') { // "hals_debug": results are wrong by design -- tests of the bounded waits only
            if (value && !test_hooks_on()) return fail(CMF_ERR_STATE, "hals_debug needs CMF_TEST_HOOKS=1");
            h->hals_debug = value;
            return CMF_OK;
        }
        if (name[5] == 'c') { // "hals_chase": per cent of the residual conv's tile rows that chase the row pipeline
            if (value > 100) return fail(CMF_ERR_ARG, "hals_chase is a percentage (0 = off)");
            h->hals_opt_chase = value;
            if (h->hals_ready) {
                HIPCHK(hipSetDevice(h->device));
                HIPCHK(hipStreamSynchronize(h->stream));
                hals_plan(h);
            }
            return CMF_OK;
        }
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(hipStreamSynchronize(h->stream));
        (name[5] == 'p' ? h->hals_opt_persist : name[5] == 'g' ? h->hals_opt_general : name[5] == 's' ? h->hals_opt_seg : h->hals_opt_lag) = value;
        if (h->hals_ready) hals_plan(h);
        return CMF_OK;
    }
    if (std::strcmp(name, "hals_prepare") == 0) { // allocate the HALS scratch and check its shape limits now (rule construction)
        HIPCHK(hipSetDevice(h->device));
        return hals_ensure(h);
    }
    if (std::strcmp(name, "conv_split") == 0) { // 0 = never cut the one-wave conv kernel's last round into quarter tiles
        h->conv_split = value;
        set_est(h, 0);
        return CMF_OK;
    }
    if (std::strcmp(name, "small_k_fuse") == 0) { // few components: 1 (default) = the element-wise update of H inside the C3 launch, 0 = a launch of its own
        if (value < 0 || value > 2) return fail(CMF_ERR_ARG, "small_k_fuse must be 0, 1 or 2");
        h->sk_fuse = value; // (2: also on one-round launches, where it is slower)
        return CMF_OK;
    }
    if (std::strcmp(name, "small_k") == 0) { // K <= 16: 1 = the few-component kernels (cmf_small_k.h; default), 0 = the general kernels
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->small_k = value != 0 && h->small_k_ok;
        h->sk_wj_gen = -1;
        h->sk_tc = h->small_k && (h->sk_tc_ok || value == 2); // (2: the few-component C3 form whatever T is -- tests, measurements)
        h->tc_S = h->sk_tc ? 2 * h->sk3_NS : h->tc_S_full; // (own block | the spill of the next block, per piece of the reduction: g_gemm_fold_small_kernel)
        h->tc_S1 = h->sk_tc ? 2 * h->sk3_NS : h->tc_S1_full;
        set_est(h, 0);
        return CMF_OK;
    }
    if (std::strcmp(name, "reuse_est") == 0) {
        h->reuse_est = value != 0;
        set_est(h, 0);
        return CMF_OK;
    }
    if (std::strcmp(name, "speculate") == 0) { // 1 (default): update_feature_maps! enqueues the next update_motifs!'s contraction behind its loss conv when the caller alternates the two calls
        h->speculate = value != 0;
        h->spec_gen = -1;
        return CMF_OK;
    }
    if (std::strcmp(name, "profile_mask") == 0) { // which kernel classes "profile" times (bit i: class i of cmf_kernel_times' names in their order; 0 = all)
        h->prof_mask = (unsigned)value;
        return CMF_OK;
    }
    if (std::strcmp(name, "profile") == 0) { // (re)start or stop the in-loop kernel timing; collected times are dropped
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(hipStreamSynchronize(h->stream));
        for (auto &r : h->prof_recs) { h->prof_pool.push_back(r.a); h->prof_pool.push_back(r.b); }
        h->prof_recs.clear();
        h->prof = value != 0;
        h->prof_every = value > 1 ? value : 1;
        for (int &c : h->prof_seen) c = 0;
        return CMF_OK;
    }
    return fail(CMF_ERR_ARG, "unknown option '%s'", name);
}

int cmf_get_data_sumsq(cmf_handle h, double *sumsq)
{
    if (!h || !sumsq) return fail(CMF_ERR_ARG, "NULL argument");
    if (h->group) { *sumsq = h->group->data_sumsq; return CMF_OK; }
    *sumsq = h->data_sumsq;
    return CMF_OK;
}

int cmf_set_factors(cmf_handle h, const double *W, const double *H)
{
    if (!h || (!W && !H)) return fail(CMF_ERR_ARG, "NULL argument");
    if (h->wb && h->wb->armed) wb_disarm(h); // (a W copy in flight would deliver the factor this call replaces)
    drop_carry(h);
    if (h->group) return group_set_factors(h->group, W, H);
    return set_factors_impl(h, W, H);
}

int cmf_get_factors(cmf_handle h, double *W, double *H)
{
    if (!h) return fail(CMF_ERR_ARG, "handle is NULL");
    if (h->group) return group_get_factors(h->group, W, H);
    return get_factors_impl(h, W, H);
}

int cmf_update_motifs(cmf_handle h, double l1W, double l2W)
{
    RoctxRange range("cmf_update_motifs");
    drop_carry(h);
    if (h && h->group) {
        CMFTRY(group_check_ready(h->group));
        CMFTRY(group_update_motifs(h->group, l1W, l2W));
        return group_join(h->group); // (enqueue workers: the phase has been handed to the streams when this returns)
    }
    CMFTRY(check_ready(h, true));
    if (h->sharded && h->T_global != h->d.Tl) return fail(CMF_ERR_STATE, "sharded handle: attach a communicator first (cmf_comm_init_rccl / cmf_comm_init_callbacks), or build the group with cmf_create_multi");
    if (h->gram) return gram_w_impl(h, l1W, l2W);
    h->last_rule_call = 1;
    return w_phase_impl(h, l1W, l2W);
}

static int update_feature_maps_body(cmf_handle h, double l1H, double l2H, double *loss)
{
    RoctxRange range("cmf_update_feature_maps");
    if (!loss) return fail(CMF_ERR_ARG, "loss is NULL");
    drop_carry(h);
    if (h && h->group) {
        CMFTRY(group_check_ready(h->group));
        CMFTRY(wb_group_post(h)); // (an armed write-back: the helpers wait for every shard's copy of its block of H)
        double ss = 0.0;
        CMFTRY(group_update_feature_maps(h->group, l1H, l2H, &ss));
        *loss = std::sqrt(ss) / h->group->data_norm;
        return CMF_OK;
    }
    CMFTRY(check_ready(h, true));
    if (h->sharded && h->T_global != h->d.Tl) return fail(CMF_ERR_STATE, "sharded handle: attach a communicator first (cmf_comm_init_rccl / cmf_comm_init_callbacks), or build the group with cmf_create_multi");
    if (h->gram) return gram_h_impl(h, l1H, l2H, loss);
    const bool speculate = h->speculate && h->last_rule_call == 1; // the caller alternates: update_motifs! comes next
    h->last_rule_call = 2;
    CMFTRY(h_update_impl(h, l1H, l2H));
    double ss = 0.0;
    CMFTRY(loss_partial_impl(h, &ss, true, nullptr, speculate));
    *loss = std::sqrt(ss) / h->data_norm;
    return CMF_OK;
}

int cmf_update_feature_maps(cmf_handle h, double l1H, double l2H, double *loss)
{
    return wb_finish(h, update_feature_maps_body(h, l1H, l2H, loss));
}

static void wb_disarm(cmf_handle_s *h)
{
    if (h->wb && h->wb->h_posted) (void)wb_drain(h);
    auto clear = [](CmfWriteback *wb) {
        if (!wb) return;
        wb->armed = wb->h_posted = wb->w_started = false;
        wb->dst_W = wb->dst_H = nullptr;
    };
    if (h->group)
        for (cmf_handle_s *s : h->group->sh) clear(s->wb);
    clear(h->wb);
}

static int wb_arm(cmf_handle_s *h, double *W, double *H)
{
    if (h->wb && h->wb->poisoned.load(std::memory_order_acquire))
        return fail(CMF_ERR_STATE, "write-back: a helper thread of this handle never returned from an earlier call; the handle takes no further "
                                   "write-backs (cmf_get_factors still works)");
    if (h->group) {
        cmf_group_s *g = h->group;
        CMFTRY(group_check_ready(g));
        CMFTRY(group_join(g)); // (the enqueue workers are idle: this thread may use the shards' streams)
        CMFTRY(wb_ensure(h, 2));
        for (size_t i = 0; i < g->sh.size(); ++i) { // every shard: its own column block of H (column-major: contiguous), shard 0 also W
            cmf_handle_s *s = g->sh[i];
            CMFTRY(wb_ensure(s, 1, i == 0));
            CmfWriteback *sw = s->wb;
            sw->dst_W = i == 0 ? W : nullptr;
            sw->dst_H = !H ? nullptr : (g->one_process ? H + (size_t)g->t0[(size_t)g->rank[i]] * g->K : H);
            sw->armed = true;
            sw->w_started = false;
            sw->h_issued.store(false, std::memory_order_relaxed);
            sw->cancel.store(false, std::memory_order_relaxed);
            if (sw->dst_W) {
                CMFTRY(group_use(s));
                CMFTRY(wb_start_W(s));
            }
        }
    } else {
        CMFTRY(check_ready(h, false));
        CMFTRY(wb_ensure(h, 3));
    }
    CmfWriteback *wb = h->wb;
    if (!(h->group && !h->root_only)) { // (a per-process shard is its own front handle: its fields were set above)
        wb->dst_W = W;
        wb->dst_H = H;
    }
    wb->armed = true;
    wb->h_posted = false;
    if (!h->group) wb->w_started = false;
    wb->armed_calls += 1;
    if (W && !h->group) CMFTRY(wb_start_W(h));
    return CMF_OK;
}

int cmf_arm_writeback(cmf_handle h, double *W, double *H)
{
    if (!h) return fail(CMF_ERR_ARG, "handle is NULL");
    if (!W && !H) {
        wb_disarm(h);
        return CMF_OK;
    }
    const int rc = wb_arm(h, W, H);
    if (rc != CMF_OK) wb_disarm(h); // (nothing half-armed keeps the caller's pointers)
    return rc;
}

int cmf_compute_loss(cmf_handle h, double *loss)
{
    if (!loss) return fail(CMF_ERR_ARG, "loss is NULL");
    drop_carry(h);
    if (h && h->group) {
        CMFTRY(group_check_ready(h->group));
        return group_compute_loss(h->group, loss);
    }
    CMFTRY(check_ready(h, true));
    double ss = 0.0;
    CMFTRY(loss_partial_impl(h, &ss));
    *loss = std::sqrt(ss) / h->data_norm;
    return CMF_OK;
}

int cmf_hals_update_motifs(cmf_handle h, double l1W, double l2W)
{
    if (h && h->group) return fail(CMF_ERR_STATE, "this rule needs a single-GPU handle (its sweeps / step control do not shard over T)");
    CMFTRY(check_ready(h, true));
    h->last_rule_call = 1;
    return hals_w_impl(h, l1W, l2W);
}

static int hals_update_feature_maps_body(cmf_handle h, double l1H, double l2H, double *loss)
{
    if (h && h->group) return fail(CMF_ERR_STATE, "this rule needs a single-GPU handle (its sweeps / step control do not shard over T)");
    if (!loss) return fail(CMF_ERR_ARG, "loss is NULL");
    CMFTRY(check_ready(h, true));
    const bool speculate = h->speculate && h->last_rule_call == 1; // the caller alternates (alternating.jl:51-54): update_motifs! comes next
    h->last_rule_call = 2;
    CMFTRY(hals_h_impl(h, l1H, l2H));
    double ss = 0.0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        // ... whose contraction on the residual and lag correlations (they need H and the residual only; l1W, l2W enter in the sweep) go
        // out behind the loss reduction, so that the device works while the loss travels and the caller's loop comes round
        if (speculate && attempt == 0 && h->hals_gram != 1) h->after_reduce = [h]() { return hals_w_speculate(h); };
        if (h->hals_gram == 1) { // hals.jl:41: norm(resids)/data_norm -- the conv with the loss fused in its epilogue, nothing stored
            CMFTRY(launch_conv<2>(h, nullptr, h->d.Tl, h->conv_gy));
            CMFTRY(reduce_partials(h, h->partial, h->conv_partials, 0, &ss));
            set_est(h, 0);
        } else {
            CMFTRY(hals_resid_and_loss(h, &ss)); // ... and the residual kept for the next W phase (part of it may have chased the sweep)
        }
        // (the loss has arrived: every kernel in front of its reduction has completed)  A bounded wait of the persistent H pipeline ran out:
        // redo the sweep from the snapshot on the stage pipeline and take the loss again; counted in "hals_pipeline_reruns"
        h->after_reduce = nullptr;
        if (attempt == 0 && h->hals_status && *h->hals_status) CMFTRY(hals_h_rerun(h));
        else break;
    }
    *loss = std::sqrt(ss) / h->data_norm;
    return CMF_OK;
}

int cmf_hals_update_feature_maps(cmf_handle h, double l1H, double l2H, double *loss)
{
    return wb_finish(h, hals_update_feature_maps_body(h, l1H, l2H, loss));
}

int cmf_set_mask(cmf_handle h, const double *mask)
{
    if (!h) return fail(CMF_ERR_ARG, "handle is NULL");
    if (h->group) return group_set_mask(h->group, mask);
    HIPCHK(hipSetDevice(h->device));
    if (h->sharded) return fail(CMF_ERR_STATE, "sharded handle: attach a communicator first (cmf_comm_init_rccl / cmf_comm_init_callbacks)");
    const CmfDims &d = h->d;
    set_est(h, 0);
    if (!mask) { // back to the plain SquareLoss
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->M) (void)hipFree(h->M);
        if (h->MT) (void)hipFree(h->MT);
        h->M = h->MT = nullptr;
        return CMF_OK;
    }
    const size_t TPNp = (size_t)d.TP * d.Np;
    if (!h->M) CMFTRY(dalloc_zero(&h->M, TPNp));
    if (!h->MT) CMFTRY(dalloc_zero(&h->MT, TPNp));
    return upload_cols(h, mask, 0, d.Tl, true, false, h->M, h->MT);
}

int cmf_pgd_reset(cmf_handle h)
{
    if (!h) return fail(CMF_ERR_ARG, "handle is NULL");
    h->pgd_stepW = h->pgd_stepH = 5.0;
    h->pgd_cur_loss = -1.0;
    return CMF_OK;
}

int cmf_pgd_set_loss(cmf_handle h, int loss_kind)
{
    if (!h) return fail(CMF_ERR_ARG, "handle is NULL");
    if (loss_kind != 0 && loss_kind != 1) return fail(CMF_ERR_ARG, "loss_kind must be 0 (SquareLoss) or 1 (AbsoluteLoss)");
    if (h->pgd_loss_abs != loss_kind) {
        set_est(h, 0);
        if (h->group)
            for (cmf_handle_s *s : h->group->sh) set_est(s, 0);
    }
    h->pgd_loss_abs = loss_kind;
    return CMF_OK;
}

int cmf_pgd_update_motifs(cmf_handle h, double pen_sq, double pen_abs, int nonneg)
{
    if (h && h->group) {
        CMFTRY(group_check_ready(h->group));
        return group_pgd_w(h, h->group, pen_sq, pen_abs, nonneg);
    }
    CMFTRY(check_ready(h, true));
    return pgd_w_impl(h, pen_sq, pen_abs, nonneg);
}

static int pgd_update_feature_maps_body(cmf_handle h, double pen_sq, double pen_abs, int nonneg, double *loss)
{
    if (!loss) return fail(CMF_ERR_ARG, "loss is NULL");
    if (h && h->group) {
        CMFTRY(group_check_ready(h->group));
        CMFTRY(wb_group_post(h)); // (an armed write-back: the helpers wait for every shard's copy of its block of H)
        return group_pgd_h(h, h->group, pen_sq, pen_abs, nonneg, loss);
    }
    CMFTRY(check_ready(h, true));
    return pgd_h_impl(h, pen_sq, pen_abs, nonneg, loss);
}

int cmf_pgd_update_feature_maps(cmf_handle h, double pen_sq, double pen_abs, int nonneg, double *loss)
{
    return wb_finish(h, pgd_update_feature_maps_body(h, pen_sq, pen_abs, nonneg, loss));
}

int cmf_pgd_get_steps(cmf_handle h, double *stepW, double *stepH)
{
    if (!h || !stepW || !stepH) return fail(CMF_ERR_ARG, "NULL argument");
    *stepW = h->pgd_stepW;
    *stepH = h->pgd_stepH;
    return CMF_OK;
}

int cmf_converged(const double *loss_hist, int64_t len, int64_t patience, double tol)
{
    // src/model.jl:91-107
    if (!loss_hist || len <= patience) return 0;
    for (int64_t i = len - patience; i < len; ++i)
        if (!(std::fabs(loss_hist[i] - loss_hist[i - 1]) < tol)) return 0;
    return 1;
}

// n MU iterations on a single-GPU handle without stalling the device between iterations: the loss of iteration i is
// copied to a pinned slot behind its reduction and read after iteration i+1 has been enqueued
static int iterate_single(cmf_handle_s *h, int64_t n, int eval_mode, double l1W, double l2W, double l1H, double l2H,
                          double *losses, double *stamps)
{
    const auto t_begin = std::chrono::steady_clock::now();
    auto now = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
    if (h->gram == 2) { // the Gram-sum loss is read back inside the entry: plain loop
        for (int64_t it = 0; it < n; ++it) {
            if (!eval_mode) CMFTRY(cmf_update_motifs(h, l1W, l2W));
            CMFTRY(cmf_update_feature_maps(h, l1H, l2H, &losses[it]));
            if (stamps) stamps[it] = now();
        }
        return CMF_OK;
    }
    CMFTRY(ensure_ring(h)); // (slots 0, 1: written by the loss reductions of this loop, polled by the host)
    volatile unsigned long long *ring = reinterpret_cast<volatile unsigned long long *>(h->h_ring);
    // cmf_fit's time_hist: a timing event behind every iteration's loss conv, so that entry i is the DEVICE time at which
    // iteration i was complete (what the reference's wall clock around the two rule calls measures, alternating.jl:49,57-58),
    // not the moment its loss reached the host half an iteration later.  An event costs the device a few microseconds per
    // iteration, so only cmf_fit asks for it; cmf_iterate's own stamps stay host times.
    DevStamps ds;
    if (h->dev_stamps && stamps) CMFTRY(ds.begin(h->stream, n));
    auto collect = [&](int64_t it) -> int {
        const int slot = (int)(it & 1);
        CMFTRY(wait_words<unsigned long long>(h->stream, ring + slot, 1, CMF_SENTINEL64));
        const unsigned long long bits = ring[slot];
        double ss;
        std::memcpy(&ss, &bits, 8);
        losses[it] = std::sqrt(ss) / h->data_norm;
        if (stamps) stamps[it] = now();
        return CMF_OK;
    };
    for (int64_t it = 0; it < n; ++it) {
        RoctxRange it_range("cmf:iteration %d", (int)it);
        if (!eval_mode) { // alternating.jl:51-53
            RoctxRange w_range("cmf:W phase (update_motifs!)");
            if (h->gram) {
                CMFTRY(gram_w_impl(h, l1W, l2W));
            } else {
                CMFTRY(w_phase_impl(h, l1W, l2W));
            }
        }
        RoctxRange h_range("cmf:H phase + loss conv (update_feature_maps!)");
        CMFTRY(h->gram ? gram_h_update(h, l1H, l2H) : h_update_impl(h, l1H, l2H)); // :54
        const int slot = (int)(it & 1);
        ring[slot] = CMF_SENTINEL64; // the slot's previous loss was collected an iteration ago
        if (!eval_mode && it + 1 < n) {
            // the per-tile sums are reduced by the next iteration's slab sum (CmfLossCarry), which stores the total into the pinned slot
            CMFTRY(launch_loss_conv(h));
            h->carry = CmfLossCarry{h->partial, h->conv_partials, h->d_scalar, h->h_ring + slot, nullptr, 0, 0};
        } else {
            CMFTRY(loss_partial_impl(h, nullptr, false, h->h_ring + slot)); // the reduction stores the sum into the pinned slot itself
        }
        CMFTRY(ds.mark(h->stream, it));
        if (it > 0) CMFTRY(collect(it - 1));
    }
    if (n > 0) CMFTRY(collect(n - 1));
    return ds.finish(stamps, n);
}

int cmf_iterate(cmf_handle h, int64_t n_iter, int eval_mode, double l1W, double l2W, double l1H, double l2H,
                double *losses, double *stamps)
{
    RoctxRange range("cmf_iterate");
    if (!h || !losses) return fail(CMF_ERR_ARG, "NULL argument");
    if (n_iter < 0) return fail(CMF_ERR_ARG, "n_iter must be >= 0");
    // An armed write-back serves the next *_update_feature_maps CALL only (include/cmf_hip.h): this loop runs the H phases itself, and
    // its hooks would post the helpers into the caller's arrays every iteration with nobody draining them.
    if (h->wb && h->wb->armed) wb_disarm(h);
    drop_carry(h);
    CarryGuard guard{h};
    int rc;
    if (h->group) {
        CMFTRY(group_check_ready(h->group));
        rc = group_iterate(h->group, n_iter, eval_mode, l1W, l2W, l1H, l2H, losses, stamps);
    } else {
        CMFTRY(check_ready(h, true));
        if (h->sharded && h->T_global != h->d.Tl) return fail(CMF_ERR_STATE, "sharded handle: attach a communicator (cmf_comm_init_*) first");
        rc = iterate_single(h, n_iter, eval_mode, l1W, l2W, l1H, l2H, losses, stamps);
    }
    guard.armed = rc != CMF_OK;
    return rc;
}

int cmf_fit(cmf_handle h, int64_t max_itr, double max_time, int check_convergence, int64_t patience, double tol,
            int eval_mode, double l1W, double l2W, double l1H, double l2H,
            double *loss_hist, double *time_hist, int64_t *n_hist, int *converged_early)
{
    if (!h) return fail(CMF_ERR_ARG, "handle is NULL");
    if (!loss_hist || !time_hist || !n_hist) return fail(CMF_ERR_ARG, "NULL output argument");
    if (patience < 1) return fail(CMF_ERR_ARG, "patience must be >= 1 (alternating.jl:30)");
    if (max_itr < 0) return fail(CMF_ERR_ARG, "max_itr must be >= 0");
    if (!h->group) {
        CMFTRY(check_ready(h, true));
        if (h->sharded && h->T_global != h->d.Tl) return fail(CMF_ERR_STATE, "sharded handle: attach a communicator (cmf_comm_init_*) first");
    }
    if (converged_early) *converged_early = 0;
    if (h->wb && h->wb->armed) wb_disarm(h); // (as in cmf_iterate: the loop's own rule calls are not the caller's)
    int64_t len = 0;
    CMFTRY(cmf_compute_loss(h, &loss_hist[0])); // alternating.jl:37
    time_hist[0] = 0.0;                        // :38
    len = 1;
    if (!check_convergence && std::isinf(max_time) && max_time > 0) {
        // neither stop test can fire (:45, :63-66): exactly max_itr iterations, run as one pipelined batch;
        // time_hist[i] = the moment the loss of iteration i reached the host
        cmf_handle_s *hs = h->group ? h->group->sh[0] : h; // (a group's times are shard 0's)
        hs->dev_stamps = true;
        const int rc_it = cmf_iterate(h, max_itr, eval_mode, l1W, l2W, l1H, l2H, loss_hist + 1, time_hist + 1);
        hs->dev_stamps = false;
        CMFTRY(rc_it);
        *n_hist = max_itr + 1;
        return CMF_OK;
    }
    int64_t itr = 1;
    while (itr <= max_itr && time_hist[len - 1] <= max_time) { // :45
        itr += 1;
        auto t0 = std::chrono::steady_clock::now();
        if (!eval_mode) CMFTRY(cmf_update_motifs(h, l1W, l2W));          // :51-53
        double loss = 0.0;
        CMFTRY(cmf_update_feature_maps(h, l1H, l2H, &loss));              // :54
        double dur = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (h->group && !h->group->one_process && std::isfinite(max_time)) {
            // one process per shard: every rank follows rank 0's clock, so they all leave the loop together
            std::vector<double> mine(1, dur), all;
            CMFTRY(group_gather_doubles(h->group, mine, all));
            dur = all[0];
        }
        time_hist[len] = time_hist[len - 1] + dur;                        // :58
        loss_hist[len] = loss;                                            // :59
        ++len;
        if (check_convergence && cmf_converged(loss_hist, len, patience, tol)) { // :63-66
            if (converged_early) *converged_early = 1;
            break;
        }
    }
    *n_hist = len;
    return CMF_OK;
}

// ---- stand-alone primitives ----------------------------------------------------------------
static int download_rows(cmf_handle_s *h, double *out, const float *buf, int row0, int64_t nrows, int width, int stride)
{
    const int64_t chunk = std::max<int64_t>(1, (int64_t)(32u << 20) / 8 / std::max(1, width));
    CMFTRY(ensure_stage(h, (size_t)std::min(chunk, nrows) * width));
    for (int64_t c0 = 0; c0 < nrows; c0 += chunk) {
        int64_t nc = std::min(chunk, nrows - c0);
        hipLaunchKernelGGL(unpack_rows_kernel, dim3(1024), dim3(256), 0, h->stream, h->stage, buf, row0, (int)c0, (int)nc, width, stride);
        KCHK("unpack_rows_kernel");
        HIPCHK(hipMemcpyAsync(out + (size_t)c0 * width, h->stage, (size_t)nc * width * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return CMF_OK;
}

// (both stand-alone primitives go through a matrix longer than one handle reaches -- max_columns_per_handle -- in column
// blocks: tensor_conv looks L-1 columns back, so a block is computed with the L-1 columns in front of it and those are dropped;
// tensor_transconv looks L-1 columns ahead, so a block takes the L-1 columns behind it along)
int cmf_tensor_conv(int device, int64_t N, int64_t T, int64_t K, int64_t L, const double *W, const double *H, double *est)
{
    if (!W || !H || !est) return fail(CMF_ERR_ARG, "NULL argument");
    if (N < 1 || T < 1 || K < 1 || L < 1) return fail(CMF_ERR_ARG, "N, T, K, L must all be >= 1");
    const int64_t per = std::max<int64_t>(max_columns_per_handle(K) - L, 1);
    for (int64_t t0 = 0; t0 < T; t0 += per) {
        const int64_t t1 = std::min(T, t0 + per), skip = std::min<int64_t>(L - 1, t0), Tb = t1 - t0 + skip;
        cmf_handle h = nullptr;
        CMFTRY(create_impl(&h, device, N, Tb, K, L, nullptr, 0, Tb, false));
        int rc = set_factors_impl(h, W, H + (size_t)(t0 - skip) * K);
        if (rc == CMF_OK) rc = launch_conv<0>(h, h->est, h->d.Tl, h->conv_gy);
        if (rc == CMF_OK) rc = download_rows(h, est + (size_t)t0 * N, h->est, h->d.PADL + (int)skip, t1 - t0, (int)N, h->d.Np);
        destroy_impl(h);
        CMFTRY(rc);
    }
    return CMF_OK;
}

int cmf_tensor_transconv(int device, int64_t N, int64_t T, int64_t K, int64_t L, const double *W, const double *X, double *out)
{
    if (!W || !X || !out) return fail(CMF_ERR_ARG, "NULL argument");
    if (N < 1 || T < 1 || K < 1 || L < 1) return fail(CMF_ERR_ARG, "N, T, K, L must all be >= 1");
    const int64_t per = std::max<int64_t>(max_columns_per_handle(K) - L, 1);
    for (int64_t t0 = 0; t0 < T; t0 += per) {
        const int64_t t1 = std::min(T, t0 + per), ahead = std::min<int64_t>(L - 1, T - t1), Tb = t1 - t0 + ahead;
        cmf_handle h = nullptr;
        CMFTRY(create_impl(&h, device, N, Tb, K, L, X + (size_t)t0 * N, 0, Tb, false));
        std::vector<double> H0((size_t)K * Tb, 0.0);
        int rc = set_factors_impl(h, W, H0.data());
        if (rc == CMF_OK) rc = launch_transconv(h, 1);
        // sum the S n-range slabs ([S][1][Tl][K32])
        float *sum = nullptr;
        if (rc == CMF_OK && hipMalloc(&sum, (size_t)h->d.Tl * h->d.K32 * sizeof(float)) != hipSuccess)
            rc = fail(CMF_ERR_HIP, "hipMalloc failed in cmf_tensor_transconv");
        if (rc == CMF_OK) rc = launch_slab_sum(h, sum, h->hslabs, h->tc_S1, (size_t)h->d.Tl * h->d.K32);
        if (rc == CMF_OK) rc = download_rows(h, out + (size_t)t0 * K, sum, 0, t1 - t0, (int)K, h->d.K32);
        if (sum) (void)hipFree(sum);
        destroy_impl(h);
        CMFTRY(rc);
    }
    return CMF_OK;
}

// ---- init_rand / gen_synthetic -----------------------------------------------------------------
static void parallel_for(size_t n, const std::function<void(size_t, size_t)> &fn)
{
    unsigned nt = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    if (n < (1u << 16)) nt = 1;
    std::vector<std::thread> th;
    size_t per = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) {
        size_t a = t * per, b = std::min(n, a + per);
        if (a >= b) break;
        th.emplace_back(fn, a, b);
    }
    for (auto &x : th) x.join();
}

// ---- fingerprints of the caller's arrays (include/cmf_hip.h: cmf_fingerprint) --------------------------------------------------
static inline uint64_t fp_mix(uint64_t x)
{
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    return x ^ (x >> 31);
}
int cmf_fingerprint(const double *a, int64_t n, int64_t line_stride, uint64_t *fp)
{
    if (!a || !fp || n < 0) return fail(CMF_ERR_ARG, "bad argument");
    if (line_stride < 1) line_stride = 1;
    // 64-byte lines of 8 doubles; every line_stride-th line is hashed together with its index, the hashes are added (order-free, so
    // the lines can be dealt to threads), and the last line and the length always count
    const int64_t nlines = (n + 7) / 8;
    const int64_t nsel = (nlines + line_stride - 1) / line_stride;
    auto line_hash = [&](int64_t line) -> uint64_t {
        uint64_t hsh = fp_mix(0x9e3779b97f4a7c15ULL * (uint64_t)(line + 1));
        const int64_t e0 = line * 8, e1 = std::min<int64_t>(n, e0 + 8);
        for (int64_t e = e0; e < e1; ++e) {
            uint64_t bits;
            std::memcpy(&bits, a + e, 8);
            hsh = fp_mix(hsh ^ bits) + 0x2545f4914f6cdd1dULL;
        }
        return hsh;
    };
    std::mutex mu;
    uint64_t total = fp_mix((uint64_t)n);
    parallel_for((size_t)nsel, [&](size_t i0, size_t i1) {
        uint64_t acc = 0;
        for (size_t i = i0; i < i1; ++i) acc += line_hash((int64_t)i * line_stride);
        std::lock_guard<std::mutex> lock(mu);
        total += acc;
    });
    if (nlines > 0 && (nlines - 1) % line_stride != 0) total += line_hash(nlines - 1);
    *fp = total;
    return CMF_OK;
}

int cmf_init_rand(int device, int64_t N, int64_t T, int64_t K, int64_t L, uint64_t seed, const double *data, double *W, double *H)
{
    // src/model.jl:113-125
    if (!data || !W || !H) return fail(CMF_ERR_ARG, "NULL argument");
    if (N < 1 || T < 1 || K < 1 || L < 1) return fail(CMF_ERR_ARG, "N, T, K, L must all be >= 1");
    const size_t nW = (size_t)K * N * L, nH = (size_t)K * T;
    const uint64_t bW = cmfrng::base(seed, 0), bH = cmfrng::base(seed, 1);
    for (size_t i = 0; i < nW; ++i) W[i] = cmfrng::u01(bW, i); // :116 rand(K, N, L)
    for (size_t i = 0; i < nH; ++i) H[i] = cmfrng::u01(bH, i); // :117 rand(K, T)
    // :119-120 on the device: est = tensor_conv(W, H) next to the uploaded data, then <data, est> and norm(est)^2 in one
    // pass over the two (padded, zero-filled) layouts -- no N x T array crosses PCIe back, none is allocated on the host.
    // A recording longer than one handle reaches (max_columns_per_handle) goes through in column blocks: each block is
    // uploaded with the L-1 columns in front of it, whose est is formed (they complete the lag windows of the block's first
    // columns) but left out of the two sums.
    double dot = 0.0, nn = 0.0;
    const int64_t per = std::max<int64_t>(max_columns_per_handle(K) - L, 1);
    for (int64_t t0 = 0; t0 < T; t0 += per) {
        const int64_t t1 = std::min(T, t0 + per), skip = t0 > 0 ? std::min<int64_t>(L - 1, t0) : 0;
        const int64_t Tb = t1 - t0 + skip;
        cmf_handle hh = nullptr;
        CMFTRY(create_impl(&hh, device, N, Tb, K, L, data + (size_t)(t0 - skip) * N, 0, Tb, false));
        const CmfDims &d = hh->d;
        const int nb = 1024;
        double *part = nullptr;
        double dot_b = 0.0, nn_b = 0.0;
        int rc = set_factors_impl(hh, W, H + (size_t)(t0 - skip) * K);
        if (rc == CMF_OK) rc = launch_conv<0>(hh, hh->est, d.Tl, hh->conv_gy);
        if (rc == CMF_OK && hipMalloc(&part, (size_t)2 * nb * sizeof(double)) != hipSuccess) rc = fail(CMF_ERR_HIP, "hipMalloc failed in cmf_init_rand");
        if (rc == CMF_OK) {
            const size_t off = (size_t)(d.PADL + skip) * d.Np, n4 = (size_t)(d.Tl - skip) * d.Np / 4; // the block's own rows, all columns
            hipLaunchKernelGGL(init_dot_kernel, dim3(nb), dim3(256), 0, hh->stream, hh->est + off, hh->X + off, n4, part);
            hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, hh->stream, part, nb, hh->d_scalar + 2, (double *)nullptr);
            hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, hh->stream, part + nb, nb, hh->d_scalar + 3, (double *)nullptr);
            if (hipGetLastError() != hipSuccess) rc = fail(CMF_ERR_HIP, "launch failed in cmf_init_rand");
        }
        if (rc == CMF_OK) rc = read_scalar(hh, 2, &dot_b);
        if (rc == CMF_OK) rc = read_scalar(hh, 3, &nn_b);
        if (part) (void)hipFree(part);
        destroy_impl(hh);
        CMFTRY(rc);
        dot += dot_b;
        nn += nn_b;
    }
    const double s = std::sqrt(std::fabs(dot / nn)); // :121-122
    for (size_t i = 0; i < nW; ++i) W[i] *= s;
    for (size_t i = 0; i < nH; ++i) H[i] *= s;
    return CMF_OK;
}

int cmf_gen_synthetic(int device, int64_t N, int64_t T, int64_t K, int64_t L, double alpha, double p_h, double sigma,
                      double noise_scale, uint64_t seed, double *data, double *Wout, double *Hout)
{
    // datasets/synthetic.jl:29-61
    if (!data) return fail(CMF_ERR_ARG, "data is NULL");
    if (N < 1 || T < 1 || K < 1 || L < 1) return fail(CMF_ERR_ARG, "N, T, K, L must all be >= 1");
    if (!(alpha > 0.0) || !(sigma > 0.0)) return fail(CMF_ERR_ARG, "alpha and sigma must be > 0");
    const uint64_t bga = cmfrng::base(seed, 10), bgb = cmfrng::base(seed, 11), bgu = cmfrng::base(seed, 12);
    const uint64_t bc = cmfrng::base(seed, 13), be = cmfrng::base(seed, 14), bb = cmfrng::base(seed, 15);
    const uint64_t bna = cmfrng::base(seed, 16), bnb = cmfrng::base(seed, 17);
    std::vector<double> W((size_t)K * N * L), H((size_t)K * T), mW((size_t)N * K);
    for (int64_t n = 0; n < N; ++n) { // :42 Dirichlet(alpha) rows
        double s = 0.0;
        for (int64_t k = 0; k < K; ++k) { double g = cmfrng::gamma(bga, bgb, bgu, (uint64_t)(n * K + k), alpha); mW[n * K + k] = g; s += g; }
        if (!(s > 0.0)) { for (int64_t k = 0; k < K; ++k) mW[n * K + k] = (k == n % K) ? 1.0 : 0.0; s = 1.0; }
        for (int64_t k = 0; k < K; ++k) mW[n * K + k] /= s;
    }
    const double inv_s2pi = 0.39894228040143267793994605993438;
    for (int64_t k = 0; k < K; ++k) // :47-51 Gaussian bump at a random lag
        for (int64_t n = 0; n < N; ++n) {
            const double cent = -1.0 + 2.0 * cmfrng::u01(bc, (uint64_t)(k + K * n));
            for (int64_t l = 0; l < L; ++l) {
                const double x = (L > 1) ? (-1.0 + 2.0 * (double)l / (double)(L - 1)) : -1.0;
                const double z = (x - cent) / sigma;
                W[k + K * (n + N * l)] = mW[n * K + k] * (inv_s2pi / sigma) * std::exp(-0.5 * z * z);
            }
        }
    for (size_t i = 0; i < (size_t)K * T; ++i) { // :54
        const double e = -std::log(cmfrng::u01o(be, i));
        H[i] = (cmfrng::u01(bb, i) < p_h) ? e : 0.0;
    }
    CMFTRY(cmf_tensor_conv(device, N, T, K, L, W.data(), H.data(), data)); // :58 (fp32 on the device)
    parallel_for((size_t)N * T, [&](size_t a, size_t b) {                  // :57-58
        for (size_t i = a; i < b; ++i) {
            const double v = data[i] + noise_scale * cmfrng::normal(bna, bnb, i);
            data[i] = v > 0.0 ? v : 0.0;
        }
    });
    if (Wout) std::memcpy(Wout, W.data(), W.size() * sizeof(double));
    if (Hout) std::memcpy(Hout, H.data(), H.size() * sizeof(double));
    return CMF_OK;
}

// ---- measurement ---------------------------------------------------------------------------------
int cmf_kernel_times(cmf_handle h, const char *name, double *avg_ms, int64_t *launches)
{
    if (!h || !name || !avg_ms || !launches) return fail(CMF_ERR_ARG, "NULL argument");
    if (h->root_only) h = h->group->sh[0]; // shard 0 stands for the group
    HIPCHK(hipSetDevice(h->device));
    int cls = -1;
    for (int c = 0; c < PROF_NCLS; ++c)
        if (std::strcmp(name, kProfNames[c]) == 0) cls = c;
    if (cls < 0) return fail(CMF_ERR_ARG, "unknown kernel class '%s'", name);
    HIPCHK(hipStreamSynchronize(h->stream));
    double total = 0.0;
    int64_t n = 0;
    for (auto &r : h->prof_recs) {
        if (r.cls != cls) continue;
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, r.a, r.b));
        total += ms;
        ++n;
    }
    *avg_ms = n ? total / (double)n : 0.0;
    *launches = n;
    return CMF_OK;
}

int cmf_time_kernel(cmf_handle h, const char *name, int reps, double *avg_ms, double *flops)
{
    if (!name || !avg_ms || !flops || reps < 1) return fail(CMF_ERR_ARG, "bad argument");
    if (h && h->group && (std::strcmp(name, "allreduce") == 0 || std::strcmp(name, "allreduce_gram") == 0 || std::strcmp(name, "allreduce_lane1") == 0 ||
                          std::strcmp(name, "allgather_halo") == 0)) {
        // The group's exchanges alone, timed with events on shard 0's stream: "allreduce" = `reps` all-reduces of the buffer an
        // iteration sends ([numW | denomW | tail], or [numW | HH | tail] when the Gram form is on), "allreduce_gram" = the Gram
        // form's payload whatever the option says, "allreduce_lane1" = the overlap form's bulk share on the communication
        // stream and its own communicator, "allgather_halo" = the (L-1)-column H halo all-gather.  EVERY rank of the group must
        // make this call (it is a collective).  *flops receives the payload in bytes.  The buffers are scratch between iterations.
        cmf_group_s *g = h->group;
        CMFTRY(group_check_ready(g));
        GroupInline scope(g);
        CMFTRY(scope.rc);
        CMFTRY(group_sync(g));
        cmf_handle_s *s0 = g->sh[0];
        const bool halo = std::strcmp(name, "allgather_halo") == 0, lane1 = std::strcmp(name, "allreduce_lane1") == 0;
        const size_t half = (size_t)g->LKN2 / 2;
        size_t count = group_tail_off(g) + (size_t)g->tail;
        if (std::strcmp(name, "allreduce_gram") == 0) count = half + (size_t)g->HHsz + (size_t)g->tail;
        if (lane1) {
            count = g->gram ? half + (size_t)g->HHsz : half;
            CMFTRY(group_ensure_lane1(g));
        }
        if (halo) count = (size_t)(2 * g->HC);
        auto once = [&]() -> int { return halo ? group_allgather(g, g->halo_send, g->halo_all, count) : group_allreduce(g, g->red, 0, count, lane1 ? 1 : 0); };
        for (size_t i = 0; i < g->sh.size() && !halo; ++i) { // finite input: the sums of `reps` all-reduces of zeros stay zeros
            CMFTRY(group_use(g->sh[i]));
            HIPCHK(hipMemsetAsync(g->red[i], 0, count * sizeof(float), lane1 ? g->sh[i]->comm_stream : g->sh[i]->stream));
        }
        hipStream_t ts = lane1 ? s0->comm_stream : s0->stream;
        CMFTRY(once()); // warm-up
        CMFTRY(group_use(s0));
        HIPCHK(hipEventRecord(s0->ev0, ts));
        for (int r = 0; r < reps; ++r) CMFTRY(once());
        CMFTRY(group_use(s0));
        HIPCHK(hipEventRecord(s0->ev1, ts));
        CMFTRY(group_sync(g));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, s0->ev0, s0->ev1));
        *avg_ms = (double)ms / reps;
        *flops = (double)count * sizeof(float);
        g->num_ready = false;
        g->halos_current = g->halos_pending = g->halo_wide = false; // (the buffers are scratch here: halo slots waiting in the tail are gone -- the next phase exchanges afresh)
        return CMF_OK;
    }
    if (h && h->root_only) h = h->group->sh[0]; // shard 0 stands for the group
    CMFTRY(check_ready(h, true));
    const CmfDims &d = h->d;
    const double S = (double)d.L * d.Tl - 0.5 * (double)d.L * (d.L - 1);
    const double f1 = 2.0 * d.K * d.N * S; // one contraction
    std::string nm(name);
    if (nm == "pair_conv_tc" || nm == "seq_conv_tc" || nm == "pair_loss_hxt" || nm == "seq_loss_hxt") {
        // Experiment: two INDEPENDENT contractions of an iteration -- conv_t and numH = transconv(W, data),
        // or the loss conv and numW = hxt(data) -- on two streams at once ("pair_") against one after the other ("seq_"):
        // does the second kernel fill the first one's drain, or does sharing the chip break their static work splits?
        const bool pair = nm[0] == 'p', first = nm.find("conv_tc") != std::string::npos;
        hipStream_t aux = nullptr, keep = h->stream;
        hipEvent_t fork = nullptr, join = nullptr;
        HIPCHK(hipStreamCreateWithFlags(&aux, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
        auto once = [&]() -> int {
            if (pair) {
                HIPCHK(hipEventRecord(fork, keep));
                HIPCHK(hipStreamWaitEvent(aux, fork, 0));
            }
            CMFTRY(first ? launch_conv<1>(h, h->estT, d.Tl + h->halo_r, h->conv_gy_ext) : launch_conv<3>(h, h->est, d.Tl, h->conv_gy));
            h->stream = pair ? aux : keep;
            int rc = first ? launch_transconv(h, 1, h->XT)
                           : h->small_k ? hxt_contract(h, h->X, h->X, 1, h->numden) // (few components: kernel + its slab sum)
                                        : launch_hxt_on(h, h->X, h->X, d.Np, 1, h->wslabs, h->hxt_nchunks1, h->hxt_chunk_len1, h->hxt_main1);
            h->stream = keep;
            CMFTRY(rc);
            if (pair) {
                HIPCHK(hipEventRecord(join, aux));
                HIPCHK(hipStreamWaitEvent(keep, join, 0));
            }
            return CMF_OK;
        };
        int rc = once();
        if (rc == CMF_OK) {
            HIPCHK(hipEventRecord(h->ev0, keep));
            for (int r = 0; r < reps && rc == CMF_OK; ++r) rc = once();
            HIPCHK(hipEventRecord(h->ev1, keep));
            HIPCHK(hipEventSynchronize(h->ev1));
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
            *avg_ms = (double)ms / reps;
            *flops = 2.0 * f1;
        }
        (void)hipStreamSynchronize(aux);
        (void)hipStreamDestroy(aux);
        (void)hipEventDestroy(fork);
        (void)hipEventDestroy(join);
        set_est(h, 0);
        return rc;
    }
    if (nm.rfind("hals_overlap", 0) == 0) {
        // Bound experiment: the HALS H row pipeline (K + (K-1)P workgroups, VALU only) with the first
        // pct % of the residual conv's tile rows on a second, CU-masked stream AT THE SAME TIME -- no dependency between them,
        // timing only, the handle's state is garbage afterwards -- and the rest of the conv on the whole chip behind both.
        // "hals_overlap:<pct>:<mode>": mode 0 = pipeline on a stream masked to 20 CUs of every XCD, conv part on the other 12;
        // 1 = pipeline on the handle's own (unmasked) stream, only the conv part masked; pct = 0: the two kernels back to back.
        int pct = 40, mode = 0;
        (void)sscanf(name, "hals_overlap:%d:%d", &pct, &mode);
        CMFTRY(hals_ensure(h));
        if (h->hals_pullers <= 0) return fail(CMF_ERR_STATE, "hals_overlap needs the persistent H pipeline");
        uint32_t mask_a[8] = {0}, mask_b[8] = {0};
        for (int j = 0; j < h->n_cu && j < 256; ++j) (((j / 8) < 20) ? mask_a : mask_b)[j / 32] |= 1u << (j % 32);
        hipStream_t sa = nullptr, sb = nullptr, keep = h->stream;
        hipEvent_t fork = nullptr, ja = nullptr, jb = nullptr;
        HIPCHK(hipExtStreamCreateWithCUMask(&sa, 8, mask_a));
        HIPCHK(hipExtStreamCreateWithCUMask(&sb, 8, mask_b));
        HIPCHK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&ja, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&jb, hipEventDisableTiming));
        const HalsRowParams q = hals_row_params(h, 0.0, 0.0);
        const int rows_t = (d.Tl + 63) / 64, ra = std::min(rows_t, std::max(0, rows_t * pct / 100));
        auto once = [&]() -> int {
            if (ra == 0) {
                CMFTRY(hals_persist_launch(h, q));
                return launch_conv<4>(h, h->est, d.Tl, h->conv_gy);
            }
            HIPCHK(hipEventRecord(fork, keep));
            HIPCHK(hipStreamWaitEvent(sb, fork, 0));
            if (mode == 0) HIPCHK(hipStreamWaitEvent(sa, fork, 0));
            h->stream = mode == 0 ? sa : keep;
            int rc = hals_persist_launch(h, q);
            h->stream = sb;
            if (rc == CMF_OK) rc = launch_conv<4>(h, h->est, std::min(d.Tl, ra * 64), h->conv_gy);
            h->stream = keep;
            CMFTRY(rc);
            if (mode == 0) { HIPCHK(hipEventRecord(ja, sa)); HIPCHK(hipStreamWaitEvent(keep, ja, 0)); }
            HIPCHK(hipEventRecord(jb, sb));
            HIPCHK(hipStreamWaitEvent(keep, jb, 0));
            if (ra < rows_t) CMFTRY(launch_conv<4>(h, h->est, d.Tl - ra * 64, h->conv_gy));
            return CMF_OK;
        };
        int rc = once();
        if (rc == CMF_OK) {
            HIPCHK(hipEventRecord(h->ev0, keep));
            for (int r = 0; r < reps && rc == CMF_OK; ++r) rc = once();
            HIPCHK(hipEventRecord(h->ev1, keep));
            HIPCHK(hipEventSynchronize(h->ev1));
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
            *avg_ms = (double)ms / reps;
            *flops = f1;
        }
        h->stream = keep;
        (void)hipStreamSynchronize(sa);
        (void)hipStreamSynchronize(sb);
        (void)hipStreamDestroy(sa);
        (void)hipStreamDestroy(sb);
        (void)hipEventDestroy(fork);
        (void)hipEventDestroy(ja);
        (void)hipEventDestroy(jb);
        if (h->hals_status) *h->hals_status = 0;
        set_est(h, 0);
        return rc;
    }
    int which = nm == "conv" ? 0 : nm == "hxt" ? 1 : nm == "transconv" ? 2 : nm == "conv_t" ? 3 : nm == "conv_loss" ? 4
              : nm == "conv_loss_store" ? 5 : -1;
    if (which < 0) return fail(CMF_ERR_ARG, "unknown kernel '%s'", name);
    auto run = [&]() -> int {
        switch (which) {
        case 0: return launch_conv<0>(h, h->est, d.Tl, h->conv_gy);
        case 1: return h->small_k ? hxt_contract(h, h->X, h->est, 2, h->numden) : launch_hxt(h); // (few components: kernel + its slab sum)
        case 2: return launch_transconv(h, 2);
        case 3: return launch_conv<1>(h, h->estT, d.Tl + h->halo_r, h->conv_gy_ext);
        case 4: return launch_conv<2>(h, nullptr, d.Tl, h->conv_gy);
        default: return launch_conv<3>(h, h->est, d.Tl, h->conv_gy);
        }
    };
    CMFTRY(run()); // warm-up
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    for (int r = 0; r < reps; ++r) CMFTRY(run());
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *avg_ms = (double)ms / reps;
    if (which == 0 || which == 5) set_est(h, 1); // est now holds plain tensor_conv(W, H), whatever it held before (a residual on HALS / PGD handles)
    *flops = (which == 1 || which == 2) ? 2.0 * f1 : f1;
    return CMF_OK;
}

