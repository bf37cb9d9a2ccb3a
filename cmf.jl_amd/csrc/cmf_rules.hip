// cmf_rules.hip -- the HALS rule (src/algs/hals.jl), the optional Gram form of the MU iteration and the PGD rule (src/algs/pgd.jl) on one handle.
#include "cmf_internal.h"

// est := tensor_conv(W,H) - data (the residual hals.jl / pgd.jl carry), with the loss sum in d_scalar[0]
int resid_and_loss(cmf_handle_s *h, double *sumsq, bool masked, bool loss_abs)
{
    const CmfDims &d = h->d;
    h->pgd_loss_abs_now = loss_abs ? 1 : 0;
    int rc = masked ? launch_conv<6>(h, h->est, d.Tl, h->conv_gy) // pgd.jl:64-70
                    : launch_conv<4>(h, h->est, d.Tl, h->conv_gy);
    h->pgd_loss_abs_now = 0;
    CMFTRY(rc);
    set_est(h, 2 + (masked ? 1 : 0) + (loss_abs ? 2 : 0));
    return reduce_partials(h, h->partial, h->conv_partials, 0, sumsq);
}

int ensure_resid(cmf_handle_s *h, bool masked, bool loss_abs)
{
    return h->est_kind == 2 + (masked ? 1 : 0) + (loss_abs ? 2 : 0) ? CMF_OK : resid_and_loss(h, nullptr, masked, loss_abs);
}

// ---- HALS (src/algs/hals.jl) -------------------------------------------------------------------
// Scratch that the Gram form of the MU rule and the HALS rule share: H as its own X operand and the lag correlations
// (compute_hh), HH, the lag-Gram taps of W (PW -> GW, GE, GWt).  Shard-aware: only the shard that holds the global right
// edge has truncated lag windows (edge taps GE, cut terms of HH).
int gram_ensure(cmf_handle_s *h)
{
    if (h->gram_ready) return CMF_OK;
    const CmfDims &d = h->d;
    const int E = 2 * d.L - 1;
    const bool has_edge = !h->sharded || h->t_offset + d.Tl == h->T_global;
    h->hals_NpH = (int)rup((int64_t)d.L * d.K32, 128);
    h->hals_t_edge0 = has_edge ? std::max(0, d.Tl - d.L + 1) : d.Tl;
    h->hals_ne = d.Tl - h->hals_t_edge0;
    h->hals_NpC = (int)rup(d.K32, 128); // pitch of H as the X operand of its own lag correlations (compute_hh)
    if ((double)d.L * d.K32 * h->hals_NpH * 4.0 >= 2147483648.0 || (double)d.TP * h->hals_NpC * 4.0 >= 2147483648.0)
        return fail(CMF_ERR_UNSUPPORTED, "Gram form: (L*K)^2 or T*K exceeds the 2 GiB the kernels' 32-bit buffer offsets address");
    {   // time rows per workgroup of the lag-correlation kernel (hals_corr_kernel): about one workgroup per CU and (a, b) block pair,
        // whole steps of four rows
        const int G = std::max(1, h->n_cu / (d.KB * d.KB));
        int64_t R = std::max<int64_t>(4, rup((d.Tl + G - 1) / G, 4));
        const int64_t fit = ((int64_t)64 * 1024 / (d.K32 * 4) - (d.L - 1) - 4) / 4 * 4; // rows of H (+ L - 1 in front, 4 behind) in 64 KB of LDS
        if (fit < 4) return fail(CMF_ERR_UNSUPPORTED, "Gram form / HALS: K * L too large for the lag-correlation kernel's LDS window");
        h->hals_corr_R = (int)std::min(R, fit);
    }
    const size_t LKN = (size_t)d.L * d.K32 * h->hals_NpH;
    const size_t LKC = (size_t)d.L * d.K32 * h->hals_NpC;
    CMFTRY(dalloc_zero(&h->hals_cslabs, (size_t)((d.Tl + h->hals_corr_R - 1) / h->hals_corr_R) * d.L * d.K32 * d.K32)); // one compact slab per workgroup
    CMFTRY(dalloc_zero(&h->hals_C, LKC));
    CMFTRY(dalloc_zero(&h->hals_HH, LKN));
    CMFTRY(dalloc_zero(&h->hals_PW, (size_t)d.L * d.L * d.K32 * d.K32));
    CMFTRY(dalloc_zero(&h->hals_GW, (size_t)d.K32 * d.K32 * E));
    CMFTRY(dalloc_zero(&h->hals_GE, (size_t)d.K32 * std::max(1, h->hals_ne) * d.K32 * E));
    CMFTRY(dalloc_zero(&h->hals_GWt, (size_t)d.K32 * (E + 1) * d.K32)); // the full-window taps as [k'][e][k] for gram_h_mfma_kernel
    h->gram_ready = true;
    return CMF_OK;
}

// Which sweep kernels run and how (re-planned whenever one of the "hals_*" options changes; no allocation here).
// The reference takes any K, L (hals.jl:90-154).  The fast on-chip sweeps have shape limits: the H sweep slides a 64-column
// window along a row with the L-1 pending columns in the lanes of one wave (L <= 64); the W sweep keeps the L*Kpad projected
// state of a unit in registers (up to 32 slots per lane) and K*L new values per unit in LDS.  Outside them the general sweeps
// run (hals_w_sweep_gen_kernel / hals_h_row_gen_kernel): the same recurrences in the same order with the state in LDS /
// global memory -- slower, no shape limit.
void hals_plan(cmf_handle_s *h)
{
    const CmfDims &d = h->d;
    const int E = 2 * d.L - 1;
    h->hals_w_general = ((int64_t)d.L * d.K32 > 2048) || ((size_t)4 * d.K * d.L * HALS_NG * sizeof(float) > 64 * 1024) || (h->hals_opt_general & 1);
    h->hals_h_general = d.L > 64 || (h->hals_opt_general & 2);
    // stage pipeline: segment length (multiple of 64, >= 256 so that the sweeps and pushes of one stage touch disjoint columns: see
    // hals_h_stage_kernel); measured at config 5: 8.60 ms (256), 8.13 (320 and 384), 8.56 (512)
    h->hals_seg = (int)rup(std::max(h->hals_opt_seg, 256), 64);
    h->hals_nseg = (d.Tl + h->hals_seg - 1) / h->hals_seg;
    // persistent H pipeline (hals_h_persist_kernel): K sweepers + (K-1) * P pullers, one workgroup per CU, all resident
    int P = 0;
    if (h->hals_opt_persist != 0 && !h->hals_h_general) {
        P = d.K > 1 ? std::min(HALS_PMAX, (h->n_cu - d.K) / (d.K - 1)) : 1;
        if (h->hals_opt_persist > 1) P = std::min(P, h->hals_opt_persist);
        const size_t lds = ((size_t)(d.K - 1) * (E + 64 + 2 * (d.L - 1)) + 1024) * sizeof(float);
        if (P < 2 && d.K > 1) P = 0; // too many rows for the chip: stage pipeline
        if (lds > 120 * 1024) P = 0;
        // the grid's workgroups wait for each other: all of them must be resident at once.  Ask the runtime how many
        // 1024-thread workgroups with this much LDS a CU takes instead of assuming one (a device with fewer usable CUs, or
        // a kernel whose registers no longer allow 1024 threads, would otherwise only show as an expired wait).
        while (P >= (d.K > 1 ? 2 : 1)) {
            int per_cu = 0;
            const size_t lds_run = std::max((size_t)(d.K - 1) * (2 * d.L - 1 + 64 + 2 * (d.L - 1)) + 1024, (size_t)h->hals_ne * (d.L + 1)) * sizeof(float);
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, hals_h_persist_kernel, 1024, lds_run) != hipSuccess) per_cu = 0;
            if ((long long)per_cu * h->n_cu >= (long long)d.K + (long long)(d.K - 1) * P) break;
            --P;
        }
        if (P < 2 && d.K > 1) P = 0;
        if (P < 1) P = 0;
        // With the residual conv chasing the pipeline (hals_chase_rows) a CU is worth more on the conv's side: three pullers per row
        // instead of four cost the pipeline 1 % (1.44 -> 1.455 ms at config 5; two: 2.07 ms) and give the chasing launch 128 CUs
        // instead of 96 (profiles/r06_hals_chase.txt: 4.35 -> 4.27 ms per iteration)
        if (P == 4 && h->hals_opt_persist == 1 && h->hals_opt_chase != 0 && d.K % 32 == 0 && !h->small_k && d.Tl >= 4096) P = 3;
    }
    h->hals_pullers = P;
}

int hals_ensure(cmf_handle_s *h)
{
    if (h->hals_ready) return CMF_OK;
    const CmfDims &d = h->d;
    if (h->sharded && h->T_global != d.Tl) return fail(CMF_ERR_STATE, "HALS needs an unsharded handle (the H sweep is sequential along T)");
    CMFTRY(gram_ensure(h));
    h->hals_TPp = (int)rup(d.Tl, 64) + (int)std::max<int64_t>(256, rup(d.L, 64) + 128); // (the general row sweep reads a ring of roundup(L, 64) + 64 columns ahead)
    CMFTRY(dalloc_zero(&h->hals_PT, (size_t)d.K32 * h->hals_TPp));
    CMFTRY(dalloc_zero(&h->hals_D, (size_t)d.K32 * h->hals_TPp)); // per row: rows run concurrently
    {   // the persistent pipeline's flags (for the largest puller count a plan may choose) and its status word
        const size_t nflags = (size_t)(d.K + d.K * HALS_PMAX + 1) * HALS_FLAG_STRIDE;
        HIPCHK(hipMalloc((void **)&h->hals_flags, nflags * sizeof(int)));
        HIPCHK(hipHostMalloc((void **)&h->hals_status, sizeof(int), hipHostMallocDefault));
        *h->hals_status = 0;
    }
    hals_plan(h);
    h->hals_ready = true;
    return CMF_OK;
}

// denomW = HH * W (gram_w_kernel): MB p blocks per workgroup, chosen so that the grid is about one workgroup per CU
static int launch_gram_w(cmf_handle_s *h, const float *HH, float *out)
{
    const CmfDims &d = h->d;
    const int nbp = d.L * d.KB, nbn = d.Np / 32;
    int MB = 1;
    for (int m = 5; m >= 2; --m)
        if (nbp % m == 0 && (nbp / m) * nbn >= h->n_cu) { MB = m; break; }
    const dim3 grid(nbn, nbp / MB);
    const size_t lds = (size_t)4 * MB * 16 * 64 * sizeof(float);
    switch (MB) {
#define CASE(M_) case M_: hipLaunchKernelGGL((gram_w_kernel<M_>), grid, dim3(256), lds, h->stream, HH, h->Wt, out, d.L * d.K32, h->hals_NpH, d.Np); break;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5)
#undef CASE
    }
    KCHK("gram_w_kernel");
    return CMF_OK;
}

// HH = H_unfold * H_unfold' (hals.jl:56-60: the row norms are its diagonal) from the lag correlations of H with itself:
// one C2 contraction on K32 columns, then an assembly pass with the right-end corrections (hals_hh_kernel)
static int compute_hh(cmf_handle_s *h, float *out = nullptr)
{
    const CmfDims &d = h->d;
    const bool shard = h->sharded && h->T_global != d.Tl; // out = this shard's additive share of HH (hals_hh_kernel)
    {   // the lag correlations C[d][a][b] = sum_t H[t - d][a] H[t][b] over the handle's own columns (hals_corr_kernel), slabs added in order
        ProfScope prof_(h, PROF_HXT_HH);
        const int R = h->hals_corr_R, G = (d.Tl + R - 1) / R;
        const size_t lds = (size_t)(R + d.L - 1 + 4) * d.K32 * sizeof(float);
        hipLaunchKernelGGL(hals_corr_kernel, dim3(G, d.KB * d.KB), dim3(256), lds, h->stream, h->H, h->hals_cslabs, d.K32, d.KB, d.PADL, d.Tl, d.L, R);
        KCHK("hals_corr_kernel");
        hipLaunchKernelGGL(hals_corr_sum_kernel, dim3((unsigned)(((size_t)d.L * d.K32 * d.K32 + 63) / 64)), dim3(1024), 0, h->stream,
                           h->hals_cslabs, h->hals_C, G, d.L, d.K32, h->hals_NpC);
        KCHK("hals_corr_sum_kernel");
    }
    hipLaunchKernelGGL(hals_hh_kernel, dim3(1024), dim3(256), 0, h->stream, h->hals_C, h->H, out ? out : h->hals_HH, d.Tl, d.L, d.K, d.K32,
                       h->hals_NpC, h->hals_NpH, d.PADL, shard ? 1 : 0, (h->t_offset + d.Tl == h->T_global) ? 1 : 0);
    KCHK("hals_hh_kernel");
    return CMF_OK;
}

// Behind the loss reduction of update_feature_maps! (the residual is current, H final): G = resid * H_unfold' and HH of the NEXT
// update_motifs! (hals.jl:56-60, 104-110), like the MU rule's w_speculate.  Taken by hals_w_impl if nothing has passed set_est since.
int hals_w_speculate(cmf_handle_s *h)
{
    if (h->est_kind != 2 || h->hals_gram == 1 || h->group) return CMF_OK;
    CMFTRY(hxt_contract(h, h->est, h->est, 1, h->numden));
    CMFTRY(compute_hh(h));
    h->hals_spec_gen = h->est_gen;
    return CMF_OK;
}

int hals_w_impl(cmf_handle_s *h, double l1W, double l2W)
{
    const CmfDims &d = h->d;
    CMFTRY(hals_ensure(h));
    // G = resid * H_unfold' (hals.jl:104-110 needs resid * h).  resid = est - data, so G = denomW - numW of the MU path:
    // numW = H_shift * data' is ONE C2 contraction on the data, denomW = H_shift * est' = HH * W a small GEMM on the Gram
    // matrix the sweep needs anyway -- no residual in this phase; but the difference of two quantities several times its
    // size carries ~20x the rounding error through the 640 dependent column updates of the sweep (parity tests), so this
    // is opt-in (hals_gram = 1) and the default contracts G from the residual the loss conv stores anyway.
    const size_t LKN = (size_t)d.L * d.K32 * d.Np;
    const float *G = h->numden, *Gsub = nullptr;
    if (h->hals_gram == 1) {
        CMFTRY(hxt_contract(h, h->X, h->X, 1, h->numden));
        CMFTRY(compute_hh(h));
        CMFTRY(launch_gram_w(h, h->hals_HH, h->numden + LKN));
        G = h->numden + LKN;
        Gsub = h->numden;
    } else {
        // (both already enqueued behind the loss of the update_feature_maps! before, for exactly this state: hals_w_speculate)
        const bool spec = h->hals_spec_gen >= 0 && h->hals_spec_gen == h->est_gen && h->est_kind == 2;
        h->hals_spec_gen = -1;
        if (spec) h->spec_hits += 1;
        else {
            CMFTRY(ensure_resid(h));
            CMFTRY(hxt_contract(h, h->est, h->est, 1, h->numden));
            CMFTRY(compute_hh(h));
        }
    }
    // the K*L sequential column updates, k outer / lag inner (hals.jl:90-97)
    if (h->hals_w_general) { // L * Kpad beyond the register-resident sweep: one workgroup per unit, the state in LDS
        const size_t lds_g = (size_t)d.L * d.K32 * sizeof(float);
        if (lds_g > 64 * 1024) return fail(CMF_ERR_UNSUPPORTED, "HALS W sweep: L * Kpad = %d exceeds the 16384 state entries of a workgroup's LDS", d.L * d.K32);
        ProfScope prof_(h, PROF_HALS_WSWEEP);
        hipLaunchKernelGGL(hals_w_sweep_gen_kernel, dim3(d.N), dim3(256), lds_g, h->stream, h->Wt, h->Wn, G, Gsub, h->hals_HH,
                           d.N, d.K, d.L, d.Np, d.K32, h->hals_NpH, (float)l1W, (float)l2W);
        KCHK("hals_w_sweep_gen_kernel");
        set_est(h, 0);
        return CMF_OK;
    }
    const int nq = (d.L * d.K32 + 63) / 64; // <= 32 here (hals_ensure)
    dim3 grid((d.N + 4 * HALS_NG - 1) / (4 * HALS_NG)), block(256);
    const size_t lds = (size_t)4 * d.K * d.L * HALS_NG * sizeof(float); // <= 64 KB
    ProfScope prof_(h, PROF_HALS_WSWEEP);
#define SWEEP(NQ_, WD_)                                                                                                     \
    hipLaunchKernelGGL((hals_w_sweep_reg_kernel<NQ_, WD_>), grid, block, lds, h->stream, h->Wt, h->Wn, G, Gsub, h->hals_HH, \
                   d.N, d.K, d.L, d.Np, d.K32, h->hals_NpH, (float)l1W, (float)l2W)
    if (nq <= 2) SWEEP(2, 8);
    else if (nq <= 4) SWEEP(4, 8);
    else if (nq <= 6) SWEEP(6, 8);
    else if (nq <= 8) SWEEP(8, 8);
    else if (nq <= 10) SWEEP(10, 8);
    else if (nq <= 12) SWEEP(12, 8);
    else if (nq <= 16) SWEEP(16, 8);
    else if (nq <= 20) SWEEP(20, 4);
    else if (nq <= 24) SWEEP(24, 4);
    else SWEEP(32, 4);
#undef SWEEP
    KCHK("hals_w_sweep_reg_kernel");
    set_est(h, 0);
    return CMF_OK;
}

// P = transconv(W, resid) (hals.jl:152 needs <W_k window, resid window>) = denomH - numH of the MU path:
// numH = transconv(W, data) is ONE C3 contraction on the data, denomH = transconv(W, conv(W, H)) comes from the lag-Gram
// taps applied to H (gram_h_kernel) -- no transposed residual, i.e. one conv launch (1 ms) less per iteration, and the
// H sweep's short recurrences (L-1 columns) do not amplify the cancellation: H stays within the residual form's test
// bars (1.8e-5 against the fp64 restatement where the residual form has 1.2e-5).  hals_gram = 0: P as one C3 contraction
// on the transposed residual.  `contract` = false repeats only the last step (P from the slabs that are still there).
// snapshot: the launch that forms P also takes the persistent pipeline's snapshot of H / H' and clears its flags (hals_p_init_kernel).
static int hals_h_project(cmf_handle_s *h, bool contract, bool snapshot = false)
{
    const CmfDims &d = h->d;
    const size_t TK = (size_t)d.Tl * d.K32;
    float *snap = nullptr;
    int *flags = nullptr;
    int nflags = 0;
    if (snapshot) {
        if (!h->hals_snap) CMFTRY(dalloc_zero(&h->hals_snap, (size_t)2 * d.TP * d.K32)); // (zero like the padding of H and H': only own columns are copied)
        snap = h->hals_snap;
        flags = h->hals_flags;
        nflags = (int)((size_t)(d.K + d.K * h->hals_pullers + 1) * HALS_FLAG_STRIDE);
    }
    const bool gram = h->hals_gram && (size_t)d.K32 * (64 + 2 * (d.L - 1)) * sizeof(float) <= 96 * 1024; // (gram_denom_h's LDS window)
    if (gram) {
        if (!h->gram_numden_h) CMFTRY(dalloc_zero(&h->gram_numden_h, 2 * TK));
        if (contract) {
            CMFTRY(launch_transconv(h, 1, h->XT));
            CMFTRY(gram_denom_h(h, h->gram_numden_h + TK));
        }
        hipLaunchKernelGGL(hals_p_init_kernel, dim3((d.Tl + 63) / 64, d.KB), dim3(256), 0, h->stream, h->hals_PT, h->hslabs, h->gram_numden_h + TK,
                           h->tc_S1, d.Tl, d.K32, h->hals_TPp, h->H, h->Ht, snap, d.TP, d.PADL, flags, nflags);
    } else {
        if (contract) {
            CMFTRY(launch_conv<5>(h, h->estT, d.Tl, h->conv_gy, h->XT));
            CMFTRY(launch_transconv(h, 1, h->estT));
        }
        hipLaunchKernelGGL(hals_p_init_kernel, dim3((d.Tl + 63) / 64, d.KB), dim3(256), 0, h->stream, h->hals_PT, h->hslabs, (const float *)nullptr,
                           h->tc_S1, d.Tl, d.K32, h->hals_TPp, h->H, h->Ht, snap, d.TP, d.PADL, flags, nflags);
    }
    KCHK("hals_p_init_kernel");
    return CMF_OK;
}

HalsRowParams hals_row_params(cmf_handle_s *h, double l1H, double l2H)
{
    const CmfDims &d = h->d;
    HalsRowParams q;
    q.PT = h->hals_PT; q.H = h->H; q.Ht = h->Ht; q.D = h->hals_D; q.GW = h->hals_GW; q.GE = h->hals_GE;
    q.k = 0; q.t_begin = 0; q.t_end = d.Tl;
    q.Tl = d.Tl; q.L = d.L; q.K32 = d.K32; q.TP = d.TP; q.TPp = h->hals_TPp; q.PADL = d.PADL; q.ne = h->hals_ne; q.t_edge0 = h->hals_t_edge0;
    q.l1 = (float)l1H; q.l2 = (float)l2H;
    return q;
}

// hals.jl:124-125 (k outer, t inner) as a software pipeline over the rows, one launch per stage (hals_h_stage_kernel);
// the order of every update is the reference's.  No co-residency requirement.
static int hals_h_sweep_stage(cmf_handle_s *h, const HalsRowParams &q)
{
    const CmfDims &d = h->d;
    HalsStageParams sp;
    sp.row = q;
    sp.Dall = h->hals_D;
    sp.K = d.K; sp.seg = h->hals_seg; sp.nseg = h->hals_nseg;
    sp.CB = (h->hals_seg + 2 * (d.L - 1) + 255) / 256;
    sp.lag = h->hals_opt_lag == 3 ? 3 : 2; // (3 = the unshifted round-1 schedule: the tests compare the two)
    sp.skew = sp.lag == 2 ? 128 : 0; // see hals_h_stage_kernel
    // the last row's last segment, +1 for the pushes of the last sweeps (no-ops for the last row)
    const int nstages = (d.Tl + sp.skew * (d.K - 1) + h->hals_seg - 1) / h->hals_seg + sp.lag * (d.K - 1) + 1;
    dim3 grid(d.K + d.K * sp.CB, std::max(1, d.K - 1));
    for (int stage = 0; stage < nstages; ++stage) {
        sp.stage = stage;
        hipLaunchKernelGGL(hals_h_stage_kernel, grid, dim3(256), 0, h->stream, sp);
        KCHK("hals_h_stage_kernel");
    }
    return CMF_OK;
}

// Any L (the on-chip sweeps stop at L = 64): row after row, each swept by one wave with its pending window in LDS
// (hals_h_row_gen_kernel), its changes then added to the later rows' projections (hals_h_push_gen_kernel) -- the literal
// k outer / t inner order of hals.jl:124-125, 2K launches.
static int hals_h_sweep_general(cmf_handle_s *h, HalsRowParams q)
{
    const CmfDims &d = h->d;
    const int M = (int)rup(d.L, 64) + 64;
    const size_t lds = (size_t)(M + d.L) * sizeof(float);
    if (lds > 64 * 1024) return fail(CMF_ERR_UNSUPPORTED, "HALS H sweep: L = %d exceeds the LDS window of the general row sweep", d.L);
    for (int k = 0; k < d.K; ++k) {
        q.k = k;
        q.D = h->hals_D + (size_t)k * q.TPp;
        hipLaunchKernelGGL(hals_h_row_gen_kernel, dim3(1), dim3(64), lds, h->stream, q);
        KCHK("hals_h_row_gen_kernel");
        if (k + 1 < d.K) {
            hipLaunchKernelGGL(hals_h_push_gen_kernel, dim3((d.Tl + 255) / 256, d.K - 1 - k), dim3(256), 0, h->stream, h->hals_PT, q.D, h->hals_GW,
                               h->hals_GE, k, d.Tl, d.L, d.K32, q.TPp, q.ne, q.t_edge0);
            KCHK("hals_h_push_gen_kernel");
        }
    }
    return CMF_OK;
}

// The whole H sweep as ONE persistent launch on h->stream (hals_h_persist_kernel): flags cleared, K sweepers + (K-1) * P pullers
int hals_persist_launch(cmf_handle_s *h, const HalsRowParams &q, int debug, bool clear_flags)
{
    const CmfDims &d = h->d;
    HalsPersistParams pp;
    pp.row = q;
    pp.Dall = h->hals_D;
    pp.flags = h->hals_flags;
    pp.host_status = h->hals_status;
    pp.K = d.K; pp.P = h->hals_pullers; pp.nblk = (d.Tl + 63) / 64;
    pp.debug = debug;
    pp.stamps = nullptr;
    const size_t nflags = (size_t)(d.K + d.K * pp.P + 1) * HALS_FLAG_STRIDE;
    const size_t lds = std::max((size_t)(d.K - 1) * (2 * d.L - 1 + 64 + 2 * (d.L - 1)) + 1024, (size_t)h->hals_ne * (d.L + 1)) * sizeof(float);
    ProfScope prof_(h, PROF_HALS_PIPE);
    if (clear_flags) HIPCHK(hipMemsetAsync(h->hals_flags, 0, nflags * sizeof(int), h->stream));
    hipLaunchKernelGGL(hals_h_persist_kernel, dim3(d.K + (d.K - 1) * pp.P), dim3(1024), lds, h->stream, pp);
    KCHK("hals_h_persist_kernel");
    return CMF_OK;
}

// How many tile rows of the residual conv chase the row pipeline (0: none).  Needs the persistent pipeline, the one-wave conv
// tiles (K a multiple of 32), the residual the conv stores (hals_gram != 1), and CUs left over beside the pipeline's.
static int hals_chase_rows(cmf_handle_s *h)
{
    const CmfDims &d = h->d;
    if (h->hals_opt_chase == 0 || h->hals_pullers <= 0 || d.K % 32 != 0 || h->small_k || h->hals_gram == 1 || h->hals_debug == 1 || h->hals_debug == 2) return 0;
    if (h->n_cu % 8 != 0 || h->n_cu > 256) return 0;
    const int per_xcd = h->n_cu / 8, need = (d.K + (d.K - 1) * h->hals_pullers + 7) / 8;
    if (per_xcd - need < 4) return 0;
    const int rows_t = (d.Tl + 63) / 64;
    if (rows_t < 64) return 0; // (short recordings: the pipeline's span is a few tile rows of conv)
    int pct = h->hals_opt_chase;
    if (pct < 0) {
        // auto: the share the chasing launch can finish while the pipeline runs.  The conv at 0.87 of the fp32 MFMA roof on the whole chip,
        // the pipeline at 0.63 of its dependency floor (43 cycles a step at 2.4 GHz), its last row starting a quarter of the span in; the
        // chasing stream has (per_xcd - need) / per_xcd of the CUs; 10 % on top, because the rest of the conv starts beside the chasing
        // launch's tail.  Config 5: 64 % (swept: 60-70 % is the flat optimum, profiles/r06_hals_chase.txt).
        const double conv_ms = 2.0 * d.K * (double)d.N * ((double)d.L * d.Tl) / (0.87 * 157.3e12) * 1e3;
        const double pipe_ms = ((double)d.Tl + (double)(d.K - 1) * (d.L - 1)) * 43.0 / 2.4e9 / 0.63 * 1e3;
        const double share = (double)(per_xcd - need) / per_xcd * (0.76 * pipe_ms) / conv_ms;
        pct = (int)std::max(10.0, std::min(95.0, 110.0 * share));
    }
    return std::min(rows_t, (int)((long long)rows_t * std::min(pct, 100) / 100));
}

// The persistent sweep with the first `ra` tile rows of the residual conv (hals.jl:41's residual, the loss fused) chasing it.
static int hals_persist_chased(cmf_handle_s *h, const HalsRowParams &q, int ra)
{
    const CmfDims &d = h->d;
    const int per_xcd = h->n_cu / 8, a_per = (d.K + (d.K - 1) * h->hals_pullers + 7) / 8;
    if (h->hals_sA && h->hals_mask_aper != a_per) { // (another puller count since: other masks)
        for (hipStream_t *st : {&h->hals_sA, &h->hals_sB}) {
            HIPCHK(hipStreamSynchronize(*st));
            HIPCHK(hipStreamDestroy(*st));
            *st = nullptr;
        }
    }
    if (!h->hals_sA) {
        // CU mask bit j = CU j / 8 of XCD j % 8 (profiles/r05_cu_mask_experiment.txt: cutting INSIDE every XCD partitions the chip cleanly)
        uint32_t mask_a[8] = {0}, mask_b[8] = {0};
        for (int j = 0; j < h->n_cu; ++j) (((j / 8) < a_per) ? mask_a : mask_b)[j / 32] |= 1u << (j % 32);
        HIPCHK(hipExtStreamCreateWithCUMask(&h->hals_sA, 8, mask_a));
        HIPCHK(hipExtStreamCreateWithCUMask(&h->hals_sB, 8, mask_b));
        for (hipEvent_t &e : h->hals_ev)
            if (!e) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->hals_cuB = 8 * (per_xcd - a_per);
        h->hals_mask_aper = a_per;
    }
    hipStream_t keep = h->stream;
    // (the flags were cleared by the projection launch, in front of the fork: both launches see them cleared)
    HIPCHK(hipEventRecord(h->hals_ev[0], keep));
    HIPCHK(hipStreamWaitEvent(h->hals_sA, h->hals_ev[0], 0));
    HIPCHK(hipStreamWaitEvent(h->hals_sB, h->hals_ev[0], 0));
    int *prog_last = h->hals_flags + (size_t)(d.K - 1) * HALS_FLAG_STRIDE;
    int *abort_word = h->hals_flags + (size_t)(d.K + d.K * h->hals_pullers) * HALS_FLAG_STRIDE;
    h->stream = h->hals_sA;
    int rc = hals_persist_launch(h, q, h->hals_debug, false);
    h->stream = h->hals_sB;
    int nA = 0;
    if (rc == CMF_OK) rc = launch_conv_rows<4>(h, h->est, 0, ra, 0, h->hals_cuB, prog_last, abort_word, h->hals_status, &nA);
    h->stream = keep;
    CMFTRY(rc);
    HIPCHK(hipEventRecord(h->hals_ev[1], h->hals_sA));
    HIPCHK(hipEventRecord(h->hals_ev[2], h->hals_sB));
    HIPCHK(hipStreamWaitEvent(keep, h->hals_ev[1], 0)); // H is final behind the pipeline; the chasing launch is joined in front of the
                                                         // loss reduction (hals_resid_and_loss): the rest of the conv starts beside it
    h->hals_chased_rows = ra;
    h->hals_chased_partials = nA;
    return CMF_OK;
}

// The residual and the loss behind an H sweep: all of the conv, or what the chasing launch has left (same tiles, same per-tile sums;
// the partials of the two launches lie one behind the other and are added in that order: reproducible)
int hals_resid_and_loss(cmf_handle_s *h, double *sumsq)
{
    const CmfDims &d = h->d;
    const int ra = h->hals_chased_rows, nA = h->hals_chased_partials, rows_t = (d.Tl + 63) / 64;
    h->hals_chased_rows = h->hals_chased_partials = 0;
    if (ra <= 0) return resid_and_loss(h, sumsq);
    int nB = 0;
    if (ra < rows_t) CMFTRY(launch_conv_rows<4>(h, h->est, ra, rows_t - ra, nA, h->n_cu, nullptr, nullptr, nullptr, &nB));
    HIPCHK(hipStreamWaitEvent(h->stream, h->hals_ev[2], 0));
    h->conv_partials = nA + nB;
    set_est(h, 2);
    return reduce_partials(h, h->partial, h->conv_partials, 0, sumsq);
}

static int hals_h_enqueue(cmf_handle_s *h, double l1H, double l2H)
{
    const CmfDims &d = h->d;
    CMFTRY(hals_ensure(h));
    // the lag-Gram taps of W (GW, and GE for the truncated windows at the right edge)
    CMFTRY(gram_tables(h));
    const bool persist = !h->hals_h_general && h->hals_pullers > 0;
    CMFTRY(hals_h_project(h, true, persist)); // (+ the snapshot the persistent pipeline starts from, and its flags cleared)
    const HalsRowParams q = hals_row_params(h, l1H, l2H);
    h->hals_l1 = l1H; h->hals_l2 = l2H;
    set_est(h, 0);
    if (h->hals_h_general) {
        ProfScope prof_(h, PROF_HALS_PIPE);
        return hals_h_sweep_general(h, q);
    }
    if (h->hals_pullers > 0) { // the whole sweep as one persistent launch (hals_h_persist_kernel)
        // Its workgroups wait for each other and every wait is bounded; if one runs out (the grid did not become resident:
        // another process or stream holds CUs) the sweep is redone from this snapshot on the stage pipeline
        // (cmf_hals_update_feature_maps looks at the status word once the stream has drained).
        h->hals_chased_rows = h->hals_chased_partials = 0;
        if (const int ra = hals_chase_rows(h)) return hals_persist_chased(h, q, ra);
        CMFTRY(hals_persist_launch(h, q, h->hals_debug, false)); // (flags cleared by the projection launch)
        return CMF_OK;
    }
    ProfScope prof_(h, PROF_HALS_PIPE); // the whole row pipeline (nstages launches) as one timed span
    return hals_h_sweep_stage(h, q);
}

int hals_h_impl(cmf_handle_s *h, double l1H, double l2H)
{
    CMFTRY(hals_h_enqueue(h, l1H, l2H));
    return wb_after_H(h);
}

// The persistent pipeline reported an expired wait (status word; the stream has drained): H and P hold a half-finished
// sweep.  Restore H from the snapshot, rebuild P from the contractions that are still in place, and run the sweep on the
// stage pipeline, which needs no co-residency.  The handle keeps to the stage pipeline from here on.
int hals_h_rerun(cmf_handle_s *h)
{
    const CmfDims &d = h->d;
    const size_t nH = (size_t)d.TP * d.K32;
    *h->hals_status = 0;
    h->hals_chased_rows = h->hals_chased_partials = 0;
    h->hals_pullers = 0;
    h->hals_opt_persist = 0; // (a later re-plan keeps to it)
    h->hals_reruns += 1;
    HIPCHK(hipMemcpyAsync(h->H, h->hals_snap, nH * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->Ht, h->hals_snap + nH, nH * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    CMFTRY(hals_h_project(h, false));
    set_est(h, 0);
    CMFTRY(hals_h_sweep_stage(h, hals_row_params(h, h->hals_l1, h->hals_l2)));
    return wb_after_H(h); // (an armed write-back has taken the half-finished H: take it again)
}

// ---- optional Gram form of the MU iteration ----------------------------------
int gram_tables(cmf_handle_s *h) // PW -> GW, GE (the lag-Gram taps of W; shared with HALS)
{
    ProfScope prof_(h, PROF_GRAM_TABLES);
    const CmfDims &d = h->d;
    hipLaunchKernelGGL(hals_pw_kernel, dim3(d.L * (d.L + 1) / 2, d.KB * d.KB), dim3(64 * PW_NW), 0, h->stream, h->Wn, h->hals_PW, d.N, d.L, d.Np, d.K32, d.KB);
    KCHK("hals_pw_kernel");
    hipLaunchKernelGGL(hals_gw_kernel, dim3(1024), dim3(256), 0, h->stream, h->hals_PW, h->hals_GW, h->hals_GE, d.L, d.K32, h->hals_ne, d.Tl, h->hals_t_edge0,
                       h->hals_GWt, 2 * d.L); // (+ the taps as [k'][e][k], E = 2L - 1 padded to an even count, for gram_h_mfma_kernel)
    KCHK("hals_gw_kernel");
    return CMF_OK;
}

int gram_denom_h(cmf_handle_s *h, float *out)
{
    ProfScope prof_(h, PROF_GRAM_DENOM_H);
    const CmfDims &d = h->d;
    const size_t lds = (size_t)d.K32 * (64 + 2 * (d.L - 1)) * sizeof(float);
    if (lds > 96 * 1024) return fail(CMF_ERR_UNSUPPORTED, "Gram form: K*L too large for the LDS window");
    // columns with the full lag window, in tiles of 128 / fw: the MFMA kernel on the transposed taps; the rest (the right edge
    // with its per-column taps, and what does not fill a tile): the scalar kernel.  fw waves share the MFMA chain of an
    // output block: short handles (a T/8 shard) need the extra waves to fill the chip.
    const int fw = d.Tl < 16384 ? 4 : (d.Tl < 100000 ? 2 : 1);
    const int tile = 128 / fw;
    const size_t lds_m = ((size_t)d.K32 * (tile + 2 * (d.L - 1)) + (fw > 1 ? 4096 : 0)) * sizeof(float);
    const int ntile = lds_m <= 120 * 1024 ? h->hals_t_edge0 / tile : 0;
    const int t_first = tile * ntile;
    if (ntile > 0) {
        // the leftover columns (the right edge with its per-column taps, and what does not fill a tile), a wave per output:
        // when they are few (the usual case) their workgroups ride at the end of the MFMA kernel's grid
        const bool ride = d.Tl - t_first <= 512;
        GramEdge edge;
        edge.GW = h->hals_GW; edge.GE = h->hals_GE;
        edge.Tl = d.Tl; edge.ne = h->hals_ne; edge.t_edge0 = h->hals_t_edge0; edge.t_first = t_first;
        edge.n_main = ntile * d.KB; edge.kq = d.K32 / 4;
        const int n_edge = ride ? (d.Tl - t_first) * edge.kq : 0;
        const int Ep = 2 * d.L; // E = 2L - 1 taps padded to an even count (hals_gw_kernel wrote them as [k'][e][k])
        hipLaunchKernelGGL(gram_h_mfma_kernel, dim3(edge.n_main + n_edge), dim3(256), lds_m, h->stream, h->Ht, h->hals_GWt, out, d.K, d.L, d.K32, d.TP, d.PADL,
                           Ep, fw, ntile, edge);
        KCHK("gram_h_mfma_kernel");
        if (ride) return CMF_OK;
    }
    const int block0 = t_first / 64, nblock = (d.Tl + 63) / 64 - block0; // (a half block in front of it is simply formed twice)
    if (nblock > 0) {
        hipLaunchKernelGGL(gram_h_kernel, dim3(nblock, d.K32 / 4), dim3(256), lds, h->stream, h->Ht, h->hals_GW, h->hals_GE, out,
                           d.Tl, d.K, d.L, d.K32, d.TP, d.PADL, h->hals_ne, h->hals_t_edge0, block0);
        KCHK("gram_h_kernel");
    }
    return CMF_OK;
}

// W phase of the Gram form in two steps, so that a T-sharded group can put its all-reduce between them:
//   gram_w_partial  numW = H_shift * data' (mult.jl:32; ONE C2 contraction on this handle's columns) -> h->numden[0, LKN),
//                   and HH = H_unfold * H_unfold' (on a shard: its additive share, see hals_hh_kernel) -> hh_out
//   gram_w_finish   denomW = H_shift * est' (mult.jl:33) = HH * W -> h->wslabs (free once the slabs are summed), W update
int gram_w_partial(cmf_handle_s *h, float *hh_out)
{
    CMFTRY(gram_ensure(h));
    CMFTRY(hxt_contract(h, h->X, h->X, 1, h->numden, true)); // (+ a loss reduction deferred by cmf_iterate)
    return compute_hh(h, hh_out);
}

int gram_w_finish(cmf_handle_s *h, const float *HH, double l1W, double l2W, const float *tail_src, float *tail_dst, int tail_n)
{
    const CmfDims &d = h->d;
    {
        ProfScope prof_(h, PROF_GRAM_W);
        CMFTRY(launch_gram_w(h, HH, h->wslabs));
    }
    return w_apply_impl(h, l1W, l2W, tail_src, tail_dst, tail_n, h->wslabs); // mult.jl:37-38
}

int gram_w_impl(cmf_handle_s *h, double l1W, double l2W)
{
    CMFTRY(gram_w_partial(h, nullptr));
    return gram_w_finish(h, h->hals_HH, l1W, l2W);
}

// H phase of the Gram form without the loss (mult.jl:44-52)
int gram_h_update(cmf_handle_s *h, double l1H, double l2H)
{
    const CmfDims &d = h->d;
    CMFTRY(gram_ensure(h));
    const size_t TK = (size_t)d.Tl * d.K32;
    if (!h->gram_numden_h) CMFTRY(dalloc_zero(&h->gram_numden_h, 2 * TK));
    // numH = tensor_transconv(W, data) (mult.jl:47): one C3 contraction; its fragment slabs are summed by the H update
    // (gram = 2 needs the sum itself for <H, numH>)
    CMFTRY(launch_transconv(h, 1, h->XT));
    if (h->gram == 2) CMFTRY(launch_slab_sum(h, h->gram_numden_h, h->hslabs, h->tc_S1, TK));
    // denomH = tensor_transconv(W, tensor_conv(W, H)) (mult.jl:44,48) through the lag-Gram taps of W
    CMFTRY(gram_tables(h));
    CMFTRY(gram_denom_h(h, h->gram_numden_h + TK));
    hipLaunchKernelGGL(h_update_kernel, dim3((d.Tl + HUPD_T - 1) / HUPD_T, d.KB), dim3(256), 0, h->stream, h->H, h->Ht,
                       h->hslabs, TK, h->tc_S1, h->gram_numden_h + TK, (size_t)0, 1,
                       d.Tl, d.K, d.K32, d.PADL, d.TP, (float)l1H, (float)(2.0 * l2H)); // mult.jl:51-52
    KCHK("h_update_kernel");
    set_est(h, 0);
    return wb_after_H(h);
}

int gram_h_impl(cmf_handle_s *h, double l1H, double l2H, double *loss)
{
    const CmfDims &d = h->d;
    CMFTRY(gram_h_update(h, l1H, l2H));
    const size_t TK = (size_t)d.Tl * d.K32;
    double ss = 0.0;
    if (h->gram == 2) {
        // ||est - data||^2 = <H, denomH(H)> - 2 <H, numH> + ||data||^2 with the NEW H (adjointness of conv/transconv)
        CMFTRY(gram_denom_h(h, h->gram_numden_h + TK));
        const int nb = (int)((TK + 1023) / 1024);
        if ((size_t)2 * nb > n_partial(h)) return fail(CMF_ERR_UNSUPPORTED, "Gram loss: partial buffer too small");
        hipLaunchKernelGGL(gram_dot_kernel, dim3(nb), dim3(256), 0, h->stream, h->H, h->gram_numden_h, h->gram_numden_h + TK, h->partial,
                           d.Tl, d.K, d.K32, d.PADL, nb);
        KCHK("gram_dot_kernel");
        hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, h->stream, h->partial, nb, h->d_scalar + 2, (double *)nullptr);
        KCHK("loss_reduce_kernel");
        hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, h->stream, h->partial + nb, nb, h->d_scalar + 3, (double *)nullptr);
        KCHK("loss_reduce_kernel");
        double a = 0.0, b = 0.0;
        CMFTRY(read_scalar(h, 2, &a));
        CMFTRY(read_scalar(h, 3, &b));
        ss = a - 2.0 * b + h->data_sumsq;
        if (ss < 0.0) ss = 0.0;
    } else {
        CMFTRY(loss_partial_impl(h, &ss)); // exact: conv with the fused loss (mult.jl:55-57)
    }
    *loss = std::sqrt(ss) / h->data_norm;
    return CMF_OK;
}

// ---- PGD (src/algs/pgd.jl) ---------------------------------------------------------------------
static int pgd_check(cmf_handle_s *h)
{
    if (h->sharded && h->T_global != h->d.Tl) return fail(CMF_ERR_STATE, "sharded handle: attach a communicator first (cmf_comm_init_rccl / cmf_comm_init_callbacks)");
    if (h->pgd_cur_loss < 0.0) h->pgd_cur_loss = h->data_norm; // pgd.jl:151 (the norm, not its square)
    return CMF_OK;
}

// pgd.jl:245-253: est (here the residual) with the new factors, loss = norm(data - est)^2, step adaptation
static int pgd_finish(cmf_handle_s *h, double *step)
{
    double loss = 0.0;
    CMFTRY(resid_and_loss(h, &loss, h->M != nullptr, h->pgd_loss_abs != 0));
    *step *= (loss < h->pgd_cur_loss) ? 1.05 : 0.70;
    h->pgd_cur_loss = loss;
    return CMF_OK;
}

// UnitNormConstraint (pgd.jl:100-110) on the freshly stepped factor: per-component norms, then the scaling
int pgd_unit_norm(cmf_handle_s *h, bool is_W)
{
    const CmfDims &d = h->d;
    if (!h->pgd_knorm) CMFTRY(dalloc_zero(&h->pgd_knorm, (size_t)d.K32));
    if (is_W) {
        hipLaunchKernelGGL(pgd_w_knorm_kernel, dim3(d.K), dim3(256), 0, h->stream, h->Wt, h->pgd_knorm, d.N, d.L, d.Np, d.K32);
        KCHK("pgd_w_knorm_kernel");
        hipLaunchKernelGGL(pgd_w_kscale_kernel, dim3(1024), dim3(256), 0, h->stream, h->Wt, h->Wn, h->pgd_knorm, d.N, d.K, d.L, d.Np, d.K32);
        KCHK("pgd_w_kscale_kernel");
    } else {
        hipLaunchKernelGGL(pgd_h_knorm_kernel, dim3(d.K), dim3(256), 0, h->stream, h->Ht, h->pgd_knorm, d.Tl, d.TP, d.PADL);
        KCHK("pgd_h_knorm_kernel");
        hipLaunchKernelGGL(pgd_h_kscale_kernel, dim3(1024), dim3(256), 0, h->stream, h->H, h->Ht, h->pgd_knorm, d.Tl, d.K, d.K32, d.TP, d.PADL);
        KCHK("pgd_h_kscale_kernel");
    }
    return CMF_OK;
}

int pgd_w_impl(cmf_handle_s *h, double pen_sq, double pen_abs, int nonneg)
{
    const CmfDims &d = h->d;
    CMFTRY(pgd_check(h));
    if (nonneg < 0 || nonneg > 2) return fail(CMF_ERR_ARG, "constraint must be 0 (none), 1 (NonnegConstraint) or 2 (UnitNormConstraint)");
    const float gscale = h->pgd_loss_abs ? 1.f : 2.f; // pgd.jl:31-33 vs :42-44
    const size_t LKN = (size_t)d.L * d.K32 * d.Np;
    CMFTRY(ensure_resid(h, h->M != nullptr, h->pgd_loss_abs != 0));                                                            // pgd.jl:230 (:64-67 with a mask)
    CMFTRY(hxt_contract(h, h->est, h->est, 1, h->numden)); // pgd.jl:206-214
    dim3 grid(d.Np / 64, d.KB, d.L);
    const int nblk = (d.Np / 64) * d.KB * d.L;
    if ((size_t)nblk > n_partial(h)) return fail(CMF_ERR_UNSUPPORTED, "PGD: partial buffer too small");
    hipLaunchKernelGGL(pgd_w_grad_kernel, grid, dim3(256), 0, h->stream, h->Wt, h->numden, h->numden + LKN, h->partial,
                       d.N, d.K, d.Np, d.K32, (float)pen_sq, (float)pen_abs, gscale);                    // pgd.jl:231-234
    KCHK("pgd_w_grad_kernel");
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, h->stream, h->partial, nblk, h->d_scalar + 1, (double *)nullptr);
    KCHK("loss_reduce_kernel");
    hipLaunchKernelGGL(pgd_w_apply_kernel, grid, dim3(256), 0, h->stream, h->Wt, h->Wn, h->numden + LKN, h->d_scalar + 1,
                       d.N, d.K, d.Np, d.K32, (float)h->pgd_stepW, nonneg == 1);                         // pgd.jl:237-241
    KCHK("pgd_w_apply_kernel");
    if (nonneg == 2) CMFTRY(pgd_unit_norm(h, true));                                                     // pgd.jl:100-110
    set_est(h, 0);
    return pgd_finish(h, &h->pgd_stepW);
}

int pgd_h_impl(cmf_handle_s *h, double pen_sq, double pen_abs, int nonneg, double *loss)
{
    const CmfDims &d = h->d;
    CMFTRY(pgd_check(h));
    if (nonneg < 0 || nonneg > 2) return fail(CMF_ERR_ARG, "constraint must be 0 (none), 1 (NonnegConstraint) or 2 (UnitNormConstraint)");
    const float gscale = h->pgd_loss_abs ? 1.f : 2.f;
    if (!h->pgd_gradH) CMFTRY(dalloc_zero(&h->pgd_gradH, (size_t)d.Tl * d.K32));
    if (h->est_kind == 2 + (h->M ? 1 : 0) + (h->pgd_loss_abs ? 2 : 0)) {
        // est already holds this residual for the resident W, H (stored by the conv that closed the W phase, pgd.jl:245): the
        // H phase's est of pgd.jl:230 is the same array, only tensor_transconv wants it transposed
        hipLaunchKernelGGL(transpose_rows_kernel, dim3(d.Np / 64, (d.Tl + 63) / 64), dim3(256), 0, h->stream, h->est, h->estT, d.Tl, d.Np, d.TP, d.PADL);
        KCHK("transpose_rows_kernel");
    } else {
        h->pgd_loss_abs_now = h->pgd_loss_abs;
        int rc_conv = h->MT ? launch_conv<7>(h, h->estT, d.Tl, h->conv_gy, h->XT) // (mask .* resid)^T (pgd.jl:64-67)
                            : launch_conv<5>(h, h->estT, d.Tl, h->conv_gy, h->XT); // resid^T (pgd.jl:230), or its sign (pgd.jl:42-44)
        h->pgd_loss_abs_now = 0;
        CMFTRY(rc_conv);
    }
    CMFTRY(launch_transconv(h, 1, h->estT));                     // pgd.jl:218-221
    dim3 grid((d.Tl + 63) / 64, d.KB);
    const int nblk = ((d.Tl + 63) / 64) * d.KB;
    if ((size_t)nblk > n_partial(h)) return fail(CMF_ERR_UNSUPPORTED, "PGD: partial buffer too small");
    hipLaunchKernelGGL(pgd_h_grad_kernel, grid, dim3(256), 0, h->stream, h->H, h->hslabs, h->tc_S1, h->pgd_gradH, h->partial,
                       d.Tl, d.K, d.K32, d.PADL, (float)pen_sq, (float)pen_abs, gscale);
    KCHK("pgd_h_grad_kernel");
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, h->stream, h->partial, nblk, h->d_scalar + 1, (double *)nullptr);
    KCHK("loss_reduce_kernel");
    hipLaunchKernelGGL(pgd_h_apply_kernel, grid, dim3(256), 0, h->stream, h->H, h->Ht, h->pgd_gradH, h->d_scalar + 1,
                       d.Tl, d.K, d.K32, d.PADL, d.TP, (float)h->pgd_stepH, nonneg == 1);
    KCHK("pgd_h_apply_kernel");
    if (nonneg == 2) CMFTRY(pgd_unit_norm(h, false)); // pgd.jl:100-110
    set_est(h, 0);
    CMFTRY(wb_after_H(h));
    CMFTRY(pgd_finish(h, &h->pgd_stepH));
    *loss = std::sqrt(h->pgd_cur_loss / (h->data_norm * h->data_norm)); // pgd.jl:201
    return CMF_OK;
}

