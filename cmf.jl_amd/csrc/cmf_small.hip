// cmf_small.hip -- launchers of the few-component contraction kernels (K <= 16: csrc/cmf_small_k.h), a translation unit of their own so
// that their many instantiations compile beside the general kernels' (cmf_api.hip).
#include "cmf_internal.h"

// The C2 contraction for few components: the rows j = l*K + k on the MFMA axis (hxt_small_kernel), compact slabs, their sum expanded to
// [L][K32][Np] (hxt_contract, cmf_api.hip, sends K <= 16 here)
int hxt_contract_small(cmf_handle_s *h, const float *X0, const float *X1, int nsrc, float *out, bool take_carry, bool slabs_only)
{
    const CmfDims &d = h->d;
    ProfScope prof_(h, (nsrc == 2 && X0 == h->X) ? PROF_HXT : (nsrc == 1 && X0 == h->X) ? PROF_HXT_NUM : (nsrc == 1 && X0 == h->est && h->est_kind == 1) ? PROF_HXT_DEN
                      : (nsrc == 1 && X0 == h->est) ? PROF_HXT_RESID : (X0 == h->hals_HX && h->hals_HX) ? PROF_HXT_HH : PROF_OTHER);
    SkHxtParams p;
    p.Ht = h->Ht; p.X0 = X0; p.X1 = X1; p.slabs = h->sk_slabs;
    p.Np = d.Np; p.TP = d.TP; p.PADL = d.PADL; p.K = d.K; p.L = d.L; p.J = h->sk_J; p.JP = h->sk_JP; p.MG = h->sk_MG; p.Tl = d.Tl;
    p.chunk_len = h->sk_chunk_len; p.nsrc = nsrc; p.RV = h->sk_RV;
    size_t lds = std::max<size_t>((size_t)8 * (d.K + 1) * SK_HS_STRIDE, 4 * 16 * 64) * sizeof(float); // (two strips per wave | the chunk reduction)
    const dim3 grid((d.Np / 32) * h->sk_MG, h->sk_ngroups, nsrc);
    switch (h->sk_MBW) {
#define CASE(M_) case M_: if (h->sk_RV) hipLaunchKernelGGL((hxt_small_kernel<(M_ <= 3 ? M_ : 3), SK_RVT>), grid, dim3(256), lds, h->stream, p); /* (1-3 MFMA blocks + VALU rows) */ \
                      else hipLaunchKernelGGL((hxt_small_kernel<M_>), grid, dim3(256), lds, h->stream, p); break;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10)
#undef CASE
    default: return fail(CMF_ERR_STATE, "internal: bad m block count %d", h->sk_MBW);
    }
    KCHK("hxt_small_kernel");
    if (slabs_only) return CMF_OK; // (the caller's next launch sums the slabs itself: w_update_small_kernel)
    CmfLossCarry carry{};
    if (take_carry && h->carry.partial) { // a loss reduction deferred by cmf_iterate rides on this launch
        carry = h->carry;
        h->carry = CmfLossCarry{};
    }
    const size_t n4 = (size_t)nsrc * d.L * d.K32 * d.Np / 4;
    hipLaunchKernelGGL(slab_sum_small_kernel, dim3((unsigned)std::min<size_t>(2048, (n4 + 255) / 256)), dim3(256), 0, h->stream, out, h->sk_slabs,
                       h->sk_ngroups, nsrc, d.L, d.K, d.K32, d.Np, h->sk_JP, carry);
    KCHK("slab_sum_small_kernel");
    return CMF_OK;
}

// C3 for few components: Wj pack, then ONE launch that forms G = Wf x XT (a plain GEMM over n) tile by tile and folds the lag sum
// out[t][k] = sum_l G[(k, l)][t + l] on chip, into hslabs [2][nsrc][Tl][K32].  update_h (sk_can_fuse_h): mult.jl:51-52 in the same launch.
// By default only where the launch is several rounds of workgroups long (no split of the reduction over n): the update of a block
// then hides behind other workgroups' MFMA loops (N = 250, T = 50000, K = 5: the iteration 0.2006 -> 0.1935 ms), while on a
// one-round launch every block completes at the end and the chain drain -> ticket -> slab loads is longer than a launch boundary
// (BASELINE configs[0], 8 pieces: 36.5 -> 45.8 us; option small_k_fuse = 2 fuses there too -- tests).
bool sk_can_fuse_h(const cmf_handle_s *h)
{
    return h->sk_tc && h->sk_fuse && (h->sk3_NS == 1 || h->sk_fuse == 2) && h->sk_cnt && h->sk3_MG == 1 &&
           (double)4 * h->sk3_NS * h->d.Tl * h->d.K32 * 4.0 < 2147483648.0;
}

int launch_transconv_small(cmf_handle_s *h, int nsrc, const float *xt0, bool update_h, float l1, float two_l2)
{
    ProfScope prof_(h, nsrc == 2 ? PROF_TRANSCONV : PROF_TRANSCONV_1);
    const CmfDims &d = h->d;
    if (h->sk_wj_gen != h->est_gen) { // (w_update_small_kernel writes the packed operand itself; whatever else touched W, H or est since: pack again)
        hipLaunchKernelGGL(wj_pack_kernel, dim3((unsigned)std::min<size_t>(1024, ((size_t)d.Np * h->sk3_JP + 255) / 256)), dim3(256), 0, h->stream,
                           h->Wn, h->sk_Wj, d.Np, d.K, d.L, d.K32, h->sk3_Kg, h->sk3_GR, h->sk3_JP);
        KCHK("wj_pack_kernel");
        h->sk_wj_gen = h->est_gen;
    }
    SkGemmParams p;
    p.Wj = h->sk_Wj; p.XT0 = xt0 ? xt0 : h->XT; p.XT1 = h->estT; p.out = h->hslabs;
    p.TP = d.TP; p.PADL = d.PADL; p.JP = h->sk3_JP; p.MG = h->sk3_MG; p.TG = h->sk_TG; p.N2 = (int)rup(d.N, 2); p.Np = d.Np; p.nsrc = nsrc;
    p.Tl = d.Tl; p.K = d.K; p.L = d.L; p.K32 = d.K32; p.Kg = h->sk3_Kg; p.RV = h->sk3_RV;
    p.NS = h->sk3_NS; p.RPS = h->sk3_RPS;
    p.TS = std::max(SK_TILE_STRIDE, (int)rup(128 + 2 * (d.L - 1), 4)); // (a staged row: L - 1 zeros, 128 columns, L - 1 zeros)
    if (p.TS % 32 == 0) p.TS += 4;
    const size_t lds = (size_t)32 * p.TS * sizeof(float);
    p.H = nullptr; p.Ht = nullptr; p.cnt = nullptr; p.target = 0; p.l1 = l1; p.two_l2 = two_l2;
    if (update_h) {
        if (nsrc != 2 || !sk_can_fuse_h(h)) return fail(CMF_ERR_STATE, "internal: this handle cannot update H inside the C3 launch");
        p.H = h->H; p.Ht = h->Ht; p.cnt = h->sk_cnt; p.target = 2 * nsrc * h->sk3_NS;
    }
    const dim3 grid(h->sk_TG / 128 + 1, nsrc * h->sk3_MG * h->sk3_NS);
    switch (h->sk3_MBW) {
#define CASE(M_) case M_: if (h->sk3_RV) hipLaunchKernelGGL((g_gemm_fold_small_kernel<(M_ <= 3 ? M_ : 3), SK_RVT>), grid, dim3(256), lds, h->stream, p); /* (1-3 MFMA blocks + VALU rows) */ \
                 else hipLaunchKernelGGL((g_gemm_fold_small_kernel<M_>), grid, dim3(256), lds, h->stream, p); break;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6)
#undef CASE
    default: return fail(CMF_ERR_STATE, "internal: bad m block count %d", h->sk3_MBW);
    }
    KCHK("g_gemm_fold_small_kernel");
    return CMF_OK;
}

