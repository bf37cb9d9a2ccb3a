// cmf_small_k.h -- the MU contractions for FEW COMPONENTS (K <= 16): the shapes the reference itself publishes on
// (README.md: K = 5; figures/fast_bcd/synthetic_comparison.jl:58-64: N = 250, K = 5, L = 20, T <= 50000).
//
// The kernels of cmf_kernels.h put the component index k on a 32-wide MFMA axis (C2, C3) or walk a whole 32-row k block
// per lag (C1), so K = 5 issues 32/5 = 6.4 times the useful MFMAs.  Here the MFMA axes carry the FLATTENED index of the
// (lag, component) pairs -- the rows of the reference's H_unfold / W_unfold (src/common.jl:133-142) -- in blocks of 32 rows:
// K = 5, L = 20 fills 100 of 128 rows (and the last 4 of them run on the VALU instead of in a fourth block), K = 16, L = 20 all of 320.
//
//   C1  est[t][n]    = sum_j Hu[j][t] Wf[j][n]          conv_small_kernel (cmf_kernels.h): conv3's tiles, ceil(K/2) k pairs per lag
//   C2  out[j][n]    = sum_t Hu[j][t] X[t][n]           hxt_small_kernel: j = l*K + k; A = Hu from lag-shifted LDS strips (LDS-DMA), B = X rows
//   C3  out[t][k]    = sum_l G[(k,l)][t+l],  G[(k,l)][t'] = sum_n Wf[(k,l)][n] XT[n][t']
//                                                       g_gemm_fold_small_kernel: the G tile (k-major rows) is folded on chip, one launch
// with Hu[j][t] = H[t-l][k], Wf[j][n] = W[l][k][n].  Results land in the buffers the element-wise update kernels already read
// (the C2 slabs are summed by w_update_small_kernel -- or slab_sum_small_kernel into numden [2][L][K32][Np] --, C3 writes
// hslabs [2 per piece of its reduction][nsrc][Tl][K32]), so everything around the contractions is shared with the general path.
// v_mfma_f32_32x32x2_f32 throughout (operand maps: cmf_kernels.h).
#pragma once

#define SK_MAXMBW 6       // 32-row m blocks per wave (template parameter MBW): a wave owns 32 * MBW consecutive rows j (C3)
#define SK_MAXMBW_C2 10   // ... of the C2 kernel: K = 16, L = 20 (320 rows) in ONE row group, so that X is read once (two groups of
                          // five blocks each re-read it: 0.51 of the roof on useful flops; 160 accumulator registers, two waves per SIMD)
#define SK_SC 64          // time rows per staged H strip of hxt_small_kernel: TWO LDS buffers per wave, the next strip filled by LDS-DMA during the current one
#define SK_HS_STRIDE 129  // floats between the k rows of a strip (>= SK_SC + 63 + 1, odd: the lag-shifted reads of a wave spread over the banks)
#define SK_MAXL 64        // a strip holds SK_SC + L - 1 <= 127 columns

// global -> LDS without registers, one dword per lane: LDS address = lds_base + 4 * lane (cmf_glds16 for the rules of the game:
// the compiler does not see the load in flight; completion is awaited by a counted s_waitcnt vmcnt)
__device__ __forceinline__ void cmf_glds4(const void *gsrc, unsigned lds_base)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_base)
                 : "memory");
}

struct SkHxtParams {
    const float *Ht;  // [K32][TP]
    const float *X0;  // data [TP][Np]
    const float *X1;  // est  [TP][Np]
    float *slabs;     // [gridDim.y][nsrc][JP][Np]
    int Np, TP, PADL, K, L, J, JP, MG, Tl;
    int RV;           // rows j in [32 * MBW, 32 * MBW + RV) are contracted on the VALU (RVT kernels; MG = 1)
    int chunk_len;    // rows per wave, a multiple of 16; 4 waves of a workgroup = 4 consecutive chunks, added through LDS
    int nsrc;
};

// C2.  grid ((Np/32) * MG, chunks/4, nsrc), 256 threads.  A wave = one 32-column n block x 32*MBW rows j x one time chunk.
// Placement (speed only): workgroup b runs on XCD b % 8, so the MG row groups of one n block -- which read the same X rows --
// are put 8 apart in blockIdx.x when the n blocks come in multiples of 8: the same L2 serves them.
// The A operand comes from a wave-private strip of H in LDS: Hs[k][c] = Ht[k][t0 - (L-1) + c] for SK_SC time rows, read at lag-shifted
// offsets.  Two strip buffers per wave: the NEXT strip is brought global -> LDS by LDS-DMA (no registers) behind the first step of the
// current one, three rounds of X prefetches ahead of its first use, so that a strip boundary is a buffer swap and a counted wait that
// holds back nothing.  (Until round 5 the strip was loaded through registers: at the boundary itself -- an exposed HBM round trip behind
// the X prefetch ring, since a wave's loads return in issue order, paid by the two waves of a SIMD at the same moment -- then, for K <= 8,
// early into 3 K registers; the DMA form is as fast or faster at every K and is the only one left: K = 16: 0.60 -> 0.74 of the roof.)
// RVT > 0: the last RV <= RVT rows j of the (single) row group are not padded to a 32-row MFMA block but contracted on the
// VALU beside the MFMAs: K = 5, L = 20 is 100 rows = 3 blocks + 4 rows -- a fourth block would multiply 28 rows of zeros
// (a quarter of the launch's MFMAs).  Lane (i, h) holds X[t + h][n0 + i] for the MFMA's B operand already; a VALU row costs one
// broadcast LDS read (issued a step ahead, like the A operands) and one FMA per step, hidden under the step's MBW MFMAs; the
// two halves of the wave (the two time parities) are added at the end.
#define SK_RVT 4
template <int MBW, int RVT = 0>
__global__ __launch_bounds__(256, MBW > 6 ? 2 : (MBW <= 4 ? 4 : 3)) void hxt_small_kernel(SkHxtParams p)
{
    extern __shared__ __attribute__((aligned(16))) float sk_lds[]; // 4 x 2 strips of (K+1) rows; reused for the chunk reduction
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
    int nb, mg;
    if ((gridDim.x / p.MG) % 8 == 0) {
        mg = (blockIdx.x >> 3) % p.MG;
        nb = (blockIdx.x & 7) + 8 * (blockIdx.x / (8 * p.MG));
    } else {
        nb = blockIdx.x / p.MG;
        mg = blockIdx.x % p.MG;
    }
    const int src = blockIdx.z;
    const int Np = p.Np, TP = p.TP, K = p.K, L = p.L;
    const float *X = src ? p.X1 : p.X0;
    constexpr int SC = SK_SC, HST = SK_HS_STRIDE;
    const int strip = (K + 1) * HST;
    float *Hs = sk_lds + wave * strip * 2; // (two buffers per wave)
    const int tc0 = (blockIdx.y * 4 + wave) * p.chunk_len;

    // per-lane read base of each m block: row j -> (l, k): Hs[k][c + (L-1) - l] is H[t0 + c - l][k]; rows j >= J read the zero row K
    int abase[MBW];
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb) {
        const int j = (mg * MBW + mb) * 32 + i;
        const int l = j / K, k = j - l * K;
        abase[mb] = (j < p.J) ? k * HST + (L - 1) - l + h : K * HST + h;
    }
    for (int c = lane; c < HST; c += 64) Hs[K * HST + c] = Hs[strip + K * HST + c] = 0.f;
    constexpr int RVN = RVT > 0 ? RVT : 2; // (the VALU rows go in pairs: one v_pk_fma_f32 per pair and step)
    static_assert(RVN % 2 == 0, "row pairs");
    int vbase[RVN];
    f32x2 accv[RVN / 2];
#pragma unroll
    for (int r = 0; r < RVN; ++r) {
        const int j = MBW * 32 + r;
        const int l = j / K, k = j - l * K;
        vbase[r] = (RVT > 0 && r < p.RV && j < p.J) ? k * HST + (L - 1) - l + h : K * HST + h;
        accv[r >> 1][r & 1] = 0.f;
    }

    f32x16 acc[MBW];
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;

    // (rows past the allocation -- a chunk that overshoots Tl by more than the padding -- are out of the descriptor's range and read 0)
    int xrows = TP - p.PADL - tc0;
    xrows = xrows < 0 ? 0 : (xrows > p.chunk_len + 2 ? p.chunk_len + 2 : xrows);
    const __amdgpu_buffer_rsrc_t xr = cmf_rsrc(X + (size_t)(p.PADL + (xrows ? tc0 : 0)) * Np, (size_t)xrows * Np * 4);
    const int xoff = (h * Np + nb * 32 + i) * 4;
    // The X operands run through a ring of NS register sets of 8 rows pairs (one round = 16 time rows = 8 MFMA steps): the set
    // of round r is loaded NS - 1 rounds ahead -- across the strip boundaries too -- because these loads come from HBM (every X
    // element is used by exactly one wave) and one round of MFMAs (2048 cycles) does not cover that latency at two waves per
    // SIMD (round 4 loaded one round ahead and started every 128-row strip with an exposed load).
    constexpr int NS = 4;
    f32x2 bx[NS][4]; // (a round's 8 X values in register pairs: the VALU rows' packed FMAs take either half)
    const int nrounds = (p.Tl - tc0 <= 0) ? 0 : (((p.chunk_len < p.Tl - tc0 + 15 ? p.chunk_len : ((p.Tl - tc0 + 15) & ~15))) >> 4); // (X rows >= Tl are zero padding: nothing to add behind them)
    auto xload = [&](f32x2 (&b)[4], int rd) {
        const int rc = (rd < nrounds) ? rd : (nrounds ? nrounds - 1 : 0); // (behind the chunk: a row it owns, never used)
#pragma unroll
        for (int u = 0; u < 8; ++u) b[u >> 1][u & 1] = cmf_bload(xr, xoff, (16 * rc + 2 * u) * Np * 4);
    };
    // DMA: strip s0 -> buffer dst; two dwords per lane and row k (width <= 127: the second one's lanes behind the width land in the
    // row's padding); 2 K untracked loads, awaited by the counted vmcnt below
    const unsigned lds_Hs = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float *)Hs);
    auto hdma = [&](int s0, int buf) {
        const float *srcp = p.Ht + (p.PADL + tc0 + s0 - (L - 1)) + lane;
        const unsigned dst = lds_Hs + (unsigned)(buf * strip) * 4u;
        for (int k = 0; k < K; ++k) {
            cmf_glds4(srcp + (size_t)k * TP, dst + (unsigned)(k * HST) * 4u);
            cmf_glds4(srcp + (size_t)k * TP + 64, dst + (unsigned)(k * HST + 64) * 4u);
        }
    };
    if (nrounds) hdma(0, 0); // (in front of the ring's first NS - 1 sets: the counted wait at the top of a pass holds for the first one too)
#pragma unroll
    for (int q = 0; q < NS - 1; ++q) xload(bx[q], q);
    int dma_s0 = -1, dma_buf = 0; // >= 0: the strip the next round issues behind its first step
    auto mround = [&](const f32x2 (&bc)[4], int r0) { // the MFMAs of one round on strip rows r0 .. r0 + 15
        // software-pipelined like conv2_lag: the LDS reads of step u+1 are issued ahead of the MFMAs of step u
        float a[MBW];
        f32x2 va[RVN / 2];
#pragma unroll
        for (int mb = 0; mb < MBW; ++mb) a[mb] = Hs[abase[mb] + r0];
#pragma unroll
        for (int r = 0; r < RVN; ++r) va[r >> 1][r & 1] = RVT > 0 ? Hs[vbase[r] + r0] : 0.f;
        __builtin_amdgcn_sched_group_barrier(0x100, MBW + RVT, 0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float an[MBW];
            f32x2 van[RVN / 2];
#pragma unroll
            for (int mb = 0; mb < MBW; ++mb) an[mb] = 0.f;
#pragma unroll
            for (int r = 0; r < RVN / 2; ++r) van[r] = f32x2{0.f, 0.f};
            if (u + 1 < 8) {
#pragma unroll
                for (int mb = 0; mb < MBW; ++mb) an[mb] = Hs[abase[mb] + r0 + 2 * (u + 1)];
                if (RVT > 0) {
#pragma unroll
                    for (int r = 0; r < RVN; ++r) van[r >> 1][r & 1] = Hs[vbase[r] + r0 + 2 * (u + 1)];
                }
                __builtin_amdgcn_sched_group_barrier(0x100, MBW + RVT, 0);
            }
#pragma unroll
            for (int mb = 0; mb < MBW; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb], bc[u >> 1][u & 1], acc[mb], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, MBW, 0);
            if (u == 0 && dma_s0 >= 0) { // behind the round's wait for its X set: nothing waits on these loads before the next round's
                __builtin_amdgcn_sched_barrier(0);
                hdma(dma_s0, dma_buf);
                dma_s0 = -1;
                __builtin_amdgcn_sched_barrier(0);
            }
            if (RVT > 0) { // the VALU rows of this step (operands read a step ago): two rows per instruction -- beside fp32 MFMAs a VALU
#pragma unroll     // instruction costs the MFMA pipe its four cycles (profiles/r06_c3_tail_pieces.txt) -- opaque to the compiler
                for (int r = 0; r < RVN / 2; ++r) accv[r] = (u & 1) ? cmf_pk_fma_opaque<true>(va[r], bc[u >> 1], accv[r]) : cmf_pk_fma_opaque<false>(va[r], bc[u >> 1], accv[r]);
            }
#pragma unroll
            for (int mb = 0; mb < MBW; ++mb) a[mb] = an[mb];
#pragma unroll
            for (int r = 0; r < RVN / 2; ++r) va[r] = van[r];
        }
    };
    const int width = SC + L - 1;
    // strip: Hs[k][c] = Ht[k][PADL + tc0 + s0 - (L-1) + c], c in [0, width)
    float *const Hs0 = Hs;
    for (int rd0 = 0; rd0 < nrounds; rd0 += SC / 16) { // a strip = SC / 16 rounds, a multiple of NS
        const int s0 = rd0 * 16;
        __builtin_amdgcn_wave_barrier(); // (a wave's LDS operations execute in issue order: the previous strip's reads are ahead of these writes)
        {
            // The strip of this pass was issued behind the first step of the previous pass (the first one: in front of the ring's
            // first sets), with 3 x 8 X prefetches behind it: a wave's loads return in issue order, so it has landed once at most
            // those 24 are outstanding -- a wait that holds back no X load.  (Issued at the TOP of a pass, in front of the round's own
            // wait for its X set, the compiler's counted wait -- which does not know these loads -- would wait for them too.)
            __builtin_amdgcn_s_waitcnt(0x0F70 | (24 & 15) | ((24 >> 4) << 14)); // vmcnt(24)
            const int buf = (rd0 / (SC / 16)) & 1;
            Hs = Hs0 + buf * strip;
            if (rd0 + SC / 16 < nrounds) { // (the other buffer's last readers -- the MFMAs of the pass before -- have their operands)
                dma_s0 = s0 + SC;
                dma_buf = buf ^ 1;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int q0 = 0; q0 < SC / 16 && rd0 + q0 < nrounds; q0 += NS) {
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                const int rd = rd0 + q0 + q;
                if (rd < nrounds) {
                    xload(bx[(q + NS - 1) % NS], rd + NS - 1);
                    __builtin_amdgcn_sched_barrier(0); // keep the prefetch at the top of the round (the scheduler otherwise sinks it below the MFMAs)
                    mround(bx[q], 16 * (q0 + q));
                }
            }
        }
    }

    // the 4 chunks of the workgroup, added in chunk order (deterministic), one m block per pass; wave w stores rows 4w .. 4w+3 of the 16
    __syncthreads();
    float *red = sk_lds; // [4 waves][16][64]
    float *slab = p.slabs + ((size_t)blockIdx.y * p.nsrc + src) * p.JP * Np;
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[mb][r];
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = wave * 4 + rr;
            float sum = red[r * 64 + lane];
#pragma unroll
            for (int v = 1; v < 4; ++v) sum += red[(v * 16 + r) * 64 + lane];
            const int j = (mg * MBW + mb) * 32 + cmf_crow(r, h);
            slab[(size_t)j * Np + nb * 32 + i] = sum;
        }
        __syncthreads();
    }
    if (RVT > 0) { // the VALU rows: the two time parities of a wave, then the 4 chunks in chunk order
#pragma unroll
        for (int r = 0; r < RVN; ++r) red[(wave * 16 + r) * 64 + lane] = accv[r >> 1][r & 1] + __shfl_xor(accv[r >> 1][r & 1], 32);
        __syncthreads();
        for (int r = wave; r < RVT; r += 4) {
            float sum = red[r * 64 + lane];
#pragma unroll
            for (int v = 1; v < 4; ++v) sum += red[(v * 16 + r) * 64 + lane];
            if (r < p.RV && h == 0) slab[(size_t)(MBW * 32 + r) * Np + nb * 32 + i] = sum;
        }
    }
}

// out[src][l][k][n] = sum over the slabs of in[slab][src][l*K+k][n]; rows k >= K of out are written as zeros.  The last
// block also performs a carried loss reduction, like slab_sum_kernel.  grid.x covers nsrc * L * K32 * Np / 4 float4 words.
static __global__ __launch_bounds__(256) void slab_sum_small_kernel(float *out, const float *in, int nslabs, int nsrc, int L, int K, int K32, int Np, int JP,
                                                              CmfLossCarry carry)
{
    const size_t n4 = (size_t)nsrc * L * K32 * Np / 4;
    const size_t sstride = (size_t)nsrc * JP * Np;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < n4; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t e = 4 * idx; // ((src * L + l) * K32 + k) * Np + n
        const int n = (int)(e % Np);
        const int k = (int)((e / Np) % K32);
        const int l = (int)((e / ((size_t)Np * K32)) % L);
        const int src = (int)(e / ((size_t)Np * K32 * L));
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < K) {
            const float *base = in + ((size_t)src * JP + (size_t)l * K + k) * Np + n;
            a = *reinterpret_cast<const float4 *>(base);
            for (int s = 1; s < nslabs; s += 4) { // four loads in flight, added in slab order
                float4 b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) b[u] = *reinterpret_cast<const float4 *>(base + (size_t)((s + u < nslabs) ? s + u : s) * sstride);
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (s + u < nslabs) { a.x += b[u].x; a.y += b[u].y; a.z += b[u].z; a.w += b[u].w; }
            }
        }
        reinterpret_cast<float4 *>(out)[idx] = a;
    }
    if (carry.partial && blockIdx.x == gridDim.x - 1) cmf_block_loss_reduce(carry);
}

// E1 for few components on a single handle, in ONE launch behind the C2 kernel: the slabs of hxt_small_kernel are summed, W is
// updated (mult.jl:37-38) in both layouts, and the A operand of the C3 GEMM (Wj, layout of wj_pack_kernel) is written from the same
// values -- the slab sum, the update and the pack were three launches of 5-8 us each on a 250 us iteration.  The last block also
// performs a loss reduction deferred by cmf_iterate.  slabs: [nslabs][2][JP][Np].
// grid (Np/64 + 1, L, K), block 256: one (lag, component) row and 64 units per workgroup (the last column: the loss reduction); its four 64-thread groups each sum every fourth
// slab (all loads of a thread independent and in flight together) and are combined in a fixed order through LDS, so the result does
// not depend on timing.  (The first form -- one workgroup per lag, the slabs added one after the other by each thread -- was 80
// workgroups of dependent loads: 14 us, as long as the three launches it replaced.)
static __global__ __launch_bounds__(256) void w_update_small_kernel(float *Wt, float *Wn, float *Wj, const float *slabs, int nslabs,
                                                              int N, int K, int L, int Np, int K32, int JP, int Kg, int GR, int JP3,
                                                              float l1, float two_l2, CmfLossCarry carry)
{
    __shared__ float red[3][64][2];
    if (blockIdx.x == gridDim.x - 1) { // the extra column of the grid: one of its blocks performs the carried loss reduction, beside the others
        if (carry.partial && blockIdx.y == 0 && blockIdx.z == 0) cmf_block_loss_reduce(carry); // (behind a block's own update it was the launch's critical path)
        return;
    }
    const int nn = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + nn, l = blockIdx.y, k = blockIdx.z;
    const size_t sstride = (size_t)2 * JP * Np;
    const float *bn = slabs + (size_t)(l * K + k) * Np + n, *bd = bn + (size_t)JP * Np;
    const size_t it = ((size_t)l * K32 + k) * Np + n;
    const float w_old = (g == 0) ? Wt[it] : 0.f; // (requested with the slabs, not behind their sum)
    float num = 0.f, den = 0.f;
    for (int s0 = g; s0 < nslabs; s0 += 32) { // eight slabs of this group per pass
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int sl = s0 + 4 * u;
            const size_t o = (size_t)(sl < nslabs ? sl : 0) * sstride;
            a[u] = bn[o];
            b[u] = bd[o];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (s0 + 4 * u < nslabs) { num += a[u]; den += b[u]; }
    }
    if (g > 0) {
        red[g - 1][nn][0] = num;
        red[g - 1][nn][1] = den;
    }
    __syncthreads();
    if (g == 0) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            num += red[q][nn][0];
            den += red[q][nn][1];
        }
        const float wn = (n < N) ? cmf_mu(w_old, num, den, l1, two_l2) : 0.f;
        Wt[it] = wn;
        Wn[((size_t)l * Np + n) * K32 + k] = wn;
        if (Wj) {
            const int gq = k / Kg;
            Wj[(size_t)n * JP3 + gq * GR + (k - gq * Kg) * L + l] = wn;
        }
    }
}

// C3 puts whole components into a row group: group g holds the components [g*Kg, (g+1)*Kg), its row kl*L + l is
// (component g*Kg + kl, lag l), padded with zero rows to 32*MBW -- k-major, so that the L rows a folded output needs lie next to
// each other (at most three components per 32-row block at L = 20) and no output is shared between two row groups.
// Wj[n][g*32*MBW + kl*L + l] = W[l][n][g*Kg + kl], from Wn [Lp][Np][K32]: the A operand of the C3 GEMM, rows contiguous per n.
static __global__ __launch_bounds__(256) void wj_pack_kernel(const float *Wn, float *Wj, int Np, int K, int L, int K32, int Kg, int GR, int JP)
{
    const size_t total = (size_t)Np * JP;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int jj = (int)(idx % JP), n = (int)(idx / JP);
        const int g = jj / GR, rr = jj - g * GR;
        const int kl = rr / L, l = rr - kl * L, k = g * Kg + kl;
        Wj[idx] = (kl < Kg && k < K) ? Wn[((size_t)l * Np + n) * K32 + k] : 0.f;
    }
}

#define SK_TILE_STRIDE 132 // least number of floats between the rows of the staged G block (SkGemmParams::TS)
#define SK_FOLD_COLS 192   // 128 + L - 1 <= 191 output columns of a workgroup (L <= SK_MAXL)

struct SkGemmParams {
    const float *Wj;  // [Np][JP]
    const float *XT0; // dataT [Np][TP]
    const float *XT1; // estT  [Np][TP]
    float *out;       // hslabs [2][nsrc][Tl][K32]: slab 0 = the part of out[t][k] from G columns of t's own 128-column block, slab 1 = the part
                      // from the next block (the last L - 1 columns of a block only; zeros elsewhere)
    int TP, PADL, JP, MG, TG, N2, Np, nsrc;
    int Tl, K, L, K32, Kg;
    int RV;           // rows [32 * MBW, 32 * MBW + RV) of the (single) row group are contracted on the VALU (RVT kernels)
    int TS;           // floats between the staged rows of the G tile in LDS: >= 128 + 2 (L - 1) (launch_transconv_small: 32 TS floats of dynamic LDS)
    int NS, RPS;      // short recordings: the reduction over n is cut into NS pieces of RPS rounds (8 rows of n each), piece q writes slabs 2q, 2q + 1
    // H != NULL (two sources, one row group): the element-wise update of H (mult.jl:51-52) runs INSIDE this launch -- whichever
    // workgroup completes a 128-column block's slabs updates that block (sk_h_update_block)
    float *H, *Ht;    // [TP][K32], [K32][TP]
    int *cnt;         // a ticket counter per 128-column block, zero between launches (the last arriver resets its own)
    int target;       // arrivals per block: 2 * nsrc * NS (its own tile's workgroups and the next tile's, whose spill it takes)
    float l1, two_l2;
};

// 16-byte agent-scope (sc1, write-through) accesses of the slabs that change hands inside the launch (cdna_hip_programming.md
// section 6 Guideline 16, R1: the storing waves drain, one lane draws the ticket)
typedef unsigned int sk_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void sk_store4_sc1(__amdgpu_buffer_rsrc_t r, int voff, int soff, f32x4 v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(sk_u32x4, v), r, voff, soff, 16);
}
__device__ __forceinline__ f32x4 sk_load4_sc1(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 16));
}

// mult.jl:51-52 on the 128 columns of block b, by the workgroup whose ticket completed the block: num / den = the sums of the
// 2 NS slabs of either source in h_update_kernel's order (four partial sums over every fourth slab, then ((p0 + p1) + p2) + p3:
// bit for bit the separate launch's H), H row by row, H' through LDS (stage: 16 rows of 129 floats).
__device__ __forceinline__ void sk_h_update_block(const SkGemmParams &p, int b, float *stage, int tid)
{
    const int nk4 = (p.K + 3) >> 2, t0 = b * 128, S = 2 * p.NS;
    const int slab_bytes = p.Tl * p.K32 * 4; // (the host takes this form only while all slabs lie below 2 GiB)
    const __amdgpu_buffer_rsrc_t rs = cmf_rsrc(p.out, (size_t)S * 2 * slab_bytes);
    for (int e = tid; e < 128 * nk4; e += 256) {
        const int tt = e / nk4, k4 = e - tt * nk4, t = t0 + tt;
        f32x4 hn = {0.f, 0.f, 0.f, 0.f};
        if (t < p.Tl) {
            const int voff = (t * p.K32 + 4 * k4) * 4;
            f32x4 n0 = {0.f, 0.f, 0.f, 0.f}, n1 = n0, n2 = n0, n3 = n0, d0 = n0, d1 = n0, d2 = n0, d3 = n0;
            for (int s = 0; s < S; s += 4) { // (S is even: slabs s, s + 1 exist)
                const f32x4 a0 = sk_load4_sc1(rs, voff, (2 * s) * slab_bytes), b0 = sk_load4_sc1(rs, voff, (2 * s + 1) * slab_bytes);
                const f32x4 a1 = sk_load4_sc1(rs, voff, (2 * s + 2) * slab_bytes), b1 = sk_load4_sc1(rs, voff, (2 * s + 3) * slab_bytes);
                n0 += a0; d0 += b0; n1 += a1; d1 += b1;
                if (s + 2 < S) {
                    const f32x4 a2 = sk_load4_sc1(rs, voff, (2 * s + 4) * slab_bytes), b2 = sk_load4_sc1(rs, voff, (2 * s + 5) * slab_bytes);
                    const f32x4 a3 = sk_load4_sc1(rs, voff, (2 * s + 6) * slab_bytes), b3 = sk_load4_sc1(rs, voff, (2 * s + 7) * slab_bytes);
                    n2 += a2; d2 += b2; n3 += a3; d3 += b3;
                }
            }
            const f32x4 num = ((n0 + n1) + n2) + n3, den = ((d0 + d1) + d2) + d3;
            f32x4 *hp = reinterpret_cast<f32x4 *>(p.H + (size_t)(p.PADL + t) * p.K32 + 4 * k4);
            const f32x4 x = *hp;
#pragma unroll
            for (int c = 0; c < 4; ++c) hn[c] = (4 * k4 + c < p.K) ? cmf_mu(x[c], num[c], den[c], p.l1, p.two_l2) : 0.f;
            *hp = hn;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) stage[(4 * k4 + c) * 129 + tt] = hn[c];
    }
    __syncthreads();
    for (int e = tid; e < 4 * nk4 * 128; e += 256) {
        const int kk = e >> 7, tt = e & 127;
        if (t0 + tt < p.Tl) p.Ht[(size_t)kk * p.TP + p.PADL + t0 + tt] = stage[kk * 129 + tt];
    }
    __syncthreads();
}

// The end of a workgroup of the fused form: its slab stores (all sc1) drained by every wave, one lane draws a ticket on its own
// block and on the block in front (whose last L - 1 columns it has just completed); a ticket that is the block's last makes this
// workgroup the one that updates the block.  lds: at least 16 * 129 floats nobody else uses any more.
__device__ __forceinline__ void sk_tickets_and_updates(const SkGemmParams &p, bool own, float *lds, int tid)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave
    __syncthreads();
    int *word = reinterpret_cast<int *>(lds);
    if (tid < 2) { // lane 0: the own block's ticket, lane 1: the ticket of the block in front -- both in flight together
        const int b = (int)blockIdx.x - tid;
        int last = 0;
        if (tid == 0 ? own : b >= 0) {
            last = __hip_atomic_fetch_add(p.cnt + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p.target - 1;
            if (last) __hip_atomic_store(p.cnt + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (zero again for the next launch)
        }
        // The updating side reads every slab with sc1 loads (sk_h_update_block), and nothing has read these lines since the launch began,
        // so neither this CU's L1 nor this XCD's L2 can hold an older copy; the agent-scope acquire of the last arriver stays
        // nevertheless (cdna_hip_programming.md section 6 Guideline 16: the measured hand-off table has one workgroup per CU, this
        // launch three or four) -- it costs about 0.5 % of the iteration here (A / B on one box: 189.4 - 190.3 us with, 188.3 - 189.1 without).
        if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        word[tid] = last;
    }
    __syncthreads();
    const int lo = word[0], lp = word[1];
    __syncthreads(); // (the words are read before the staging below overwrites them)
    if (lp) sk_h_update_block(p, (int)blockIdx.x - 1, lds, tid);
    if (lo) sk_h_update_block(p, (int)blockIdx.x, lds, tid);
}

// One staged block of the G tile folded into outs: thread (output column t = c0 - (L-1) + tid) adds rows row0 .. row0 + nrows of the
// group -- row r is (component, lag) = (r / L, r % L), the same for every thread, so the bookkeeping is scalar -- at columns
// trel + lag.  All NR reads are issued FIRST, branch-free: as a loop of guarded reads the fold was a chain of NR dependent LDS round
// trips per block, 13.7 us of a 64 us launch at K = 5, L = 20 (in-kernel stamps, round 5).  Round 6: a staged row carries L - 1 zero
// columns on either side, so a read needs no range test (clamp + two selects per row before: 8 VALU instructions per row, 2 now --
// beside other workgroups' fp32 MFMA loops every VALU instruction queues behind a 64-cycle MFMA, and a fold block took 9.3 us there
// against 1.4 alone: profiles/r06_c3_tail_pieces.txt).  The sums are taken in the same order as ever (bitwise the same result).
// NB: rows read ahead of their sums (NR, or NR / 2 in the instantiation with the tightest register budget: all NR addresses and values
// live at once cost it 8 spilled registers)
template <int NR, int NB = NR>
__device__ __forceinline__ void sk_fold_rows(const float *tile, int TS, float *outs, int tid, int L, int kn, int row0, int nrows)
{
    static_assert(NR % NB == 0, "whole batches");
    int kk = row0 / L, ll = row0 - kk * L;
    float s = 0.f;
#pragma unroll
    for (int r0 = 0; r0 < NR; r0 += NB) {
        float v[NB];
        int lr = ll;
#pragma unroll
        for (int rl = 0; rl < NB; ++rl) { // (row r0 + rl, G column t + lr: the L - 1 zero columns on either side of a staged row stand for the columns outside the tile)
            v[rl] = tile[(r0 + rl) * TS + tid + lr];
            lr = (lr + 1 == L) ? 0 : lr + 1;
        }
#pragma unroll
        for (int rl = 0; rl < NB; ++rl) {
            if (r0 + rl < nrows) {
                s += v[rl];
                if (++ll == L) {
                    if (kk < kn) outs[kk * SK_FOLD_COLS + tid] += s;
                    s = 0.f;
                    ll = 0;
                    ++kk;
                }
            }
        }
    }
    if (ll != 0 && kk < kn) outs[kk * SK_FOLD_COLS + tid] += s; // (a component that continues in the next block)
}

// C3 in ONE launch: out[t][k] = sum_l G[(k, l)][t + l] with G[(k, l)][t'] = sum_n W[l][n][k] XT[n][t'] (common.jl:71-81 with the
// lag sum taken after the contraction over n).  grid (TG/128 + 1, nsrc*MG), 256 threads.  The four waves of a workgroup form
// the G tile of 32*MBW rows x 128 columns t' in [c0, c0 + 128) exactly like a plain GEMM -- both operands read from global
// memory directly in MFMA layout (128 contiguous bytes per half-wave; Wj is L2-resident), eight n pairs ahead -- and then FOLD it
// on chip: the tile goes through LDS one 32-row block at a time, thread c adds, for output column t = c0 - (L-1) + c, the rows
// of each component at the columns t + l that lie inside the tile (lag order inside a block, blocks in order: deterministic).
// Outputs t >= c0 go to slab 0; the L - 1 columns in front of the tile -- whose other lags lie in the previous workgroup's
// tile -- go to slab 1, which this workgroup also zero-fills for the rest of the previous block; h_update / the slab sums add
// the two slabs.  G never exists in memory (until round 4: 2*JP*TG floats written by the GEMM and read back by a fold kernel,
// as much HBM traffic as a pass over data), and no MFMA work is repeated.  The extra workgroup at the end of grid.x only
// zero-fills slab 1 of the last block.
// RVT > 0 (one row group): the last RV <= RVT rows of the group are not padded to a 32-row MFMA block (K = 5, L = 20: 100 rows, a
// fourth block with 4 live rows) but contracted on the VALU, next to the MFMAs of the other blocks: their Wj columns lie in LDS
// (in the tile's space, which the fold uses only afterwards), one broadcast b128 read per n pair gives a lane the RVT values
// of its n parity, read one step ahead; the two parities meet in the epilogue and the rows go through the fold as a short block.
template <int MBW, int RVT = 0>
__global__ __launch_bounds__(256, MBW <= 4 ? 4 : 3) void g_gemm_fold_small_kernel(SkGemmParams p)
{
    static_assert(RVT == 0 || RVT == 4, "the VALU rows are read as one b128");
    extern __shared__ __attribute__((aligned(16))) float tile[]; // 32 staged rows of TS floats: L - 1 zeros | 128 columns of G | L - 1 zeros (| padding)
    __shared__ float outs[16 * SK_FOLD_COLS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int TS = p.TS;
    const int src = blockIdx.y % p.nsrc, mg = (blockIdx.y / p.nsrc) % p.MG, q = blockIdx.y / (p.nsrc * p.MG);
    const int c0 = blockIdx.x * 128;
    const int L = p.L, K32 = p.K32, Tl = p.Tl;
    const int kbase = mg * p.Kg;
    const int kn = (p.K - kbase < p.Kg) ? p.K - kbase : p.Kg;       // components of this row group
    // columns k it writes: its components (+, for the last group, the zeros up to the next multiple of 4: h_update reads whole groups of
    // four components; the columns behind that are zero in every slab since the handle was made -- every writer of hslabs writes
    // zeros there -- and writing them again was 19 MB per launch at K = 5)
    const int kw = ((mg == p.MG - 1) ? ((p.K + 3) & ~3) : kbase + kn) - kbase;
    float *slab0 = p.out + ((size_t)2 * q * p.nsrc + src) * Tl * K32 + kbase;
    float *slab1 = slab0 + (size_t)p.nsrc * Tl * K32;
    if (c0 >= p.TG) { // nothing spills into the last block
        if (p.H) { // (fused form: kw is a multiple of 4)
            const __amdgpu_buffer_rsrc_t r1 = cmf_rsrc(slab1, (size_t)Tl * K32 * 4);
            const int nk4 = kw >> 2;
            for (int e = tid; e < 128 * nk4; e += 256) {
                const int tt = e / nk4, k4 = e - tt * nk4, t = c0 - 128 + tt;
                if (t >= 0 && t < Tl) sk_store4_sc1(r1, (t * K32 + 4 * k4) * 4, 0, f32x4{0.f, 0.f, 0.f, 0.f});
            }
            sk_tickets_and_updates(p, false, tile, tid);
            return;
        }
        for (int e = tid; e < 128 * kw; e += 256) {
            const int tt = e / kw, kk = e - tt * kw, t = c0 - 128 + tt;
            if (t >= 0 && t < Tl) slab1[(size_t)t * K32 + kk] = 0.f;
        }
        return;
    }
    for (int e = tid; e < 16 * SK_FOLD_COLS; e += 256) outs[e] = 0.f;

    const int tb = c0 + wave * 32;
    const float *XT = src ? p.XT1 : p.XT0;
    const int n0 = q * p.RPS * 8; // first row of n of this piece (0: the whole reduction)
    const __amdgpu_buffer_rsrc_t ar = cmf_rsrc(p.Wj + (size_t)n0 * p.JP + mg * MBW * 32, ((size_t)(p.Np - 1 - n0) * p.JP + MBW * 32) * 4);
    const __amdgpu_buffer_rsrc_t br = cmf_rsrc(XT + (size_t)n0 * p.TP + p.PADL + tb, ((size_t)(p.Np - 1 - n0) * p.TP + 32) * 4);
    const int aoff = (h * p.JP + i) * 4, boff = (h * p.TP + i) * 4;
    f32x16 acc[MBW];
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
    // U n pairs per round.  The A operand (Wj, L2-resident) is loaded one round ahead into the other of two register sets; the B
    // operand (XT: every element is used by one wave, so it comes from HBM, in 128-byte pieces of rows TP floats apart) runs
    // through a ring of three sets and is loaded TWO rounds ahead: one round of MFMAs (1024 cycles) does not cover that latency.
    constexpr int U = 4, NSB = 3;
    float a[2][U][MBW], b[NSB][U];
    const int nr_all = (p.N2 + 2 * U - 1) / (2 * U) - q * p.RPS; // (the rows a last round reads past N2 are zero padding of both operands)
    const int nrounds = nr_all < p.RPS ? nr_all : p.RPS;     // >= 1: the host cuts no empty piece
    float accv[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 wv = {0.f, 0.f, 0.f, 0.f};
    const f32x4 *wl = reinterpret_cast<const f32x4 *>(tile) + h; // wl[n] = Wj[n][32 MBW .. + 4) (zero rows behind N: the host checks that 8 nrounds + 2 rows fit)
    if (RVT > 0) {
        for (int e = tid; e < (8 * nrounds + 2) * 4; e += 256) {
            const int n = e >> 2, r = e & 3;
            tile[e] = (n0 + n < p.Np && r < p.RV) ? p.Wj[(size_t)(n0 + n) * p.JP + MBW * 32 + r] : 0.f;
        }
        __syncthreads();
        wv = wl[0];
    }
    auto loadA = [&](float (&x)[U][MBW], int rd) {
        const int nx = ((rd < nrounds) ? rd : nrounds - 1) * 2 * U;
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int mb = 0; mb < MBW; ++mb) x[u][mb] = cmf_bload(ar, aoff + mb * 128, (nx + 2 * u) * p.JP * 4);
    };
    auto loadB = [&](float (&x)[U], int rd) {
        const int nx = ((rd < nrounds) ? rd : nrounds - 1) * 2 * U;
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = cmf_bload(br, boff, (nx + 2 * u) * p.TP * 4);
    };
    loadB(b[0], 0);
    loadB(b[1], 1);
    loadA(a[0], 0);
    for (int rd0 = 0; rd0 < nrounds; rd0 += 6) { // 6 = lcm(2, NSB): the set indices below are compile-time constants
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int rd = rd0 + q;
            if (rd < nrounds) {
                loadB(b[(q + 2) % NSB], rd + 2); // (A issued first, so that it does not queue behind this HBM load, measured slower: 84 -> 86 us)
                loadA(a[(q + 1) % 2], rd + 1);
                __builtin_amdgcn_sched_barrier(0); // the prefetch stays in front of the round's MFMAs
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    f32x4 wvn = wv;
                    if (RVT > 0) {
                        wvn = wl[rd * 2 * U + 2 * (u + 1)]; // the next step's VALU operands
                        __builtin_amdgcn_sched_barrier(0); // (sched_group_barrier does not classify the asm FMAs: without a full barrier they are hoisted to right behind their operands' reads)
                    }
#pragma unroll
                    for (int mb = 0; mb < MBW; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q % 2][u][mb], b[q % NSB][u], acc[mb], 0, 0, 0);
                    if (RVT > 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) accv[r] = cmf_fma_opaque(wv[r], b[q % NSB][u], accv[r]); // (opaque: see hxt_small_kernel; the packed form is slower here: 60.4 -> 61.2 us)
                        __builtin_amdgcn_sched_barrier(0);
                        wv = wvn;
                    }
                }
            }
        }
    }

    // the fold: thread c < 128 + L - 1 owns output column t = c0 - (L-1) + c
    const bool folder = tid < 128 + L - 1;
#pragma unroll
    for (int mb = 0; mb < MBW + (RVT > 0 ? 1 : 0); ++mb) {
        __syncthreads(); // (the zero fill of outs / the previous block's reads / the main loops' reads of the VALU rows' operands)
        if (mb == 0) { // the zero columns on either side of the staged rows (the VALU rows' operands lay here during the main loop)
            const int m2 = 2 * (L - 1);
            for (int e = tid; e < 32 * m2; e += 256) {
                const int r = e / m2, j = e - r * m2;
                tile[r * TS + (j < L - 1 ? j : 128 + j)] = 0.f;
            }
        }
        if (mb < MBW) {
#pragma unroll
            for (int r = 0; r < 16; ++r) tile[cmf_crow(r, h) * TS + (L - 1) + wave * 32 + i] = acc[mb < MBW ? mb : 0][r];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float g = accv[r] + __shfl_xor(accv[r], 32); // the two n parities
                if (h == 0) tile[r * TS + (L - 1) + wave * 32 + i] = g;
            }
        }
        __syncthreads();
        if (folder) {
            if (mb < MBW) sk_fold_rows<32, (MBW == 4 ? 16 : 32)>(tile, TS, outs, tid, L, kn, 32 * mb, 32);
            else sk_fold_rows<4>(tile, TS, outs, tid, L, kn, 32 * mb, p.RV);
        }
    }
    __syncthreads();
    if (p.H) { // the slabs change hands inside the launch: 16-byte sc1 stores, then the tickets (kw is a multiple of 4, one row group)
        const __amdgpu_buffer_rsrc_t r0 = cmf_rsrc(slab0, (size_t)Tl * K32 * 4), r1 = cmf_rsrc(slab1, (size_t)Tl * K32 * 4);
        const int nk4 = kw >> 2, spill0 = 128 - (L - 1);
        for (int e = tid; e < 128 * nk4; e += 256) {
            const int tt = e / nk4, k4 = e - tt * nk4;
            const int ts = tt >= spill0 ? tt - spill0 : 0;
            f32x4 v0, v1;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int kk = 4 * k4 + c, kc = kk < kn ? kk : 0;
                const float o0 = outs[kc * SK_FOLD_COLS + tt + (L - 1)], o1 = outs[kc * SK_FOLD_COLS + ts];
                v0[c] = kk < kn ? o0 : 0.f;
                v1[c] = (kk < kn && tt >= spill0) ? o1 : 0.f;
            }
            const int t = c0 + tt, tp = t - 128;
            if (t < Tl) sk_store4_sc1(r0, (t * K32 + 4 * k4) * 4, 0, v0);
            if (tp >= 0 && tp < Tl) sk_store4_sc1(r1, (tp * K32 + 4 * k4) * 4, 0, v1);
        }
        sk_tickets_and_updates(p, true, tile, tid);
        return;
    }
    for (int e = tid; e < 128 * kw; e += 256) {
        const int tt = e / kw, kk = e - tt * kw;
        const int t = c0 + tt; // own block: slab 0
        if (t < Tl) slab0[(size_t)t * K32 + kk] = kk < kn ? outs[kk * SK_FOLD_COLS + tt + (L - 1)] : 0.f;
        const int tp = c0 - 128 + tt; // previous block: slab 1 (its last L - 1 columns carry this tile's share)
        if (tp >= 0 && tp < Tl) slab1[(size_t)tp * K32 + kk] = (kk < kn && tt >= 128 - (L - 1)) ? outs[kk * SK_FOLD_COLS + tt - (128 - (L - 1))] : 0.f;
    }
}
