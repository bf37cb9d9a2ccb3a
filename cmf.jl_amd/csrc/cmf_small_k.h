// cmf_small_k.h -- the MU contractions for FEW COMPONENTS (K <= 16): the shapes the reference itself publishes on
// (README.md: K = 5; figures/fast_bcd/synthetic_comparison.jl:58-64: N = 250, K = 5, L = 20, T <= 50000).
//
// The kernels of cmf_kernels.h put the component index k on a 32-wide MFMA axis (C2, C3) or walk a whole 32-row k block
// per lag (C1), so K = 5 issues 32/5 = 6.4 times the useful MFMAs.  Here the MFMA axes carry the FLATTENED index
// j = l*K + k of the J = L*K (lag, component) pairs -- the rows of the reference's H_unfold / W_unfold
// (src/common.jl:133-142) -- in MG groups of MBW 32-row blocks (MBW <= 6 blocks per wave, chosen so that the padding
// JP - J = 32 * MBW * MG - J is smallest): K = 5, L = 20 fills 100 of 128 rows, K = 16, L = 20 all of 320.
//
//   C1  est[t][n]    = sum_j Hu[j][t] Wf[j][n]          conv_small_kernel (cmf_kernels.h): conv3's tiles, ceil(K/2) k pairs per lag
//   C2  out[j][n]    = sum_t Hu[j][t] X[t][n]           hxt_small_kernel: A = Hu from a lag-shifted LDS strip, B = X rows
//   C3  out[t][k]    = sum_l G[l*K+k][t+l],  G[j][t'] = sum_n Wf[j][n] XT[n][t']
//                                                       g_gemm_small_kernel (a plain GEMM) + fold_small_kernel (the shifted sum)
// with Hu[j][t] = H[t-l][k], Wf[j][n] = W[l][k][n].  Results land in the buffers the element-wise update kernels already read
// (numden [2][L][K32][Np] through slab_sum_small_kernel, hslabs [1][2][Tl][K32]), so everything around the contractions is
// shared with the general path.  v_mfma_f32_32x32x2_f32 throughout (operand maps: cmf_kernels.h).
#pragma once

#define SK_MAXMBW 6       // 32-row m blocks per wave (template parameter MBW): a wave owns 32 * MBW consecutive rows j
#define SK_SC 128         // time rows per staged H strip of hxt_small_kernel
#define SK_HS_STRIDE 201  // floats between the k rows of the strip (>= SK_SC + 64 + 1, odd: the lag-shifted reads of a wave spread over the banks)
#define SK_MAXL 64        // the strip holds SK_SC + L - 1 <= 191 columns

struct SkHxtParams {
    const float *Ht;  // [K32][TP]
    const float *X0;  // data [TP][Np]
    const float *X1;  // est  [TP][Np]
    float *slabs;     // [gridDim.y][nsrc][JP][Np]
    int Np, TP, PADL, K, L, J, JP, MG, Tl;
    int chunk_len;    // rows per wave, a multiple of 16; 4 waves of a workgroup = 4 consecutive chunks, added through LDS
    int nsrc;
};

// C2.  grid ((Np/32) * MG, chunks/4, nsrc), 256 threads.  A wave = one 32-column n block x 32*MBW rows j x one time chunk.
// Placement (speed only): workgroup b runs on XCD b % 8, so the MG row groups of one n block -- which read the same X rows --
// are put 8 apart in blockIdx.x when the n blocks come in multiples of 8: the same L2 serves them.
template <int MBW>
__global__ __launch_bounds__(256, MBW <= 4 ? 4 : 3) void hxt_small_kernel(SkHxtParams p)
{
    extern __shared__ __attribute__((aligned(16))) float sk_lds[]; // 4 strips of (K+1) rows; reused for the chunk reduction
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
    const int strip = (K + 1) * SK_HS_STRIDE;
    float *Hs = sk_lds + wave * strip;
    const int tc0 = (blockIdx.y * 4 + wave) * p.chunk_len;

    // per-lane read base of each m block: row j -> (l, k): Hs[k][c + (L-1) - l] is H[t0 + c - l][k]; rows j >= J read the zero row K
    int abase[MBW];
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb) {
        const int j = (mg * MBW + mb) * 32 + i;
        const int l = j / K, k = j - l * K;
        abase[mb] = (j < p.J) ? k * SK_HS_STRIDE + (L - 1) - l + h : K * SK_HS_STRIDE + h;
    }
    for (int c = lane; c < SK_HS_STRIDE; c += 64) Hs[K * SK_HS_STRIDE + c] = 0.f;

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
    const int width = SK_SC + L - 1;
    for (int s0 = 0; s0 < p.chunk_len && tc0 + s0 < p.Tl; s0 += SK_SC) { // (X rows >= Tl are zero padding: nothing to add behind them)
        const int rows = (p.chunk_len - s0 < SK_SC) ? p.chunk_len - s0 : SK_SC; // a multiple of 16
        __builtin_amdgcn_wave_barrier();
        // strip: Hs[k][c] = Ht[k][PADL + tc0 + s0 - (L-1) + c], c in [0, width)
        for (int k = 0; k < K; ++k) {
            const float *srcp = p.Ht + (size_t)k * TP + (p.PADL + tc0 + s0 - (L - 1));
            for (int c = lane; c < width; c += 64) Hs[k * SK_HS_STRIDE + c] = srcp[c];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // 8 steps (16 time rows) per round; the X operands of a round are loaded a round ahead into the OTHER of two register
        // sets (a copy from a "next" set into the current one would make every round wait for its own prefetch)
        float b0[8], b1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) b0[u] = cmf_bload(xr, xoff, (s0 + 2 * u) * Np * 4);
        auto round = [&](const float (&bc)[8], float (&bn)[8], int r0) {
            const int nx = (r0 + 16 < rows) ? r0 + 16 : r0;
#pragma unroll
            for (int u = 0; u < 8; ++u) bn[u] = cmf_bload(xr, xoff, (s0 + nx + 2 * u) * Np * 4);
            __builtin_amdgcn_sched_barrier(0); // keep the prefetch at the top of the round (the scheduler otherwise sinks it below the MFMAs)
            // software-pipelined like conv2_lag: the LDS reads of step u+1 are issued ahead of the MFMAs of step u
            float a[MBW];
#pragma unroll
            for (int mb = 0; mb < MBW; ++mb) a[mb] = Hs[abase[mb] + r0];
            __builtin_amdgcn_sched_group_barrier(0x100, MBW, 0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float an[MBW];
#pragma unroll
                for (int mb = 0; mb < MBW; ++mb) an[mb] = 0.f;
                if (u + 1 < 8) {
#pragma unroll
                    for (int mb = 0; mb < MBW; ++mb) an[mb] = Hs[abase[mb] + r0 + 2 * (u + 1)];
                    __builtin_amdgcn_sched_group_barrier(0x100, MBW, 0);
                }
#pragma unroll
                for (int mb = 0; mb < MBW; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb], bc[u], acc[mb], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, MBW, 0);
#pragma unroll
                for (int mb = 0; mb < MBW; ++mb) a[mb] = an[mb];
            }
        };
        for (int r0 = 0; r0 < rows; r0 += 32) {
            round(b0, b1, r0);
            if (r0 + 16 < rows) round(b1, b0, r0 + 16);
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
}

// out[src][l][k][n] = sum over the slabs of in[slab][src][l*K+k][n]; rows k >= K of out are written as zeros.  The last
// block also performs a carried loss reduction, like slab_sum_kernel.  grid.x covers nsrc * L * K32 * Np / 4 float4 words.
__global__ __launch_bounds__(256) void slab_sum_small_kernel(float *out, const float *in, int nslabs, int nsrc, int L, int K, int K32, int Np, int JP,
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

// Wj[n][j] = W[l][n][k] (j = l*K + k; zero for j >= J), from Wn [Lp][Np][K32]: the A operand of the C3 GEMM, j contiguous.
__global__ __launch_bounds__(256) void wj_pack_kernel(const float *Wn, float *Wj, int Np, int K, int K32, int J, int JP)
{
    const size_t total = (size_t)Np * JP;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(idx % JP), n = (int)(idx / JP);
        float v = 0.f;
        if (j < J) {
            const int l = j / K, k = j - l * K;
            v = Wn[((size_t)l * Np + n) * K32 + k];
        }
        Wj[idx] = v;
    }
}

struct SkGemmParams {
    const float *Wj;  // [Np][JP]
    const float *XT0; // dataT [Np][TP]
    const float *XT1; // estT  [Np][TP]
    float *G;         // [nsrc][JP][TG]
    int TP, PADL, JP, MG, TG, N2, Np; // N2 = roundup(N, 2): rows n >= N2 of XT and Wj are zero
    int nsrc;
};

// C3, first half: G[j][t'] = sum_n Wj[n][j] XT[n][t'].  grid (TG/128, nsrc*MG), 256 threads; a wave = 32*MBW rows j x 32 columns t'.
// Both operands are read from global memory directly in MFMA layout (128 contiguous bytes per half-wave; Wj is L2-resident),
// eight n pairs ahead.
template <int MBW>
__global__ __launch_bounds__(256, MBW <= 4 ? 4 : 3) void g_gemm_small_kernel(SkGemmParams p)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int src = blockIdx.y % p.nsrc, mg = blockIdx.y / p.nsrc;
    const int tb = (blockIdx.x * 4 + wave) * 32;
    if (tb >= p.TG) return;
    const float *XT = src ? p.XT1 : p.XT0;
    const __amdgpu_buffer_rsrc_t ar = cmf_rsrc(p.Wj + mg * MBW * 32, ((size_t)(p.Np - 1) * p.JP + MBW * 32) * 4);
    const __amdgpu_buffer_rsrc_t br = cmf_rsrc(XT + p.PADL + tb, ((size_t)(p.Np - 1) * p.TP + 32) * 4);
    const int aoff = (h * p.JP + i) * 4, boff = (h * p.TP + i) * 4;
    f32x16 acc[MBW];
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
    constexpr int U = 4; // n pairs per round; the operands of a round are loaded a round ahead into the other of two register sets
    float a0[U][MBW], b0[U], a1[U][MBW], b1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        b0[u] = cmf_bload(br, boff, (2 * u) * p.TP * 4);
#pragma unroll
        for (int mb = 0; mb < MBW; ++mb) a0[u][mb] = cmf_bload(ar, aoff + mb * 128, (2 * u) * p.JP * 4);
    }
    const int nrounds = (p.N2 + 2 * U - 1) / (2 * U); // (the rows a last round reads past N2 are zero padding of both operands)
    auto round = [&](const float (&ac)[U][MBW], const float (&bc)[U], float (&an)[U][MBW], float (&bn)[U], int rd) {
        const int nx = ((rd + 1 < nrounds) ? rd + 1 : rd) * 2 * U;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bn[u] = cmf_bload(br, boff, (nx + 2 * u) * p.TP * 4);
#pragma unroll
            for (int mb = 0; mb < MBW; ++mb) an[u][mb] = cmf_bload(ar, aoff + mb * 128, (nx + 2 * u) * p.JP * 4);
        }
        __builtin_amdgcn_sched_barrier(0); // the prefetch stays in front of the round's MFMAs
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int mb = 0; mb < MBW; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u][mb], bc[u], acc[mb], 0, 0, 0);
    };
    for (int rd = 0; rd < nrounds; rd += 2) {
        round(a0, b0, a1, b1, rd);
        if (rd + 1 < nrounds) round(a1, b1, a0, b0, rd + 1);
    }
    float *G = p.G + ((size_t)src * p.JP + mg * MBW * 32) * p.TG + tb;
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) G[(size_t)(mb * 32 + cmf_crow(r, h)) * p.TG + i] = acc[mb][r];
}

// C3, second half: out[src][t][k] = sum_l G[src][l*K + k][t + l]  (k < K; zeros for K <= k < K32), t in [0, Tl).
// grid (ceil(Tl/64), nsrc), 256 threads: thread (tt = tid % 64, kq = tid / 64) sums the components k = kq, kq + 4, ... of
// column t (reads coalesced along t), the 64 x 32 tile goes out through LDS in rows of K32 (coalesced along k).
__global__ __launch_bounds__(256) void fold_small_kernel(const float *G, float *out, int Tl, int K, int L, int K32, int JP, int TG)
{
    __shared__ float tile[64][33];
    const int tt = threadIdx.x & 63, kq = threadIdx.x >> 6;
    const int t0 = blockIdx.x * 64, src = blockIdx.y;
    const float *Gs = G + (size_t)src * JP * TG;
    for (int k = kq; k < 32; k += 4) {
        float s = 0.f;
        if (k < K && t0 + tt < Tl)
            for (int l = 0; l < L; ++l) s += Gs[(size_t)(l * K + k) * TG + t0 + tt + l];
        tile[tt][k] = s;
    }
    __syncthreads();
    float *o = out + (size_t)src * Tl * K32;
    for (int e = threadIdx.x; e < 64 * 32; e += 256) {
        const int t = e >> 5, k = e & 31;
        if (t0 + t < Tl) o[(size_t)(t0 + t) * K32 + k] = tile[t][k];
    }
}
