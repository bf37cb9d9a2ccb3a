// cmf_kernels.h -- hand-written gfx950 (CDNA4, wave64) kernels of the MU hot path.
//
// Device layouts (all fp32; r = PADL + t is the padded time row):
//   H   [TP][K32]        H[t][k]      (Julia's K x T memory order, padded)      primary
//   Ht  [K32][TP]        H^T                                                     copy for conv
//   Wt  [Lp][K32][Np]    W[l][k][n]   (n fastest)                                primary
//   Wn  [Lp][Np][K32]    W[l][n][k]   (Julia's K x N x L memory order, padded)   copy for transconv
//   X   [TP][Np]         data[t][n]   (Julia's N x T memory order, padded)
//   XT  [Np][TP]         data^T
//   est [TP][Np] / estT [Np][TP]      same shapes as X / XT
// All padding (k >= K, n >= N, rows outside [PADL, PADL+Tl) except halos) is zero and is
// kept zero by every kernel, so the MFMA loops need no bounds checks.
//
// The three contractions of the reference are each one kernel built on
// v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles per SIMD):
//   conv_kernel       C1  est[t][n]   = sum_{l,k} H[t-l][k] W[l][k][n]   (common.jl:24-34)
//   hxt_kernel        C2  out[l][k][n]= sum_t H[t-l][k] X[t][n]          (mult.jl:31-34)
//   transconv_kernel  C3  out[t][k]   = sum_{l,n} X[t+l][n] W[l][n][k]   (common.jl:71-81)
// MFMA operand maps (32x32x2 f32): lane = 32*h + i.  A: A[row i][kk h], B: B[kk h][col i],
// C/D: col = i, row = (reg&3) + 8*(reg>>2) + 4*h.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CMF_EPS_F 2.220446049250313e-16f

struct CmfDims {
    int N, Tl, K, L;
    int Np;   // roundup(N, 128)
    int KB;   // ceil(K / 32)
    int K32;  // 32 * KB
    int PADL; // roundup(L-1, 32) + 32   left time padding (halo + slack)
    int TP;   // PADL + roundup(Tl, 512) + 256   padded time rows
    int Lp;   // lags allocated in Wt/Wn: roundup(L,4) if L <= 32 else roundup(L,32)
};

__device__ __forceinline__ int cmf_crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Buffer (SRSRC) loads: a wave-uniform base + one per-lane VGPR offset + a scalar offset per load,
// instead of a 64-bit per-lane address per load (saves the address VGPRs of long unrolled streams).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t cmf_rsrc(const void *base, size_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0,
                                             (int)(bytes > 0xFFFFFFFFull ? 0xFFFFFFFFull : bytes), 0x00020000);
}
__device__ __forceinline__ float cmf_bload(__amdgpu_buffer_rsrc_t r, int voff_bytes, int soff_bytes)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff_bytes, soff_bytes, 0));
}

// ---------------------------------------------------------------------------------------------
// C1: tensor_conv.  One workgroup = 128 (t) x 128 (n) output tile, 4 waves as 2x2, each wave a
// 64x64 sub-tile = 2x2 MFMA blocks.  The K-row H strip of the tile (with its left lag halo) is
// staged once per 32-lag block in LDS; the W slab streams through a double-buffered LDS chunk,
// one lag at a time (reduction order: k-block, lag, k).
//   MODE 0: store est[t][n]      (A operand = H, B = W)
//   MODE 1: store estT[n][t]     (A operand = W, B = H; same registers, swapped MFMA roles)
//   MODE 2: no store; per-workgroup sum of (est - data)^2 -> partial[]  (mult.jl:55-57 fused)
//   MODE 3: MODE 0 + MODE 2
//   MODE 4: store est - data (the residual of hals.jl / pgd.jl) in the [t][n] layout + the loss sum
//   MODE 5: store (est - data)^T in the [n][t] layout (p.data = dataT)
//   MODE 6: MODE 4 with the residual multiplied by p.mask [t][n]  (MaskedLoss, pgd.jl:58-70)
//   MODE 7: MODE 5 with the residual multiplied by p.mask = maskT [n][t]
//   (modes 4-7 with p.loss_abs: the stored quantity is the AbsoluteLoss gradient sign(est - data) [.* mask] and the
//    loss sum is sum |mask .* (est - data)|, pgd.jl:41-47)
// ---------------------------------------------------------------------------------------------
struct ConvParams {
    const float *Ht;
    const float *Wt;
    float *out;
    const float *data; // X [TP][Np] (modes 2, 3, 4, 6) or XT [Np][TP] (modes 5, 7)
    const float *mask; // same layout as data (modes 6, 7)
    double *partial;   // [gridDim.x * gridDim.y]
    int Np, TP, PADL, K, KB, L;
    int T_store; // rows t < T_store are stored / counted
    int N;       // columns n >= N are padding: a 32-column MFMA block that lies wholly behind N is not computed (its sums are 0)
    int loss_abs; // residual modes (4-7) only: 1 = AbsoluteLoss (pgd.jl:41-47): store sign(est - data) [.* mask], sum |.|
};

// agent-scope accesses (global_load / global_store ... sc1): the hand-off forms of MI355X_MICROARCH.md "inter-workgroup visibility"
__device__ __forceinline__ int cmf_load_sc1(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float cmf_load_sc1(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void cmf_store_sc1(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void cmf_store_sc1(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void cmf_drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// a float4 of a buffer another workgroup of a CONCURRENT kernel has published: four agent-scope loads (bypass this CU's L1)
__device__ __forceinline__ f32x4 cmf_load4_sc1(const float *p)
{
    f32x4 v;
    v[0] = cmf_load_sc1(p); v[1] = cmf_load_sc1(p + 1); v[2] = cmf_load_sc1(p + 2); v[3] = cmf_load_sc1(p + 3);
    return v;
}

#define CONV_TRANSPOSED(MODE) ((MODE) == 1 || (MODE) == 5 || (MODE) == 7)
#define CONV_HS_STRIDE 160
#define CONV_HS_FLOATS (32 * CONV_HS_STRIDE)
#define CONV_WS_FLOATS (32 * 128)

// WAVES = waves per workgroup: 4 (a 128 x 128 tile as 2 x 2 waves) or 1 (the workgroup IS one 64 x 64 wave tile)
__device__ __forceinline__ void cmf_bstore(float v, __amdgpu_buffer_rsrc_t r, int voff_bytes, int soff_bytes)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, voff_bytes, soff_bytes, 0);
}

// Every access of the epilogue is a buffer load / store: descriptor on the wave's 64 x 64 sub-tile (a wave-uniform
// base), one per-lane byte offset, and a scalar offset per element -- no vector address arithmetic (VALU slots are
// what a workgroup at the edge of its life is short of, see conv2_kernel), and nothing wide for the compiler to hoist
// out of a tile loop.  In the [t][n] layout the descriptor ends at row T_store, so the rows of a partial last tile
// that do not exist are dropped by the bounds check instead of by per-element branches.
template <int MODE, int WAVES = 4, bool PRE = false>
__device__ __forceinline__ void conv_epilogue_(f32x16 (&acc)[2][2], const ConvParams &p, int t0, int n0, int wt, int wn,
                                               int i, int h, int lane, int wave, int tid, int pidx, const float (&pre)[2][2][16])
{
    if (pidx < 0) pidx = blockIdx.y * gridDim.x + blockIdx.x; // slot of this tile's loss partial
    const int Np = p.Np, TP = p.TP;
    constexpr bool LOSS = (MODE == 2 || MODE == 3 || MODE == 4 || MODE == 6);
    constexpr bool RESID = (MODE == 4 || MODE == 6);
    constexpr bool MASKED = (MODE == 6 || MODE == 7);
    // wave-uniform origin of this wave's 64 x 64 sub-tile
    const int tw = __builtin_amdgcn_readfirstlane(t0 + wt * 64);
    const int nw = __builtin_amdgcn_readfirstlane(n0 + wn * 64);
    if (!CONV_TRANSPOSED(MODE)) {
        // acc[ti][ni][r]: t = tw + ti*32 + crow(r,h), n = nw + ni*32 + i
        int rows = p.T_store - tw; // rows of the sub-tile that exist
        rows = rows < 0 ? 0 : (rows > 64 ? 64 : rows);
        const size_t origin = (size_t)(p.PADL + tw) * Np + nw;
        const size_t bytes = rows ? ((size_t)(rows - 1) * Np + 64) * 4 : 0;
        const __amdgpu_buffer_rsrc_t ro = cmf_rsrc(p.out + origin, bytes);
        const __amdgpu_buffer_rsrc_t rd = cmf_rsrc(p.data + origin, bytes);
        const __amdgpu_buffer_rsrc_t rm = cmf_rsrc(p.mask + origin, bytes);
        const int voff = (4 * h * Np + i) * 4;
        float lsum = 0.f;
        // the body twice, selected by one wave-uniform branch: only the last row of tiles pays for the row test
        auto body = [&](auto partial_rows, auto abs_loss) {
            constexpr bool PARTIAL = decltype(partial_rows)::value;
            constexpr bool ABS = decltype(abs_loss)::value; // AbsoluteLoss (residual modes only): one wave-uniform branch, not a select per element
            // all operand loads of a group first (the W registers are dead by now), then arithmetic and stores: a load
            // queued behind stores would wait for them on the in-order vmcnt.  A group is the whole 64 x 64 sub-tile,
            // or one 32 x 32 block when the mask doubles the operands (the registers do not stretch further).
            constexpr int GT = MASKED ? 1 : 2; // blocks per group along t and n
#pragma unroll
            for (int gt = 0; gt < 2; gt += GT)
#pragma unroll
                for (int gn = 0; gn < 2; gn += GT) {
                    float dv[GT][GT][16], mv[GT][GT][16];
#pragma unroll
                    for (int ti = 0; ti < GT; ++ti)
#pragma unroll
                        for (int ni = 0; ni < GT; ++ni)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int so = (((gt + ti) * 32 + (r & 3) + 8 * (r >> 2)) * Np + (gn + ni) * 32) * 4;
                                dv[ti][ni][r] = !LOSS ? 0.f : (PRE && !MASKED) ? pre[gt + ti][gn + ni][r] // (the caller loaded the data tile under its MFMA loop: conv3_tile)
                                                                               : cmf_bload(rd, voff, so); // rows past T_store read as 0 (masked below)
                                mv[ti][ni][r] = MASKED ? cmf_bload(rm, voff, so) : 1.f;
                            }
#pragma unroll
                    for (int ti = 0; ti < GT; ++ti)
#pragma unroll
                        for (int ni = 0; ni < GT; ++ni)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int so = (((gt + ti) * 32 + (r & 3) + 8 * (r >> 2)) * Np + (gn + ni) * 32) * 4;
                                const float v = acc[gt + ti][gn + ni][r];
                                float d = MASKED ? (v - dv[ti][ni][r]) * mv[ti][ni][r] : v - dv[ti][ni][r];
                                if (MODE == 0 || MODE == 3) cmf_bstore(v, ro, voff, so);
                                if (RESID) {
                                    if (ABS) {
                                        const float sg = (v > dv[ti][ni][r]) ? 1.f : ((v < dv[ti][ni][r]) ? -1.f : 0.f);
                                        cmf_bstore(MASKED ? sg * mv[ti][ni][r] : sg, ro, voff, so);
                                    } else {
                                        cmf_bstore(d, ro, voff, so);
                                    }
                                }
                                if (LOSS) {
                                    if (PARTIAL) d = ((gt + ti) * 32 + cmf_crow(r, h) < rows) ? d : 0.f;
                                    lsum = ABS ? lsum + fabsf(d) : fmaf(d, d, lsum);
                                }
                            }
                }
        };
        if (RESID && p.loss_abs) {
            if (rows == 64) body(std::false_type{}, std::true_type{});
            else body(std::true_type{}, std::true_type{});
        } else {
            if (rows == 64) body(std::false_type{}, std::false_type{});
            else body(std::true_type{}, std::false_type{});
        }
        if (LOSS) {
            // wave sum on the DPP network (row shifts, then the two row broadcasts): the total lands in lane 63.
            // 4096 squares per wave in fp32, fp64 from the per-tile partials on (fixed order: deterministic)
            float x = lsum;
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x111, 0xf, 0xf, false)); // row_shr:1
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x112, 0xf, 0xf, false)); // row_shr:2
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x114, 0xf, 0xf, false)); // row_shr:4
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x118, 0xf, 0xf, false)); // row_shr:8
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x142, 0xa, 0xf, false)); // row_bcast:15
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x143, 0xc, 0xf, false)); // row_bcast:31
            if (WAVES == 4) {
                __shared__ float red[4];
                if (lane == 63) red[wave] = x;
                __syncthreads();
                if (tid == 0) p.partial[pidx] = ((double)red[0] + (double)red[1]) + ((double)red[2] + (double)red[3]);
            } else {
                if (lane == 63) p.partial[pidx] = (double)x;
            }
        }
    } else {
        // MODE 1 / 5 / 7: acc[ni][ti][r]: n = nw + ni*32 + crow(r,h), t = tw + ti*32 + i
        // (MODE 5 / 7 store est - data in the transposed layout; p.data is then dataT [Np][TP])
        const size_t origin = (size_t)nw * TP + p.PADL + tw;
        const size_t bytes = ((size_t)63 * TP + 64) * 4;
        const __amdgpu_buffer_rsrc_t ro = cmf_rsrc(p.out + origin, bytes);
        const __amdgpu_buffer_rsrc_t rd = cmf_rsrc(p.data + origin, bytes);
        const __amdgpu_buffer_rsrc_t rm = cmf_rsrc(p.mask + origin, bytes);
        const int voff = (4 * h * TP + i) * 4;
        const bool full = (tw + 64 <= p.T_store); // wave-uniform
        const bool abs_t = (MODE != 1) && p.loss_abs; // wave-uniform; MODE 1 (the MU path) compiles to the plain store loop
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                if (full || tw + ti * 32 + i < p.T_store) {
                    float dv[16], mv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int so = ((ni * 32 + (r & 3) + 8 * (r >> 2)) * TP + ti * 32) * 4;
                        dv[r] = (MODE != 1) ? cmf_bload(rd, voff, so) : 0.f;
                        mv[r] = MASKED ? cmf_bload(rm, voff, so) : 1.f;
                    }
                    if (MODE != 1 && abs_t) { // AbsoluteLoss: the stored quantity is the gradient sign(est - data)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int so = ((ni * 32 + (r & 3) + 8 * (r >> 2)) * TP + ti * 32) * 4;
                            const float e = acc[ni][ti][r] - dv[r];
                            const float d = (e > 0.f) ? 1.f : ((e < 0.f) ? -1.f : 0.f);
                            cmf_bstore(MASKED ? d * mv[r] : d, ro, voff, so);
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int so = ((ni * 32 + (r & 3) + 8 * (r >> 2)) * TP + ti * 32) * 4;
                            cmf_bstore(MASKED ? (acc[ni][ti][r] - dv[r]) * mv[r] : acc[ni][ti][r] - dv[r], ro, voff, so);
                        }
                    }
                }
            }
    }
}
template <int MODE, int WAVES = 4>
__device__ __forceinline__ void conv_epilogue(f32x16 (&acc)[2][2], const ConvParams &p, int t0, int n0, int wt, int wn,
                                              int i, int h, int lane, int wave, int tid, int pidx = -1)
{
    const float none[2][2][16] = {};
    conv_epilogue_<MODE, WAVES, false>(acc, p, t0, n0, wt, wn, i, h, lane, wave, tid, pidx, none);
}


template <int MODE, int NKP_CT>
__global__ __launch_bounds__(256) void conv_kernel(ConvParams p)
{
    __shared__ __attribute__((aligned(16))) float smem[CONV_HS_FLOATS + 2 * CONV_WS_FLOATS];
    float *Hs = smem;
    float *Ws = smem + CONV_HS_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int wt = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.x * 128;
    const int t0 = blockIdx.y * 128;
    const int Np = p.Np, TP = p.TP;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int LB = (p.L + 31) >> 5;
    // W chunk loader mapping: 32 rows (k) x 128 floats (n); thread -> row = tid/32 + 8q, col4 = tid%32
    const int wrow = tid >> 5, wcol = (tid & 31) * 4;

    int buf = 0;
    for (int kb = 0; kb < p.KB; ++kb) {
        int nkp = NKP_CT;
        if (NKP_CT == 0) {
            int krem = p.K - 32 * kb;
            nkp = krem >= 32 ? 16 : ((krem + 1) >> 1);
        }
        for (int lb = 0; lb < LB; ++lb) {
            const int lbeg = lb * 32;
            const int lend = (p.L < lbeg + 32) ? p.L : (lbeg + 32);
            __syncthreads(); // everyone is done with Hs / Ws of the previous block
            {   // H strip: Hs[r][c] = Ht[kb*32 + r][PADL + t0 - lbeg - 32 + c], c in [0,160): 8 threads per row, 5 float4 each
                const int r = tid >> 3, c = (tid & 7) * 4;
                const float *src = p.Ht + (size_t)(kb * 32 + r) * TP + (p.PADL + t0 - lbeg - 32 + c);
                float *dst = Hs + r * CONV_HS_STRIDE + c;
                f32x4 v[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) v[j] = *reinterpret_cast<const f32x4 *>(src + 32 * j);
#pragma unroll
                for (int j = 0; j < 5; ++j) *reinterpret_cast<f32x4 *>(dst + 32 * j) = v[j];
            }
            f32x4 wreg[4];
            {   // first W chunk of this block straight into Ws[buf]
                const float *src = p.Wt + ((size_t)lbeg * p.KB * 32 + kb * 32) * Np + n0;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    wreg[q] = *reinterpret_cast<const f32x4 *>(src + (size_t)(wrow + 8 * q) * Np + wcol);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<f32x4 *>(Ws + buf * CONV_WS_FLOATS + (wrow + 8 * q) * 128 + wcol) = wreg[q];
            }
            __syncthreads();
            for (int l = lbeg; l < lend; ++l) {
                const bool more = (l + 1 < lend);
                {   // prefetch the next lag's W chunk into registers (clamped on the last lag)
                    const int ln = more ? l + 1 : l;
                    const float *src = p.Wt + ((size_t)ln * p.KB * 32 + kb * 32) * Np + n0;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        wreg[q] = *reinterpret_cast<const f32x4 *>(src + (size_t)(wrow + 8 * q) * Np + wcol);
                }
                // keep the global prefetch ahead of the MFMA stream (the scheduler otherwise sinks it
                // below the 64 MFMAs and exposes the whole L2 latency before the barrier)
                __builtin_amdgcn_sched_barrier(0);
                // operands of lag l: H window shifted left by (l - lbeg)
                const float *hsb = Hs + h * CONV_HS_STRIDE + 32 + wt * 64 + i - (l - lbeg);
                const float *wsb = Ws + buf * CONV_WS_FLOATS + h * 128 + wn * 64 + i;
#define CONV_MFMA4(A0, A1, B0, B1)                                                                  \
    if (CONV_TRANSPOSED(MODE)) {                                                                   \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(B0, A0, acc[0][0], 0, 0, 0);               \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(B0, A1, acc[0][1], 0, 0, 0);               \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(B1, A0, acc[1][0], 0, 0, 0);               \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(B1, A1, acc[1][1], 0, 0, 0);               \
    } else {                                                                                        \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, B0, acc[0][0], 0, 0, 0);               \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, B1, acc[0][1], 0, 0, 0);               \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, B0, acc[1][0], 0, 0, 0);               \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, B1, acc[1][1], 0, 0, 0);               \
    }
                if (NKP_CT != 0) {
                    // software-pipelined: the LDS reads of step kp+1 are issued before the MFMAs of step kp
                    float a0 = hsb[0], a1 = hsb[32], b0 = wsb[0], b1 = wsb[32];
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); // the two ds_read2 of step 0
#pragma unroll
                    for (int kp = 0; kp < NKP_CT; ++kp) {
                        float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
                        if (kp + 1 < NKP_CT) {
                            na0 = hsb[(kp + 1) * 2 * CONV_HS_STRIDE];
                            na1 = hsb[(kp + 1) * 2 * CONV_HS_STRIDE + 32];
                            nb0 = wsb[(kp + 1) * 256];
                            nb1 = wsb[(kp + 1) * 256 + 32];
                            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); // reads of step kp+1 ...
                        }
                        CONV_MFMA4(a0, a1, b0, b1)
                        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);     // ... ahead of the MFMAs of step kp
                        a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
                    }
                } else {
                    for (int kp = 0; kp < nkp; ++kp) {
                        float a0 = hsb[kp * 2 * CONV_HS_STRIDE], a1 = hsb[kp * 2 * CONV_HS_STRIDE + 32];
                        float b0 = wsb[kp * 256], b1 = wsb[kp * 256 + 32];
                        CONV_MFMA4(a0, a1, b0, b1)
                    }
                }
#undef CONV_MFMA4
                if (more) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<f32x4 *>(Ws + (buf ^ 1) * CONV_WS_FLOATS + (wrow + 8 * q) * 128 + wcol) = wreg[q];
                    __syncthreads();
                    buf ^= 1;
                }
            }
        }
    }

    conv_epilogue<MODE>(acc, p, t0, n0, wt, wn, i, h, lane, wave, tid);
}

// ---------------------------------------------------------------------------------------------
// C1, fast path for K a multiple of 32: same tile and wave layout as conv_kernel, but the W operand
// is not staged in LDS.  Each wave streams its own W rows (128 contiguous bytes per half-wave,
// L2-resident: the 320 KB slab of an n tile is shared by every workgroup of that tile) straight into
// registers with buffer loads, one lag ahead (two ping-pong register sets), so the lag loop has NO
// workgroup barrier; only the H strip is shared through LDS (one barrier per 32-lag block).
// ---------------------------------------------------------------------------------------------
// NBL = live 32-column n blocks of the wave's 64 columns (2, or 1 when the second block is all padding: at N = 2000 the
// last 64-column tile holds 16 real columns, and skipping its dead half is 1.6 % of the launch's MFMAs)
// NKP = k pairs per lag: 16 for a whole 32-row k block; the few-component kernel (conv_small_kernel, K <= 16) runs the same
// lag loop over the ceil(K / 2) pairs that hold data
template <int NBL = 2, int NKP = 16>
__device__ __forceinline__ void conv2_load_w(float (&w)[NKP][2], __amdgpu_buffer_rsrc_t wr, int woff, int lag, int lagbytes, int rowbytes)
{
#pragma unroll
    for (int kp = 0; kp < NKP; ++kp) {
        w[kp][0] = cmf_bload(wr, woff, lag * lagbytes + kp * 2 * rowbytes);
        if (NBL == 2) w[kp][1] = cmf_bload(wr, woff + 128, lag * lagbytes + kp * 2 * rowbytes);
    }
}

// FIRST: the accumulators hold nothing yet -- the kp = 0 MFMAs take a zero C operand (an inline constant) instead of
// 64 register writes of an explicit zero fill
template <int MODE, int STRIDE = CONV_HS_STRIDE, bool FIRST = false, int NBL = 2, int NKP = 16>
__device__ __forceinline__ void conv2_lag(f32x16 (&acc)[2][2], const float *hsb, const float (&w)[NKP][2])
{
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float a0 = hsb[0], a1 = hsb[32];
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
    for (int kp = 0; kp < NKP; ++kp) {
        float na0 = 0.f, na1 = 0.f;
        if (kp + 1 < NKP) {
            na0 = hsb[(kp + 1) * 2 * STRIDE];
            na1 = hsb[(kp + 1) * 2 * STRIDE + 32];
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        const bool zc = FIRST && kp == 0;
        if (CONV_TRANSPOSED(MODE)) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[kp][0], a0, zc ? zero16 : acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[kp][0], a1, zc ? zero16 : acc[0][1], 0, 0, 0);
            if (NBL == 2) {
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[kp][1], a0, zc ? zero16 : acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[kp][1], a1, zc ? zero16 : acc[1][1], 0, 0, 0);
            }
        } else {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, w[kp][0], zc ? zero16 : acc[0][0], 0, 0, 0);
            if (NBL == 2) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, w[kp][1], zc ? zero16 : acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, w[kp][0], zc ? zero16 : acc[1][0], 0, 0, 0);
            if (NBL == 2) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, w[kp][1], zc ? zero16 : acc[1][1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NBL == 2 ? 4 : 2, 0);
        a0 = na0; a1 = na1;
    }
}

// The lag pairs of one (kb, lb) block for a wave: wA holds lag 0 on entry.  `first`: the block's first MFMAs start the
// accumulators (zero C operand).
template <int MODE, int STRIDE, int NBL, int NKP = 16>
__device__ __forceinline__ void conv2_lag_pairs(f32x16 (&acc)[2][2], const float *hsb, float (&wA)[NKP][2], float (&wB)[NKP][2],
                                                __amdgpu_buffer_rsrc_t wr, int woff, int npair, bool first, int lagbytes, int rowbytes)
{
    int pr = 0;
    if (first) { // peeled first lag pair
        conv2_load_w<NBL, NKP>(wB, wr, woff, 1, lagbytes, rowbytes);
        __builtin_amdgcn_sched_barrier(0);
        conv2_lag<MODE, STRIDE, true, NBL, NKP>(acc, hsb, wA);
        conv2_load_w<NBL, NKP>(wA, wr, woff, (1 < npair) ? 2 : 0, lagbytes, rowbytes);
        __builtin_amdgcn_sched_barrier(0);
        conv2_lag<MODE, STRIDE, false, NBL, NKP>(acc, hsb - 1, wB);
        pr = 1;
    }
    for (; pr < npair; ++pr) {
        const int l0 = 2 * pr; // lag offsets inside the block
        conv2_load_w<NBL, NKP>(wB, wr, woff, l0 + 1, lagbytes, rowbytes);
        __builtin_amdgcn_sched_barrier(0);
        conv2_lag<MODE, STRIDE, false, NBL, NKP>(acc, hsb - l0, wA);
        conv2_load_w<NBL, NKP>(wA, wr, woff, (pr + 1 < npair) ? l0 + 2 : l0, lagbytes, rowbytes);
        __builtin_amdgcn_sched_barrier(0);
        conv2_lag<MODE, STRIDE, false, NBL, NKP>(acc, hsb - l0 - 1, wB);
    }
}

// the accumulators of the n block that was not computed (NBL == 1)
template <int MODE>
__device__ __forceinline__ void conv2_clear_dead(f32x16 (&acc)[2][2])
{
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        if (CONV_TRANSPOSED(MODE)) { acc[1][0][q] = 0.f; acc[1][1][q] = 0.f; }
        else { acc[0][1][q] = 0.f; acc[1][1][q] = 0.f; }
    }
}


template <int MODE>
__global__ __launch_bounds__(256, 3) void conv2_kernel(ConvParams p)
{
    __shared__ __attribute__((aligned(16))) float Hs[CONV_HS_FLOATS];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int wt = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.x * 128;
    const int t0 = blockIdx.y * 128;
    const int Np = p.Np, TP = p.TP;
    const int K32 = p.KB * 32;

    f32x16 acc[2][2]; // started by the zero-C MFMAs of the first lag
    const int LB = (p.L + 31) >> 5;
    const int rowbytes = Np * 4;
    const int lagbytes = K32 * Np * 4;
    const int woff = (h * Np + n0 + wn * 64 + i) * 4; // per-lane part of the W address
    const bool half = n0 + wn * 64 + 32 >= p.N;       // wave-uniform: the wave's second n block is padding
    float wA[16][2], wB[16][2];

    for (int kb = 0; kb < p.KB; ++kb) {
        for (int lb = 0; lb < LB; ++lb) {
            const int lbeg = lb * 32;
            const int lend = (p.L < lbeg + 32) ? p.L : (lbeg + 32);
            const int npair = (lend - lbeg + 1) >> 1; // lags are processed in pairs; Wt is zero-padded to Lp
            // descriptor base: Wt[lbeg][kb*32][0]
            const __amdgpu_buffer_rsrc_t wr = cmf_rsrc(p.Wt + ((size_t)lbeg * K32 + kb * 32) * Np, (size_t)(2 * npair) * lagbytes);
            if (!half) conv2_load_w<2>(wA, wr, woff, 0, lagbytes, rowbytes); else conv2_load_w<1>(wA, wr, woff, 0, lagbytes, rowbytes);
            __syncthreads(); // everyone is done with Hs of the previous block
            {   // H strip: Hs[r][c] = Ht[kb*32 + r][PADL + t0 - lbeg - 32 + c], c in [0,160): 8 threads per row, 5 float4 each
                const int r = tid >> 3, c = (tid & 7) * 4;
                const float *src = p.Ht + (size_t)(kb * 32 + r) * TP + (p.PADL + t0 - lbeg - 32 + c);
                float *dst = Hs + r * CONV_HS_STRIDE + c;
                f32x4 v[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) v[j] = *reinterpret_cast<const f32x4 *>(src + 32 * j);
#pragma unroll
                for (int j = 0; j < 5; ++j) *reinterpret_cast<f32x4 *>(dst + 32 * j) = v[j];
            }
            __syncthreads();
            const float *hsb = Hs + h * CONV_HS_STRIDE + 32 + wt * 64 + i;
            if (!half) conv2_lag_pairs<MODE, CONV_HS_STRIDE, 2>(acc, hsb, wA, wB, wr, woff, npair, kb == 0 && lb == 0, lagbytes, rowbytes);
            else conv2_lag_pairs<MODE, CONV_HS_STRIDE, 1>(acc, hsb, wA, wB, wr, woff, npair, kb == 0 && lb == 0, lagbytes, rowbytes);
        }
    }
    if (half) conv2_clear_dead<MODE>(acc);
    conv_epilogue<MODE>(acc, p, t0, n0, wt, wn, i, h, lane, wave, tid);
}

// ---------------------------------------------------------------------------------------------
// C1c: tensor_conv with one-wave workgroups (K a multiple of 32).  The workgroup is one wave and one 64 (t) x 64 (n)
// output tile; its H strip (32 k rows x 96 columns: 64 + the 32-lag halo) lives in a wave-private 12 KB of LDS, so
// there is no workgroup barrier and the hardware dispatcher balances the chip in units of a quarter of conv2's
// tile: the per-CU quantisation (25 vs 24.4 big tiles at config 2, 4 vs 3.06 on a T/8 shard) and the drain at the
// end of the launch shrink fourfold.  Main loop and epilogue are conv2's.
// ---------------------------------------------------------------------------------------------
#define CONV3_STRIDE 96
// SC1: the H strip is read with agent-scope loads -- Ht is being written by a CONCURRENT kernel (the HALS row pipeline) whose
// sweepers have published the tile's columns (conv3_chase_kernel)
template <int MODE, int NBL = 2, int NKP = 16, bool PREQ = false, bool SC1 = false>
__device__ __forceinline__ void conv3_tile(const ConvParams &p, float *Hs, int t0, int n0, int lane, int pidx)
{
    const int i = lane & 31, h = lane >> 5;
    const int Np = p.Np, TP = p.TP;
    const int K32 = p.KB * 32;

    f32x16 acc[2][2]; // started by the zero-C MFMAs of the first lag
    const int LB = (p.L + 31) >> 5;
    const int rowbytes = Np * 4;
    const int lagbytes = K32 * Np * 4;
    const int woff = (h * Np + n0 + i) * 4; // per-lane part of the W address
    float wA[NKP][2], wB[NKP][2];
    constexpr int NQ = (2 * NKP + 7) / 8; // 8-row passes of the strip load that hold live k rows
    // Few components (NKP <= 4), loss + store, SHORT launches (PREQ: the host asks for it when there are at most four tiles per SIMD):
    // the tile's MFMA loop is short (240 MFMAs at K = 5, L = 20) and the data tile the loss needs was loaded in the epilogue, an exposed
    // HBM round trip per tile; here it is requested before the loop (64 registers that the few k pairs leave free) with the epilogue's
    // descriptor and offsets, and the epilogue finds it there (protocol shape: 37.8 -> 35.8 us).  On a launch of many rounds (N = 2000:
    // eight tiles per SIMD slot, bandwidth-bound) the W rows of the first lags queue behind these 64 loads and it costs 7 %: not used there.
    constexpr bool PRE = (PREQ && MODE == 3 && NKP <= 4 && NBL == 2);
    float dpre[2][2][16];
    if (PRE) {
        int rows = p.T_store - t0;
        rows = rows < 0 ? 0 : (rows > 64 ? 64 : rows);
        const size_t origin = (size_t)(p.PADL + t0) * Np + n0;
        const __amdgpu_buffer_rsrc_t rd = cmf_rsrc(p.data + origin, rows ? ((size_t)(rows - 1) * Np + 64) * 4 : 0);
        const int voff = (4 * h * Np + i) * 4;
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    dpre[ti][ni][r] = cmf_bload(rd, voff, ((ti * 32 + (r & 3) + 8 * (r >> 2)) * Np + ni * 32) * 4);
    }

    for (int kb = 0; kb < p.KB; ++kb) {
        for (int lb = 0; lb < LB; ++lb) {
            const int lbeg = lb * 32;
            const int lend = (p.L < lbeg + 32) ? p.L : (lbeg + 32);
            const int npair = (lend - lbeg + 1) >> 1; // lags are processed in pairs; Wt is zero-padded to Lp
            const __amdgpu_buffer_rsrc_t wr = cmf_rsrc(p.Wt + ((size_t)lbeg * K32 + kb * 32) * Np, (size_t)(2 * npair) * lagbytes);
            conv2_load_w<NBL, NKP>(wA, wr, woff, 0, lagbytes, rowbytes);
            {   // H strip: Hs[r][c] = Ht[kb*32 + r][PADL + t0 - lbeg - 32 + c], c in [0,96): 8 lanes per row, 8 rows per pass
                const int r = lane >> 3, c = (lane & 7) * 4;
                const float *src = p.Ht + (size_t)(kb * 32 + r) * TP + (p.PADL + t0 - lbeg - 32 + c);
                float *dst = Hs + r * CONV3_STRIDE + c;
                f32x4 v[3 * NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        v[q * 3 + j] = SC1 ? cmf_load4_sc1(src + (size_t)(8 * q) * TP + 32 * j) : *reinterpret_cast<const f32x4 *>(src + (size_t)(8 * q) * TP + 32 * j);
                __builtin_amdgcn_wave_barrier(); // every lane is done reading the previous strip
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int j = 0; j < 3; ++j) *reinterpret_cast<f32x4 *>(dst + (8 * q) * CONV3_STRIDE + 32 * j) = v[q * 3 + j];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const float *hsb = Hs + h * CONV3_STRIDE + 32 + i;
            conv2_lag_pairs<MODE, CONV3_STRIDE, NBL, NKP>(acc, hsb, wA, wB, wr, woff, npair, kb == 0 && lb == 0, lagbytes, rowbytes);
        }
    }
    if (NBL == 1) conv2_clear_dead<MODE>(acc);
    if constexpr (PRE) conv_epilogue_<MODE, 1, true>(acc, p, t0, n0, 0, 0, i, h, lane, 0, lane, pidx, dpre);
    else conv_epilogue<MODE, 1>(acc, p, t0, n0, 0, 0, i, h, lane, 0, lane, pidx);
}

// One 32 x 32 block of the epilogue (the quarter tiles of conv3_kernel): acc[r] is t = tb + crow(r,h), n = nb + i, or
// in the transposed modes n = nb + crow(r,h), t = tb + i.  Same buffer-addressed loads / stores as conv_epilogue.
template <int MODE>
__device__ __forceinline__ void conv_epilogue_block(const f32x16 &acc, const ConvParams &p, int tb, int nb, int i, int h, int lane, int pidx)
{
    const int Np = p.Np, TP = p.TP;
    constexpr bool LOSS = (MODE == 2 || MODE == 3 || MODE == 4 || MODE == 6);
    constexpr bool RESID = (MODE == 4 || MODE == 6);
    constexpr bool MASKED = (MODE == 6 || MODE == 7);
    const bool abs_loss = (MODE >= 4) && p.loss_abs; // wave-uniform; a quarter tile's 16-element epilogue takes the selects
    if (!CONV_TRANSPOSED(MODE)) {
        int rows = p.T_store - tb;
        rows = rows < 0 ? 0 : (rows > 32 ? 32 : rows);
        const size_t origin = (size_t)(p.PADL + tb) * Np + nb;
        const size_t bytes = rows ? ((size_t)(rows - 1) * Np + 32) * 4 : 0;
        const __amdgpu_buffer_rsrc_t ro = cmf_rsrc(p.out + origin, bytes);
        const __amdgpu_buffer_rsrc_t rd = cmf_rsrc(p.data + origin, bytes);
        const __amdgpu_buffer_rsrc_t rm = cmf_rsrc(p.mask + origin, bytes);
        const int voff = (4 * h * Np + i) * 4;
        float dv[16], mv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int so = (((r & 3) + 8 * (r >> 2)) * Np) * 4;
            dv[r] = LOSS ? cmf_bload(rd, voff, so) : 0.f; // rows past T_store read as 0 (masked below)
            mv[r] = MASKED ? cmf_bload(rm, voff, so) : 1.f;
        }
        float lsum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int so = (((r & 3) + 8 * (r >> 2)) * Np) * 4;
            const float v = acc[r];
            float d = MASKED ? (v - dv[r]) * mv[r] : v - dv[r];
            if (MODE == 0 || MODE == 3) cmf_bstore(v, ro, voff, so);
            if (RESID) {
                const float sg = (v > dv[r]) ? 1.f : ((v < dv[r]) ? -1.f : 0.f);
                cmf_bstore(abs_loss ? (MASKED ? sg * mv[r] : sg) : d, ro, voff, so);
            }
            if (LOSS) {
                d = (cmf_crow(r, h) < rows) ? d : 0.f;
                lsum = abs_loss ? lsum + fabsf(d) : fmaf(d, d, lsum);
            }
        }
        if (LOSS) {
            float x = lsum;
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x111, 0xf, 0xf, false)); // row_shr:1
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x112, 0xf, 0xf, false)); // row_shr:2
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x114, 0xf, 0xf, false)); // row_shr:4
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x118, 0xf, 0xf, false)); // row_shr:8
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x142, 0xa, 0xf, false)); // row_bcast:15
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x143, 0xc, 0xf, false)); // row_bcast:31
            if (lane == 63) p.partial[pidx] = (double)x;
        }
    } else {
        const size_t origin = (size_t)nb * TP + p.PADL + tb;
        const size_t bytes = ((size_t)31 * TP + 32) * 4;
        const __amdgpu_buffer_rsrc_t ro = cmf_rsrc(p.out + origin, bytes);
        const __amdgpu_buffer_rsrc_t rd = cmf_rsrc(p.data + origin, bytes);
        const __amdgpu_buffer_rsrc_t rm = cmf_rsrc(p.mask + origin, bytes);
        const int voff = (4 * h * TP + i) * 4;
        if (tb + i < p.T_store) {
            float dv[16], mv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int so = (((r & 3) + 8 * (r >> 2)) * TP) * 4;
                dv[r] = (MODE != 1) ? cmf_bload(rd, voff, so) : 0.f;
                mv[r] = MASKED ? cmf_bload(rm, voff, so) : 1.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int so = (((r & 3) + 8 * (r >> 2)) * TP) * 4;
                float d = acc[r] - dv[r];
                if (abs_loss) d = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
                cmf_bstore(MASKED ? d * mv[r] : d, ro, voff, so);
            }
        }
    }
}

// one lag of a quarter tile: 16 k pairs, one MFMA each, into the single accumulator -- the same k / lag order per
// output element as conv2_lag, so a quarter tile's results are bitwise those of the 64 x 64 tile path
template <int MODE, bool FIRST = false, int NKP = 16>
__device__ __forceinline__ void convq_lag(f32x16 &acc, const float *hsb, const float (&w)[NKP])
{
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float a0 = hsb[0];
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
    for (int kp = 0; kp < NKP; ++kp) {
        float na0 = 0.f;
        if (kp + 1 < NKP) {
            na0 = hsb[(kp + 1) * 2 * CONV3_STRIDE];
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        const bool zc = FIRST && kp == 0;
        if (CONV_TRANSPOSED(MODE)) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[kp], a0, zc ? zero16 : acc, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, w[kp], zc ? zero16 : acc, 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        a0 = na0;
    }
}

template <int NKP = 16>
__device__ __forceinline__ void convq_load_w(float (&w)[NKP], __amdgpu_buffer_rsrc_t wr, int woff, int lag, int lagbytes, int rowbytes)
{
#pragma unroll
    for (int kp = 0; kp < NKP; ++kp) w[kp] = cmf_bload(wr, woff, lag * lagbytes + kp * 2 * rowbytes);
}

// A quarter tile: one wave, one 32 (t) x 32 (n) block at (t0, n0); H strip = 32 k rows x 64 columns (32 + the 32-lag halo)
// NKP: k pairs per lag that hold data (16 = a whole k block; the few-component kernel passes ceil(K / 2): rows k >= K of Ht and Wt are zero)
template <int MODE, int NKP = 16, bool SC1 = false>
__device__ __forceinline__ void conv3_quarter(const ConvParams &p, float *Hs, int t0, int n0, int lane, int pidx)
{
    const int i = lane & 31, h = lane >> 5;
    const int Np = p.Np, TP = p.TP;
    const int K32 = p.KB * 32;
    f32x16 acc;
    const int LB = (p.L + 31) >> 5;
    const int rowbytes = Np * 4;
    const int lagbytes = K32 * Np * 4;
    const int woff = (h * Np + n0 + i) * 4;
    float wA[NKP], wB[NKP];
    constexpr int NQ = (2 * NKP + 7) / 8 < 4 ? (2 * NKP + 7) / 8 : 4; // 8-row passes of the strip load that hold live k rows
    for (int kb = 0; kb < p.KB; ++kb) {
        for (int lb = 0; lb < LB; ++lb) {
            const int lbeg = lb * 32;
            const int lend = (p.L < lbeg + 32) ? p.L : (lbeg + 32);
            const int npair = (lend - lbeg + 1) >> 1;
            const __amdgpu_buffer_rsrc_t wr = cmf_rsrc(p.Wt + ((size_t)lbeg * K32 + kb * 32) * Np, (size_t)(2 * npair) * lagbytes);
            convq_load_w<NKP>(wA, wr, woff, 0, lagbytes, rowbytes);
            {   // H strip: Hs[r][c] = Ht[kb*32 + r][PADL + t0 - lbeg - 32 + c], c in [0,64): 8 lanes per row, 8 rows per pass
                const int r = lane >> 3, c = (lane & 7) * 4;
                const float *src = p.Ht + (size_t)(kb * 32 + r) * TP + (p.PADL + t0 - lbeg - 32 + c);
                float *dst = Hs + r * CONV3_STRIDE + c;
                f32x4 v[8];
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        v[q * 2 + j] = SC1 ? cmf_load4_sc1(src + (size_t)(8 * q) * TP + 32 * j) : *reinterpret_cast<const f32x4 *>(src + (size_t)(8 * q) * TP + 32 * j);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4 *>(dst + (8 * q) * CONV3_STRIDE + 32 * j) = v[q * 2 + j];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const float *hsb = Hs + h * CONV3_STRIDE + 32 + i;
            int pr = 0;
            if (kb == 0 && lb == 0) {
                convq_load_w<NKP>(wB, wr, woff, 1, lagbytes, rowbytes);
                __builtin_amdgcn_sched_barrier(0);
                convq_lag<MODE, true, NKP>(acc, hsb, wA);
                convq_load_w<NKP>(wA, wr, woff, (1 < npair) ? 2 : 0, lagbytes, rowbytes);
                __builtin_amdgcn_sched_barrier(0);
                convq_lag<MODE, false, NKP>(acc, hsb - 1, wB);
                pr = 1;
            }
            for (; pr < npair; ++pr) {
                const int l0 = 2 * pr;
                convq_load_w<NKP>(wB, wr, woff, l0 + 1, lagbytes, rowbytes);
                __builtin_amdgcn_sched_barrier(0);
                convq_lag<MODE, false, NKP>(acc, hsb - l0, wA);
                convq_load_w<NKP>(wA, wr, woff, (pr + 1 < npair) ? l0 + 2 : l0, lagbytes, rowbytes);
                __builtin_amdgcn_sched_barrier(0);
                convq_lag<MODE, false, NKP>(acc, hsb - l0 - 1, wB);
            }
        }
    }
    conv_epilogue_block<MODE>(acc, p, t0, n0, i, h, lane, pidx);
}

// ---- sixteenth tiles: one wave, one 16 (t) x 16 (n) block on v_mfma_f32_16x16x4_f32 -----------------------------------
// When the remainder of a launch is so small that even its quarter tiles would leave most SIMDs idle (64 tiles = 256
// quarters on 1024 SIMDs at a T/8 shard), it is cut sixteen ways instead: 1024 pieces of a quarter of the work each.
// Operand maps (16x16x4 f32): lane = 16*kq + j.  A: A[row j][k kq], B: B[k kq][col j], C/D: col = j, row = 4*kq + reg.
// The k sum runs four at a time here (two at a time in the 32x32x2 kernels), so a sixteenth tile's est differs from the
// whole-tile path's at rounding level; every launch with the same shape cuts the same tiles, so results stay reproducible.
#define CONV16_STRIDE 80 // k rows 80 floats apart: the four k rows a ds_read touches land in different bank groups
template <int MODE>
__device__ __forceinline__ void conv16_epilogue(const f32x4 &acc, const ConvParams &p, int tb, int nb, int j, int kq, int lane, int pidx)
{
    const int Np = p.Np, TP = p.TP;
    constexpr bool LOSS = (MODE == 2 || MODE == 3 || MODE == 4 || MODE == 6);
    constexpr bool RESID = (MODE == 4 || MODE == 6);
    constexpr bool MASKED = (MODE == 6 || MODE == 7);
    const bool abs_loss = (MODE >= 4) && p.loss_abs;
    if (!CONV_TRANSPOSED(MODE)) { // acc[r]: t = tb + 4*kq + r, n = nb + j
        int rows = p.T_store - tb;
        rows = rows < 0 ? 0 : (rows > 16 ? 16 : rows);
        const size_t origin = (size_t)(p.PADL + tb) * Np + nb;
        const size_t bytes = rows ? ((size_t)(rows - 1) * Np + 16) * 4 : 0;
        const __amdgpu_buffer_rsrc_t ro = cmf_rsrc(p.out + origin, bytes);
        const __amdgpu_buffer_rsrc_t rd = cmf_rsrc(p.data + origin, bytes);
        const __amdgpu_buffer_rsrc_t rm = cmf_rsrc(p.mask + origin, bytes);
        const int voff = (4 * kq * Np + j) * 4;
        float dv[4], mv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dv[r] = LOSS ? cmf_bload(rd, voff, r * Np * 4) : 0.f;
            mv[r] = MASKED ? cmf_bload(rm, voff, r * Np * 4) : 1.f;
        }
        float lsum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = acc[r];
            float d = MASKED ? (v - dv[r]) * mv[r] : v - dv[r];
            if (MODE == 0 || MODE == 3) cmf_bstore(v, ro, voff, r * Np * 4);
            if (RESID) {
                const float sg = (v > dv[r]) ? 1.f : ((v < dv[r]) ? -1.f : 0.f);
                cmf_bstore(abs_loss ? (MASKED ? sg * mv[r] : sg) : d, ro, voff, r * Np * 4);
            }
            if (LOSS) {
                d = (4 * kq + r < rows) ? d : 0.f;
                lsum = abs_loss ? lsum + fabsf(d) : fmaf(d, d, lsum);
            }
        }
        if (LOSS) {
            float x = lsum;
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x111, 0xf, 0xf, false)); // row_shr:1
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x112, 0xf, 0xf, false)); // row_shr:2
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x114, 0xf, 0xf, false)); // row_shr:4
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x118, 0xf, 0xf, false)); // row_shr:8
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x142, 0xa, 0xf, false)); // row_bcast:15
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x143, 0xc, 0xf, false)); // row_bcast:31
            if (lane == 63) p.partial[pidx] = (double)x;
        }
    } else { // acc[r]: n = nb + 4*kq + r, t = tb + j
        const size_t origin = (size_t)nb * TP + p.PADL + tb;
        const size_t bytes = ((size_t)15 * TP + 16) * 4;
        const __amdgpu_buffer_rsrc_t ro = cmf_rsrc(p.out + origin, bytes);
        const __amdgpu_buffer_rsrc_t rd = cmf_rsrc(p.data + origin, bytes);
        const __amdgpu_buffer_rsrc_t rm = cmf_rsrc(p.mask + origin, bytes);
        const int voff = (4 * kq * TP + j) * 4;
        if (tb + j < p.T_store) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dvr = (MODE != 1) ? cmf_bload(rd, voff, r * TP * 4) : 0.f;
                const float mvr = MASKED ? cmf_bload(rm, voff, r * TP * 4) : 1.f;
                float d = acc[r] - dvr;
                if (abs_loss) d = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
                cmf_bstore(MASKED ? d * mvr : d, ro, voff, r * TP * 4);
            }
        }
    }
}

template <int MODE>
__device__ __forceinline__ void conv16_lag(f32x4 &acc, const float *hsb, const float (&w)[8])
{
    float a = hsb[0];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const float na = (ks + 1 < 8) ? hsb[(ks + 1) * 4 * CONV16_STRIDE] : 0.f;
        if (CONV_TRANSPOSED(MODE)) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ks], a, acc, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w[ks], acc, 0, 0, 0);
        a = na;
    }
}

__device__ __forceinline__ void conv16_load_w(float (&w)[8], __amdgpu_buffer_rsrc_t wr, int woff, int lag, int lagbytes, int rowbytes)
{
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) w[ks] = cmf_bload(wr, woff, lag * lagbytes + ks * 4 * rowbytes);
}

template <int MODE, bool SC1 = false>
__device__ __forceinline__ void conv3_sixteenth(const ConvParams &p, float *Hs, int t0, int n0, int lane, int pidx)
{
    const int j = lane & 15, kq = lane >> 4;
    const int Np = p.Np, TP = p.TP;
    const int K32 = p.KB * 32;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int LB = (p.L + 31) >> 5;
    const int rowbytes = Np * 4;
    const int lagbytes = K32 * Np * 4;
    const int woff = (kq * Np + j) * 4; // the block's n0 goes into the descriptor base
    float wA[8], wB[8];
    for (int kb = 0; kb < p.KB; ++kb) {
        for (int lb = 0; lb < LB; ++lb) {
            const int lbeg = lb * 32;
            const int lend = (p.L < lbeg + 32) ? p.L : (lbeg + 32);
            const int npair = (lend - lbeg + 1) >> 1; // lags in pairs; Wt is zero-padded to Lp
            const __amdgpu_buffer_rsrc_t wr = cmf_rsrc(p.Wt + ((size_t)lbeg * K32 + kb * 32) * Np + n0, (size_t)(2 * npair) * lagbytes - (size_t)n0 * 4);
            conv16_load_w(wA, wr, woff, 0, lagbytes, rowbytes);
            {   // H strip: Hs[r][c] = Ht[kb*32 + r][PADL + t0 - lbeg - 32 + c], c in [0,64): 8 lanes per row, 8 rows per pass
                const int r = lane >> 3, c = (lane & 7) * 4;
                const float *src = p.Ht + (size_t)(kb * 32 + r) * TP + (p.PADL + t0 - lbeg - 32 + c);
                float *dst = Hs + r * CONV16_STRIDE + c;
                f32x4 v[8];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
                        v[q * 2 + jj] = SC1 ? cmf_load4_sc1(src + (size_t)(8 * q) * TP + 32 * jj) : *reinterpret_cast<const f32x4 *>(src + (size_t)(8 * q) * TP + 32 * jj);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) *reinterpret_cast<f32x4 *>(dst + (8 * q) * CONV16_STRIDE + 32 * jj) = v[q * 2 + jj];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const float *hsb = Hs + kq * CONV16_STRIDE + 32 + j;
            for (int pr = 0; pr < npair; ++pr) {
                const int l0 = 2 * pr;
                conv16_load_w(wB, wr, woff, l0 + 1, lagbytes, rowbytes);
                conv16_lag<MODE>(acc, hsb - l0, wA);
                conv16_load_w(wA, wr, woff, (pr + 1 < npair) ? l0 + 2 : l0, lagbytes, rowbytes);
                conv16_lag<MODE>(acc, hsb - l0 - 1, wB);
            }
        }
    }
    conv16_epilogue<MODE>(acc, p, t0, n0, j, kq, lane, pidx);
}

// grid: n_full + P * (tiles - n_full) one-wave workgroups (P = 4 or 16).  Workgroups [0, n_full) are whole 64 x 64 tiles
// (tile index = n tile fastest); the remaining tiles -- the part of the grid that would otherwise run as a thin last
// round on a few SIMDs while the rest of the chip idles (3136 tiles on 3072 wave slots at a T/8 shard) -- are cut into
// P pieces each (32 x 32 quarter tiles, or 16 x 16 sixteenth tiles when there are too few remainder tiles for the
// quarters to reach every SIMD), dispatched last, so that the tail is spread over P times as many SIMDs.
template <int MODE>
__global__ __launch_bounds__(64, 3) void conv3_kernel(ConvParams p, int gx, int n_full, int pieces)
{
    __shared__ __attribute__((aligned(16))) float Hs[32 * CONV3_STRIDE];
    const int b = blockIdx.x;
    if (b < n_full) {
        const int n0 = (b % gx) * 64;
        if (n0 + 32 < p.N) conv3_tile<MODE, 2>(p, Hs, (b / gx) * 64, n0, threadIdx.x, b);
        else conv3_tile<MODE, 1>(p, Hs, (b / gx) * 64, n0, threadIdx.x, b);
    } else if (pieces == 4) {
        const int q = b - n_full, tile = n_full + (q >> 2), sub = q & 3;
        conv3_quarter<MODE>(p, Hs, (tile / gx) * 64 + (sub >> 1) * 32, (tile % gx) * 64 + (sub & 1) * 32, threadIdx.x, b);
    } else {
        const int q = b - n_full, tile = n_full + (q >> 4), sub = q & 15;
        conv3_sixteenth<MODE>(p, Hs, (tile / gx) * 64 + (sub >> 2) * 16, (tile % gx) * 64 + (sub & 3) * 16, threadIdx.x, b);
    }
}

// conv3_kernel on tile rows [row0, ...) of the recording, optionally CHASING the HALS row pipeline (hals_h_persist_kernel) that is
// still writing H on other CUs: tile row r (columns 64 r .. 64 r + 63) reads H[:, 64 r - (L-1) .. 64 r + 63], final once the sweeper of
// the LAST row of H has published block r (rows above are further ahead by construction of the pipeline, and a sweeper publishes a
// block only after its agent-scope stores of H and D have completed).  gate != NULL: every workgroup first waits -- bounded, like
// every wait of the pipeline -- until *gate >= r + 1, then reads the strip with agent-scope loads.  A wait that runs out, or a
// pipeline that has aborted, ends the workgroup without output: the host sees the status word and redoes sweep and conv.
// gate == NULL: plain offset tiles (the part of the conv that runs behind the pipeline).  pidx0: first loss partial of this launch.
// SC1: the strips are read with agent-scope loads (the chasing launch); false: plain loads (tile rows at an offset, nothing concurrent).
template <int MODE, bool SC1>
__global__ __launch_bounds__(64, 3) void conv3_chase_kernel(ConvParams p, int gx, int n_full, int pieces, int row0, int pidx0,
                                                            const int *gate, int *abort_word, int *host_status)
{
    __shared__ __attribute__((aligned(16))) float Hs[32 * CONV3_STRIDE];
    const int b = blockIdx.x;
    const int tile = b < n_full ? b : n_full + ((b - n_full) >> (pieces == 4 ? 2 : 4));
    const int row = row0 + tile / gx, n0 = (tile % gx) * 64;
    if (gate) {
        bool ok = false;
#pragma nounroll
        for (int n = 0; n < (1 << 19); ++n) { // ~2 s: three orders of magnitude above the pipeline's span
            const int v = cmf_load_sc1(gate), a = cmf_load_sc1(abort_word);
            ok = v >= row + 1;
            if ((int)ok | (int)(a != 0)) break;
            __builtin_amdgcn_s_sleep(100); // (~3 us: a thousand waiting waves must not flood the flag's channel)
        }
        if (!ok) {
            if (threadIdx.x == 0) {
                cmf_store_sc1(abort_word, 1);
                __hip_atomic_store(host_status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return;
        }
    }
    if (b < n_full) {
        if (n0 + 32 < p.N) conv3_tile<MODE, 2, 16, false, SC1>(p, Hs, row * 64, n0, threadIdx.x, pidx0 + b);
        else conv3_tile<MODE, 1, 16, false, SC1>(p, Hs, row * 64, n0, threadIdx.x, pidx0 + b);
    } else if (pieces == 4) {
        const int sub = (b - n_full) & 3;
        conv3_quarter<MODE, 16, SC1>(p, Hs, row * 64 + (sub >> 1) * 32, n0 + (sub & 1) * 32, threadIdx.x, pidx0 + b);
    } else {
        const int sub = (b - n_full) & 15;
        conv3_sixteenth<MODE, SC1>(p, Hs, row * 64 + (sub >> 2) * 16, n0 + (sub & 3) * 16, threadIdx.x, pidx0 + b);
    }
}

// C1 for few components (K <= 16, one k block): conv3's one-wave 64 x 64 tiles with the lag loop running over the NKP =
// ceil(K / 2) k pairs that hold data instead of all 16 -- K = 5 issues 3/16 of the MFMAs of the padded k block.
// grid: n_full whole tiles + 4 quarter pieces for each of the remaining tiles, dispatched last (conv3_kernel's scheme): the
// reference's protocol shape is 3128 tiles on 1024 SIMDs -- 3.05 per SIMD, so that a handful of SIMDs ran a fourth whole tile
// while the chip waited (36 us for 19.5 us of MFMA issue per SIMD); in quarters the excess is a quarter tile on a quarter of the SIMDs.
template <int MODE, int NKP, bool PREQ = false>
__global__ __launch_bounds__(64, (NKP <= 4 && (MODE <= 1 || MODE == 5)) ? 4 : 3) void conv_small_kernel(ConvParams p, int gx, int n_full)
{
    __shared__ __attribute__((aligned(16))) float Hs[32 * CONV3_STRIDE];
    const int b = blockIdx.x;
    if (b < n_full) {
        const int n0 = (b % gx) * 64;
        if (n0 + 32 < p.N) conv3_tile<MODE, 2, NKP, PREQ>(p, Hs, (b / gx) * 64, n0, threadIdx.x, b);
        else conv3_tile<MODE, 1, NKP>(p, Hs, (b / gx) * 64, n0, threadIdx.x, b);
    } else {
        const int q = b - n_full, tile = n_full + (q >> 2), sub = q & 3;
        conv3_quarter<MODE, NKP>(p, Hs, (tile / gx) * 64 + (sub >> 1) * 32, (tile % gx) * 64 + (sub & 1) * 32, threadIdx.x, b);
    }
}

// ---------------------------------------------------------------------------------------------
// C2: out[l][k][n] = sum_t H[t-l][k] * X[t][n]   (numW with X=data, denomW with X=est).
// One wave = one 32-wide n block, one 32-wide k block, one time chunk, 2*LP lags; every operand
// is loaded from global memory directly in MFMA layout (H rows and X rows are both 128-byte
// contiguous per half-wave), so there is no LDS and no barrier.  The A operand of lag l at
// step s (t = 2s, 2s+1) is the row pair R(2s-l); even lags reuse E(s-m) = R(2(s-m)), odd lags
// O(s-m) = R(2(s-m)-1), so each step loads just E(s), O(s) and the X pair and issues 2*LP
// MFMAs from a register ring.  Partial sums go to a per-chunk slab (deterministic; summed by
// slab_sum_kernel / the W update).
// ---------------------------------------------------------------------------------------------
struct HxtParams {
    const float *H;  // [TP][K32]
    const float *X0; // data [TP][Np]
    const float *X1; // est  [TP][Np]
    float *slabs;    // [nchunks][nsrc][L][K32][Np]
    int Np, K32, KB, PADL, L, Tl;
    int chunk_len;   // multiple of 6*LP
    int G;           // lag groups of 2*LP lags (fastest-varying part of blockIdx.x, so the groups
                     // that re-read the same X rows are dispatched together)
    int nsrc;        // 2: X0 and X1 (numW, denomW);  1: X0 only (HALS Gram of H_unfold)
    int CG;          // time chunks per workgroup (1, 2 or 4): the workgroup's waves are CG chunks x 4/CG adjacent n blocks and
                     // the CG partial sums of a block are added through LDS before the store (slabs: [ceil(nchunks/CG)]...)
};

// E[u] = R(2u), O[u] = R(2u-1), B[u] = X rows (2u, 2u+1) of the group that starts `row` rows after the
// bases of the two buffer descriptors (hoff/xoff: per-lane byte offsets; rows as scalar offsets)
template <int LP>
__device__ __forceinline__ void hxt_load(float (&E)[LP], float (&O)[LP], float (&B)[LP], __amdgpu_buffer_rsrc_t hr,
                                         __amdgpu_buffer_rsrc_t xr, int hoff, int xoff, int row, int K32, int Np)
{
#pragma unroll
    for (int u = 0; u < LP; ++u) {
        E[u] = cmf_bload(hr, hoff, (row + 2 * u + 1) * K32 * 4);
        O[u] = cmf_bload(hr, hoff, (row + 2 * u) * K32 * 4);
        B[u] = cmf_bload(xr, xoff, (row + 2 * u) * Np * 4);
    }
}

// the 2*LP*LP MFMAs of one group: step u, lag pair m uses E/O of step u-m (previous group's ring when u < m)
template <int LP>
__device__ __forceinline__ void hxt_group(f32x16 (&acc)[2 * LP], const float (&Ep)[LP], const float (&Op)[LP],
                                          const float (&Ec)[LP], const float (&Oc)[LP], const float (&Bc)[LP])
{
#pragma unroll
    for (int u = 0; u < LP; ++u) {
#pragma unroll
        for (int m = 0; m < LP; ++m) {
            const float ae = (u - m >= 0) ? Ec[(u - m >= 0) ? (u - m) : 0] : Ep[(u - m >= 0) ? 0 : (u - m + LP)];
            const float ao = (u - m >= 0) ? Oc[(u - m >= 0) ? (u - m) : 0] : Op[(u - m >= 0) ? 0 : (u - m + LP)];
            acc[2 * m] = __builtin_amdgcn_mfma_f32_32x32x2f32(ae, Bc[u], acc[2 * m], 0, 0, 0);
            acc[2 * m + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ao, Bc[u], acc[2 * m + 1], 0, 0, 0);
        }
    }
}

template <int LP>
__global__ __launch_bounds__(256, 1) void hxt_kernel(HxtParams p)
{
    // wave index as a scalar: the time chunk (and with it the buffer descriptors) depends on it, and a descriptor the
    // compiler believes to be divergent turns every buffer load into a waterfall loop
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
    // blockIdx.x -> (n block quad, lag group).  Placement (speed only): workgroup b runs on XCD b % 8, so when the
    // quads come in multiples of 8 the lag groups of one quad are put 8 apart -- the same XCD re-reads their common
    // X rows from its L2 instead of every lag group fetching them from HBM.
    int quad, lg;
    if ((gridDim.x / p.G) % 8 == 0) {
        lg = (blockIdx.x >> 3) % p.G;
        quad = (blockIdx.x & 7) + 8 * (blockIdx.x / (8 * p.G));
    } else {
        quad = blockIdx.x / p.G;
        lg = blockIdx.x % p.G;
    }
    const int CG = p.CG;
    const int cw = wave % CG, nw = wave / CG; // wave = nw * CG + cw
    const int nb = quad * (4 / CG) + nw;
    const int c = blockIdx.y * CG + cw;
    const int src = blockIdx.z % p.nsrc;
    const int kb = blockIdx.z / p.nsrc;
    const int lag0 = lg * 2 * LP;
    const int Np = p.Np, K32 = p.K32;
    const float *X = src ? p.X1 : p.X0;

    const int tc0 = c * p.chunk_len;
    int tc1 = tc0 + p.chunk_len;
    if (tc1 > p.Tl) tc1 = p.Tl;
    const int ngroups = (tc1 > tc0) ? (tc1 - tc0 + 6 * LP - 1) / (6 * LP) * 3 : 0;

    // Descriptor bases (wave-uniform).  H: row (PADL + tc0 - lag0 - 2*LP - 1), so that R(q) of the text
    // is row offset q + 2*LP + 1 >= 0;  X: row (PADL + tc0 - 2*LP) for the same group numbering.
    const float *hbase = p.H + (size_t)(p.PADL + tc0 - lag0 - 2 * LP - 1) * K32;
    const float *xbase = X + (size_t)(p.PADL + tc0 - 2 * LP) * Np;
    const __amdgpu_buffer_rsrc_t hr = cmf_rsrc(hbase, (size_t)(p.chunk_len + 12 * LP + 8) * K32 * 4);
    const __amdgpu_buffer_rsrc_t xr = cmf_rsrc(xbase, (size_t)(p.chunk_len + 12 * LP + 8) * Np * 4);
    const int hoff = (h * K32 + kb * 32 + i) * 4;
    const int xoff = (h * Np + nb * 32 + i) * 4;

    f32x16 acc[2 * LP];
#pragma unroll
    for (int a = 0; a < 2 * LP; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    // Three register sets rotate through the roles (previous group's ring, current group, prefetch
    // of the next group), so no register copies exist and the loads issued at the top of one group
    // are first needed a whole group (2*LP*LP MFMAs) later.
    float E0[LP], O0[LP], B0[LP], E1[LP], O1[LP], B1[LP], E2[LP], O2[LP], B2[LP];
    if (ngroups > 0) {
        hxt_load<LP>(E0, O0, B0, hr, xr, hoff, xoff, 0, K32, Np);      // ring of group -1 (its B is unused)
        hxt_load<LP>(E1, O1, B1, hr, xr, hoff, xoff, 2 * LP, K32, Np); // group 0
        // ngroups is a multiple of 3 (chunk_len is a multiple of 6*LP; a ragged last chunk is rounded
        // up and reads zero rows of X), so the body is straight-line code: no exits inside a rotation.
        // The 3*LP loads of a group go one per two MFMAs beside its first 6*LP MFMAs (they are needed a group later, so they must not
        // trail to the end of this one): the two waves of a SIMD run in lockstep, and a burst of loads at the top of the
        // group leaves the MFMA pipe idle in both.
#define HXT_SCHED()                                                                   \
    do {                                                                              \
        constexpr int per = (2 * LP * LP >= 9 * LP) ? 2 : ((2 * LP * LP >= 6 * LP) ? 1 : 0); /* the loads stay in the first part of the group */ \
        _Pragma("unroll") for (int q = 0; q < 3 * LP; ++q) {                          \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                        \
            __builtin_amdgcn_sched_group_barrier(0x008, per, 0);                      \
        }                                                                             \
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * LP * LP - per * 3 * LP, 0);   \
        __builtin_amdgcn_sched_barrier(0);                                            \
    } while (0)
        for (int g = 0; g < ngroups; g += 3) {
            const int g3 = (g + 3 < ngroups) ? g + 3 : g;
            hxt_load<LP>(E2, O2, B2, hr, xr, hoff, xoff, 2 * LP * (g + 2), K32, Np);
            hxt_group<LP>(acc, E0, O0, E1, O1, B1);
            HXT_SCHED();
            hxt_load<LP>(E0, O0, B0, hr, xr, hoff, xoff, 2 * LP * (g + 3), K32, Np);
            hxt_group<LP>(acc, E1, O1, E2, O2, B2);
            HXT_SCHED();
            hxt_load<LP>(E1, O1, B1, hr, xr, hoff, xoff, 2 * LP * (g3 + 1), K32, Np);
            hxt_group<LP>(acc, E2, O2, E0, O0, B0);
            HXT_SCHED();
        }
#undef HXT_SCHED
    }

    // store: acc[a][r] -> lag lag0+a, k = kb*32 + crow(r,h), n = nb*32 + i, into the slab of this chunk group
    float *slab = p.slabs + (size_t)(blockIdx.y * p.nsrc + src) * p.L * K32 * Np;
    if (CG == 1) {
#pragma unroll
        for (int a = 0; a < 2 * LP; ++a) {
            int l = lag0 + a;
            if (l < p.L) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int k = kb * 32 + cmf_crow(r, h);
                    slab[((size_t)l * K32 + k) * Np + nb * 32 + i] = acc[a][r];
                }
            }
        }
        return;
    }
    // The CG waves that hold the same (n block, k block, lag group) for CG consecutive time chunks add their partial
    // sums through LDS in a fixed order (chunk 0, 1, ...: deterministic) and each stores 16/CG of the registers:
    // CG times less slab traffic out of this kernel and into the slab sum.  One accumulator block per pass, two LDS
    // buffers, one barrier per pass.
    __shared__ float red[2][4][16][64];
#pragma unroll
    for (int a = 0; a < 2 * LP; ++a) {
        const int buf = a & 1;
#pragma unroll
        for (int r = 0; r < 16; ++r) red[buf][wave][r][lane] = acc[a][r];
        __syncthreads();
        const int l = lag0 + a;
        const int per = 16 / CG;
        for (int rr = 0; rr < per; ++rr) {
            const int r = cw * per + rr;
            float sum = red[buf][nw * CG][r][lane];
            for (int v = 1; v < CG; ++v) sum += red[buf][nw * CG + v][r][lane];
            const int k = kb * 32 + cmf_crow(r, h);
            if (l < p.L) slab[((size_t)l * K32 + k) * Np + nb * 32 + i] = sum;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// C3: out[t][k] = sum_{l,n} XT[n][t+l] * Wn[l][n][k]   (numH with X=data, denomH with X=est).
// A *pair* is 128 consecutive t (4 MFMA blocks) x one 32-wide k block x one source; its work is
// C = roundup(N,8)/8 chunks of 8 rows (n).  The U = pairs*C chunk units are dealt out evenly and
// statically to the W resident waves (wave w owns units [w*U/W, (w+1)*U/W)), so every wave does
// the same amount of MFMA work whatever the shape, in a single round, and the result does not
// depend on timing.  A wave's range covers one or two pairs; for each it accumulates its chunk
// range and writes the partial sum to the slab of its *fragment* index (its position among the
// waves that share the pair); the wave holding a pair's last chunk zero-fills the fragment slabs
// the pair does not use, so consumers simply sum all F slabs.
// Per pair segment the wave stages 8 rows (n) x 160 columns (t, incl. the right lag halo) of XT
// in its own LDS region (wave-private: no workgroup barrier anywhere), reads the 20 lag-shifted
// windows from LDS as the A operand, and streams the W operand (128-byte rows of Wn) straight
// from L2 into registers one n pair ahead.
// ---------------------------------------------------------------------------------------------
struct TcParams {
    const float *Wn;  // [Lp][Np][K32]
    const float *XT0; // dataT [Np][TP]
    const float *XT1; // estT  [Np][TP]
    float *slabs;     // [F][nsrc][Tl][K32]
    int NpW;          // rows per lag in Wn (= Np)
    int TP, PADL, K32, KB, L, Tl;
    int nsrc;         // 1: only XT0 (stand-alone transconv), 2: both
    int C;            // chunks (8 rows of n) per pair = roundup(N, 8) / 8: XT rows beyond are all zero
    int t_first;      // first column of pair block 0: 0, or -128 on a shard that also updates the L-1 columns in front of its own (the halo
                      // then travels in the W-phase all-reduce: cmf_groups.hip); slab row r holds column t_first + r
    int slab_rows;    // rows per slab (Tl - t_first)
    int W;            // waves that share the work (grid = ceil(W/4) workgroups)
    int F;            // fragment slabs
    const int4 *wtab; // [W] per wave: {first pair, first chunk in it, number of chunk units, fragment index of the first segment}
};

#define TC_ROW 160
#define TC_CHUNK (8 * TC_ROW)

template <int LT>
__device__ __forceinline__ void tc_load_w(float (&b)[LT], __amdgpu_buffer_rsrc_t wr, int woff, int row, int K32, int lagbytes)
{
#pragma unroll
    for (int l = 0; l < LT; ++l) b[l] = cmf_bload(wr, woff, row * K32 * 4 + l * lagbytes);
}

// one n pair: LT lags x 4 t blocks; the LDS reads of lag l+1 are issued ahead of the MFMAs of lag l
template <int LT>
__device__ __forceinline__ void tc_pair(f32x16 (&acc)[4], const float *sa, const float (&b)[LT])
{
    float a0 = sa[0], a1 = sa[32], a2 = sa[64], a3 = sa[96];
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
    for (int l = 0; l < LT; ++l) {
        float n0 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f;
        if (l + 1 < LT) {
            n0 = sa[l + 1]; n1 = sa[l + 33]; n2 = sa[l + 65]; n3 = sa[l + 97];
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[l], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b[l], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b[l], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b[l], acc[3], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        a0 = n0; a1 = n1; a2 = n2; a3 = n3;
    }
}

// global -> LDS copy of 16 bytes per lane without a register stop-over: LDS destination = lds_base + lane * 16
// (wave-uniform base in M0), global source = a per-lane 64-bit address.  Written as inline assembly so that the
// compiler does not see an LDS-DMA in flight: it would otherwise drain vmcnt to 0 at the next use of any ordinary
// load (cdna_hip_programming.md, "Pipelining across barriers"); completion is awaited by an explicit counted
// s_waitcnt vmcnt(N) -- an untracked older operation can only make the compiler's own counted waits stricter.
__device__ __forceinline__ void cmf_glds16(const void *gsrc, unsigned lds_base)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_base)
                 : "memory");
}

// the wave whose range holds chunk unit u (ranges: wave w owns [w*U/W, (w+1)*U/W))
__host__ __device__ __forceinline__ long long tc_wave_of(long long u, long long U, long long W)
{
    return ((u + 1) * W - 1) / U;
}

template <int LT>
__global__ __launch_bounds__(256, 2) void transconv_kernel(TcParams p)
{
    __shared__ __attribute__((aligned(16))) float smem[4 * 2 * TC_CHUNK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    float *S = smem + wave * (2 * TC_CHUNK);
    const int Np = p.NpW, TP = p.TP, K32 = p.K32;
    const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
    if (gw >= p.W) return;
    const int4 wt = p.wtab[gw];
    int pr = wt.x, c0 = wt.y, left = wt.z, frag = wt.w;
    const int LB = (p.L + 31) >> 5;
    const size_t lagstride = (size_t)Np * K32;
    const int lagbytes = (int)(lagstride * 4);
    // LDS byte address of this wave's region (wave-uniform)
    const unsigned lds_S = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float *)S);
    unsigned xoff[5]; // byte offset of this lane's float4 of an 8 x 160 chunk: element idx = lane + 64 q -> (row, column)
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        int idx = lane + 64 * q, row = idx / 40, c4 = idx - row * 40;
        xoff[q] = (unsigned)(row * TP + c4 * 4) * 4u;
    }

    while (left > 0) {
        // pair index -> (t block, k block, source); the source is the fastest index
        const int c1 = (left < p.C - c0) ? c0 + left : p.C;
        const int src = pr % p.nsrc;
        const int kb = (pr / p.nsrc) % p.KB;
        const int t0 = p.t_first + (pr / (p.nsrc * p.KB)) * 128;
        const float *XT = src ? p.XT1 : p.XT0;
        const int nlo = 8 * c0;
        const int nchunks = c1 - c0;

        f32x16 acc[4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

        for (int lb = 0; lb < LB; ++lb) {
            const float *xsrc = XT + (size_t)nlo * TP + (p.PADL + t0 + 32 * lb);
            // chunk 0 straight into LDS buffer 0 (global_load_lds: wave-uniform LDS base + lane * 16 bytes, which is
            // exactly the strip's layout: element idx = lane + 64 q of the 8 x 160 chunk)
            __builtin_amdgcn_wave_barrier(); // lag block > 0: the previous block's LDS reads are done
#pragma unroll
            for (int q = 0; q < 5; ++q) cmf_glds16(reinterpret_cast<const char *>(xsrc) + xoff[q], lds_S + q * 1024);
            // W operand for n pair `it`: Wn[lb*32 + l][nlo + 2*it + h][kb*32 + i]  (descriptor base: lag lb*32)
            const __amdgpu_buffer_rsrc_t wr = cmf_rsrc(p.Wn + (size_t)(lb * 32) * lagstride, (size_t)LT * lagstride * 4);
            const int woff = (h * K32 + kb * 32 + i) * 4;
            float b0[LT], b1[LT]; // ping-pong: no register copies, so a load is first needed one pair later
            tc_load_w<LT>(b0, wr, woff, nlo, K32, lagbytes);
            int buf = 0;
            for (int c = 0; c < nchunks; ++c) {
                float *Sb = S + buf * TC_CHUNK;
                // chunk c was issued one iteration ago; only the W prefetch (LT loads) is younger
                __builtin_amdgcn_s_waitcnt(0x0F70 | (LT & 15) | ((LT >> 4) << 14)); // vmcnt(LT), expcnt/lgkmcnt untouched
                __builtin_amdgcn_wave_barrier();
                if (c + 1 < nchunks) {
                    const float *xs2 = xsrc + (size_t)(8 * (c + 1)) * TP;
                    const unsigned Sn = lds_S + (buf ^ 1) * (TC_CHUNK * 4);
#pragma unroll
                    for (int q = 0; q < 5; ++q) cmf_glds16(reinterpret_cast<const char *>(xs2) + xoff[q], Sn + q * 1024);
                }
                const int nrow = nlo + 8 * c;
                const float *sa = Sb + h * TC_ROW + i;
                tc_load_w<LT>(b1, wr, woff, nrow + 2, K32, lagbytes);
                __builtin_amdgcn_sched_barrier(0);
                tc_pair<LT>(acc, sa, b0);
                tc_load_w<LT>(b0, wr, woff, nrow + 4, K32, lagbytes);
                __builtin_amdgcn_sched_barrier(0);
                tc_pair<LT>(acc, sa + 2 * TC_ROW, b1);
                tc_load_w<LT>(b1, wr, woff, nrow + 6, K32, lagbytes);
                __builtin_amdgcn_sched_barrier(0);
                tc_pair<LT>(acc, sa + 4 * TC_ROW, b0);
                tc_load_w<LT>(b0, wr, woff, nrow + ((c + 1 < nchunks) ? 8 : 0), K32, lagbytes);
                __builtin_amdgcn_sched_barrier(0);
                tc_pair<LT>(acc, sa + 6 * TC_ROW, b1);
                buf ^= 1;
            }
        }
        // fragment index = position of this wave among the waves sharing the pair (0 for a pair it starts)
        const size_t slabstride = (size_t)p.nsrc * p.slab_rows * K32;
        float *slab = p.slabs + (((size_t)frag * p.nsrc + src) * p.slab_rows - p.t_first) * K32; // (row t - t_first)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int t = t0 + tb * 32 + cmf_crow(r, h);
                if (t < p.Tl) slab[(size_t)t * K32 + kb * 32 + i] = acc[tb][r];
            }
        if (c1 == p.C) { // the pair is complete: clear the fragment slabs it does not use
            for (int f = frag + 1; f < p.F; ++f) {
                float *z = slab + (size_t)(f - frag) * slabstride;
#pragma unroll
                for (int tb = 0; tb < 4; ++tb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int t = t0 + tb * 32 + cmf_crow(r, h);
                        if (t < p.Tl) z[(size_t)t * K32 + kb * 32 + i] = 0.f;
                    }
            }
        }
        left -= nchunks;
        ++pr;
        c0 = 0;
        frag = 0;
    }
}

// ---------------------------------------------------------------------------------------------
// Element-wise MU updates (mult.jl:37-38 and :51-52) with the slab sums folded in.
//   x <- max(eps, x * (num / (((den + l1) + (2*l2)*x) + eps)))      padding stays exactly 0
// Each also refreshes the transposed copy of its factor through an LDS tile.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float cmf_mu(float x, float num, float den, float l1, float two_l2)
{
    float d = ((den + l1) + two_l2 * x) + CMF_EPS_F;
    float y = x * (num / d);
    return (y != y) ? y : fmaxf(CMF_EPS_F, y);
}

// grid: (Np/64, KB, L), block 256.  num, den: [L][K32][Np] each
// tail_src / tail_dst (may be NULL): block (0,0,0) also copies `tail_n` (<= 256) floats -- the loss pairs behind the
// [numW | denomW] all-reduce buffer of a sharded group -- to pinned host memory that the host has filled with a
// sentinel pattern and polls, so the read-back costs neither a launch nor an event (see loss_reduce_kernel).
static __global__ __launch_bounds__(256) void w_update_kernel(float *Wt, float *Wn, const float *num_p, const float *den_p,
                                                        int N, int K, int L, int Np, int K32, float l1, float two_l2,
                                                        const float *tail_src, float *tail_dst, int tail_n)
{
    __shared__ float tile[32][65];
    const int tid = threadIdx.x;
    if (tail_dst && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid < tail_n) // relaxed system-scope word stores:
        __hip_atomic_store(tail_dst + tid, tail_src[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // the host polls every word
    const int n0 = blockIdx.x * 64, kb = blockIdx.y, l = blockIdx.z;
    {
        const int nn = tid & 63;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            int kk = q * 4 + (tid >> 6);
            int k = kb * 32 + kk, n = n0 + nn;
            size_t idx = ((size_t)l * K32 + k) * Np + n;
            const float num = num_p[idx], den = den_p[idx];
            float w = Wt[idx];
            float wn = (k < K && n < N) ? cmf_mu(w, num, den, l1, two_l2) : 0.f;
            Wt[idx] = wn;
            tile[kk][nn] = wn;
        }
    }
    __syncthreads();
    {
        const int kk = tid & 31;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            int nn = q * 8 + (tid >> 5);
            Wn[((size_t)l * Np + n0 + nn) * K32 + kb * 32 + kk] = tile[kk][nn];
        }
    }
}

// grid: (ceil(Tl/8), KB), block 256.  slabs: [S][2][Tl][K32]
// A workgroup owns 8 columns x 32 components = 64 float4 elements; its four 64-thread groups each sum every
// fourth slab (the slab count grows as the shard gets shorter: 36 at T = 6250), the groups are combined in a
// fixed order through LDS, so the result does not depend on timing.
#define HUPD_T 8
// num / den: Snum / Sden partial-sum slabs of [Tl][K32] floats, `num_stride` / `den_stride` floats apart (the two-source
// transconv writes [S][2][Tl][K32]: num = slabs, den = slabs + Tl*K32, both strides 2*Tl*K32; the Gram form has the S1
// slabs of the one-source launch for num and ONE array for den).
static __global__ __launch_bounds__(256) void h_update_kernel(float *H, float *Ht, const float *nump, size_t num_stride, int Snum,
                                                        const float *denp, size_t den_stride, int Sden,
                                                        int Tl, int K, int K32, int PADL, int TP, float l1, float two_l2)
{
    __shared__ f32x4 red[3][64][2];
    __shared__ float tile[32][HUPD_T + 1];
    const int tid = threadIdx.x;
    const int e = tid & 63, g = tid >> 6;
    const int tt = e >> 3, k4 = e & 7;
    const int t0 = blockIdx.x * HUPD_T, kb = blockIdx.y;
    const int t = t0 + tt, k = kb * 32 + 4 * k4;
    const f32x4 *sn = reinterpret_cast<const f32x4 *>(nump);
    const f32x4 *sd = reinterpret_cast<const f32x4 *>(denp);
    const size_t ns4 = num_stride / 4, ds4 = den_stride / 4;
    f32x4 num = {0.f, 0.f, 0.f, 0.f}, den = {0.f, 0.f, 0.f, 0.f};
    f32x4 *hp = reinterpret_cast<f32x4 *>(H + (size_t)(PADL + (t < Tl ? t : 0)) * K32 + (k < K ? k : 0));
    f32x4 x = {0.f, 0.f, 0.f, 0.f};
    if (g == 0 && t < Tl && k < K) x = *hp; // (requested with the slabs, not behind their sum)
    if (t < Tl && k < K) { // (a group of four components that lies wholly in the padding of the k block reads nothing: K = 5 of 32)
        const size_t idx = ((size_t)t * K32 + k) / 4;
        // (up to four slabs of either sum in flight per thread, added in slab order: one at a time the loop is a chain of round trips)
        for (int s0 = g; s0 < Snum || s0 < Sden; s0 += 16) {
            f32x4 vn[4], vd[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int sl = s0 + 4 * u;
                vn[u] = sn[(size_t)(sl < Snum ? sl : 0) * ns4 + idx];
                vd[u] = sd[(size_t)(sl < Sden ? sl : 0) * ds4 + idx];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (s0 + 4 * u < Snum) num += vn[u];
                if (s0 + 4 * u < Sden) den += vd[u];
            }
        }
    }
    if (g > 0) {
        red[g - 1][e][0] = num;
        red[g - 1][e][1] = den;
    }
    __syncthreads();
    if (g == 0) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            num += red[q][e][0];
            den += red[q][e][1];
        }
        f32x4 hn = {0.f, 0.f, 0.f, 0.f};
        if (t < Tl && k < K) { // (the padding stays the zero it is)
#pragma unroll
            for (int c = 0; c < 4; ++c) hn[c] = (k + c < K) ? cmf_mu(x[c], num[c], den[c], l1, two_l2) : 0.f;
            *hp = hn;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) tile[4 * k4 + c][tt] = hn[c];
    }
    __syncthreads();
    {
        const int kk = tid >> 3, t2 = t0 + (tid & 7);
        if (t2 < Tl) Ht[(size_t)(kb * 32 + kk) * TP + PADL + t2] = tile[kk][tid & 7];
    }
}

// A thread's share of the sum of n per-tile loss sums (256 threads; element e goes to thread e % 256, accumulator (e / 256) % 8):
// four passes of eight loads in flight together, added pass after pass -- bit for bit the sums of the one-pass loop this was (6000
// partials were three dependent round trips on the critical path of a 5 us launch; config 2's 25 000 twelve, 11.6 us between the
// loss conv and the host).  Every loss reduction of the library sums in this order.
__device__ __forceinline__ double cmf_thread_loss_sum(const double *partial, int n)
{
    double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int base = threadIdx.x; base < n; base += 4 * 8 * 256) {
        double v[4][8];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = base + 2048 * q + 256 * u;
                v[q][u] = partial[e < n ? e : 0];
            }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += (base + 2048 * q + 256 * u < n) ? v[q][u] : 0.0;
    }
    return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

// A loss reduction that rides on another launch (cmf_iterate): the per-tile partials of the loss conv are summed by
// one block of the NEXT iteration's slab sum instead of by a 4 us launch of their own right behind the conv -- a
// launch that short in front of the C2 kernel exposes that kernel's dispatch set-up (measured: a 5.6 us hole in an
// 0.75 ms iteration).  partial == NULL: nothing to carry.
struct CmfLossCarry {
    const double *partial; // per-tile sums of (est - data)^2
    int n;
    double *out;           // device scalar
    double *host_out;      // pinned slot the host polls (single GPU) or NULL
    float *tail;           // tail of the [numW | denomW] all-reduce buffer (groups) or NULL: (hi, lo) in this rank's slots, 0 elsewhere
    int tail_len, rank;
};

__device__ __forceinline__ void cmf_block_loss_reduce(const CmfLossCarry &c)
{
    __shared__ double red[256];
    red[threadIdx.x] = cmf_thread_loss_sum(c.partial, c.n);
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    const double tot = red[0];
    if (threadIdx.x == 0) {
        *c.out = tot;
        if (c.host_out) __hip_atomic_store(c.host_out, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (c.tail) {
        const float hi = (float)tot, lo = (float)(tot - (double)hi);
        for (int j = threadIdx.x; j < c.tail_len; j += 256) c.tail[j] = (j == 2 * c.rank) ? hi : (j == 2 * c.rank + 1) ? lo : 0.f;
    }
}

// The last rows of a C2 contraction, added by the slab sum: the C2 kernel works in whole rotations of its register ring
// (6*LP time rows), and leaving the < 6*LP rows behind the last whole rotation to this kernel lets the chunks divide the
// rest exactly (T = 6250 over 16 chunks: 390 rows each + 10 here, instead of 420 each with 30 of them padding).
//   out[src][l][k][n] += sum_{t = t0}^{t0 + rows - 1} H[t - l][k] * X_src[t][n]       (t = padded row index)
struct CmfHxtTail {
    const float *H, *X0, *X1; // rows == 0: nothing to add
    int rows, t0, L, K32, Np;
};

// out[i] = sum_s in[s*stride + i]  (deterministic slab combine; float4 lanes); the last block also performs a carried
// loss reduction (same summation order as loss_reduce_kernel / loss_tail_kernel)
static __global__ __launch_bounds__(256) void slab_sum_kernel(float *out, const float *in, int nslabs, size_t stride, size_t n4, CmfLossCarry carry,
                                                        CmfHxtTail tail)
{
    // (a carried loss reduction has a block of its own, the FIRST one: behind the last block's share of the sums it was the end of the launch)
    unsigned bx = blockIdx.x, nbx = gridDim.x;
    if (carry.partial) {
        if (bx == 0) { cmf_block_loss_reduce(carry); return; }
        --bx; --nbx;
    }
    for (size_t idx = bx * (size_t)blockDim.x + threadIdx.x; idx < n4; idx += (size_t)nbx * blockDim.x) {
        float4 a = reinterpret_cast<const float4 *>(in)[idx];
        for (int s = 1; s < nslabs; s += 4) { // four loads in flight, added in slab order (one at a time the loop is a chain of HBM round trips)
            float4 b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int su = (s + u < nslabs) ? s + u : s; // (a slab that exists; its value is dropped below)
                b[u] = reinterpret_cast<const float4 *>(in + (size_t)su * stride)[idx];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (s + u < nslabs) { a.x += b[u].x; a.y += b[u].y; a.z += b[u].z; a.w += b[u].w; }
        }
        if (tail.rows > 0) { // element 4 * idx = ((src * L + l) * K32 + k) * Np + n
            const size_t e = 4 * idx;
            const int n = (int)(e % tail.Np);
            const int k = (int)((e / tail.Np) % tail.K32);
            const int l = (int)((e / ((size_t)tail.Np * tail.K32)) % tail.L);
            const int src = (int)(e / ((size_t)tail.Np * tail.K32 * tail.L));
            const float *X = src ? tail.X1 : tail.X0;
            for (int r0 = 0; r0 < tail.rows; r0 += 6) { // six row pairs in flight (one at a time: a chain of round trips)
                float hv[6];
                float4 xv[6];
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const int t = tail.t0 + ((r0 + u < tail.rows) ? r0 + u : r0);
                    hv[u] = tail.H[(size_t)(t - l) * tail.K32 + k];
                    xv[u] = *reinterpret_cast<const float4 *>(X + (size_t)t * tail.Np + n);
                }
#pragma unroll
                for (int u = 0; u < 6; ++u)
                    if (r0 + u < tail.rows) {
                        a.x = fmaf(hv[u], xv[u].x, a.x); a.y = fmaf(hv[u], xv[u].y, a.y);
                        a.z = fmaf(hv[u], xv[u].z, a.z); a.w = fmaf(hv[u], xv[u].w, a.w);
                    }
            }
        }
        reinterpret_cast<float4 *>(out)[idx] = a;
    }
}

// *out = sum(partial[0..n))   one block; eight independent loads per thread and round (the one-wave conv kernel
// writes 25 024 partials at config 2: a plain strided loop is a chain of dependent HBM round trips)
// host_out (may be NULL): pinned, coherent host memory that receives the sum as well (the pipelined loss read-back of
// cmf_iterate).  It is ONE relaxed system-scope 8-byte store: the host has filled the slot with a sentinel bit pattern
// and polls until it changes.  No event and no system-scope fence: either would write the whole dirty L2 back (the est
// the conv just stored) and idle the device for ~6 us between this kernel and the next.
#define CMF_SENTINEL64 0xFFFFFFFFFFFFFFFFull
#define CMF_SENTINEL32 0xFFFFFFFFu
static __global__ __launch_bounds__(256) void loss_reduce_kernel(const double *partial, int n, double *out, double *host_out = nullptr)
{
    __shared__ double red[256];
    red[threadIdx.x] = cmf_thread_loss_sum(partial, n);
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *out = red[0];
        if (host_out) __hip_atomic_store(host_out, red[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}


// =============================================================================================
// HALS rule (BASELINE config 5; src/algs/hals.jl).  The reference performs K*L rank-1 residual
// sweeps for W and K*T dependent window updates for H on the N x T residual.  Here the same
// Gauss-Seidel recurrences (same visiting order: k outer / lag inner for W, k outer / t inner for H)
// run on Gram-projected state instead, which is algebraically identical:
//   W:  G[n][j] = <resid[n,:], h_j> (= denomW - numW of the MU kernels),  HH = H_unfold H_unfold'
//       step j=(k,l): v = G[:,j] - w_old*HH[j][j];  w_new = max((-v - l1)/(HH[j][j]+eps+l2), 0);
//                     G += (w_new - w_old) (x) HH[j][:]                       (hals.jl:100-112)
//   H:  P[k][t] = <W_k window, resid[:, t:t+L-1]> (= denomH - numH),  GW[k][k'][e] = sum_l <W[k,:,l], W[k',:,l-e]>
//       step (k,t): x = max((h_old*nrm - P[k][t] - l1)/(nrm+eps+l2), 0);  P[k'][t+e] += (x - h_old) GW[k][k'][e]
//       with the window truncated at the right edge exactly like hals.jl:136-146.
// =============================================================================================

// ---- HH = H_unfold * H_unfold' (hals.jl:56-60) from lag correlations ---------------------------------------------
// HH[(l,k)][(l',k')] = sum_{t >= max(l,l')}^{T-1} H[k][t-l] * H[k'][t-l'] depends on (k, k', l - l') only, up to the
// few terms at the right end that the shift cuts off:
//   l >= l', d = l - l':  HH = C[d][k][k'] - sum_{u = T-l'}^{T-1} H[k][u-d] * H[k'][u]
//   l <  l', d = l' - l:  HH = C[d][k'][k] - sum_{u = T-l}^{T-1}  H[k'][u-d] * H[k][u]
// with C[d][a][b] = sum_t H[a][t-d] * H[b][t], which is the C2 contraction (hxt_kernel) of H with ITSELF as the X
// operand: K32 columns (padded to 128) instead of the L*K32 columns of a materialised H_unfold' -- a fifth of the MFMA
// work of round 1's form at K = 32, L = 20, and no 128 MB H_unfold' to build.
// The lag correlations themselves (round 6; until round 5 the general C2 kernel ran with H packed as its own X operand: K32 columns in a
// 128-column pitch, 2048 waves of a few hundred MFMAs each and 64 slabs behind them -- 72 + 12 us for 2 GFLOP at config 5):
//   C[d][a][b] = sum_{t in [0, Tl)} H[t - d][a] * H[t][b]     d < L;  a, b < K32   (H[t - d] for t < d: the left halo / the zero padding)
// grid (G, KB * KB), 256 threads: workgroup g owns the time rows [g R, (g+1) R), its 4 waves own the lags d = wave, wave + 4, ... --
// five accumulator blocks per pass, ceil(L / 20) passes -- and both MFMA operands are rows of H read in operand layout (A: lane i of
// half h = H[t0 + h - d][a0 + i], B: H[t0 + h][b0 + i]), one step of two time rows ahead.  Partial sums to slab g ([L][K32][K32],
// compact); hals_corr_sum_kernel adds the slabs in order (deterministic) into C's [L][K32][NpC] layout.
#define HALS_CORR_LPW 5
static __global__ __launch_bounds__(256) void hals_corr_kernel(const float *H, float *slabs, int K32, int KB, int PADL, int Tl, int L, int R)
{
    extern __shared__ float hc_rows[]; // [R + L - 1 (+ 4: the read-ahead)][K32]: rows t_begin - (L-1) .. t_end - 1 of H, zeros behind t_end
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int ab = blockIdx.y / KB, bb = blockIdx.y % KB;
    const int t_begin = blockIdx.x * R, t_end = (t_begin + R < Tl) ? t_begin + R : Tl;
    float *slab = slabs + (size_t)blockIdx.x * L * K32 * K32;
    {   // stage the rows once (the operands were read from L2 row by row in the first form of this kernel: one wave per SIMD cannot
        // hide that latency -- 80 us, no better than the C2 kernel it replaced)
        const f32x4 *src = reinterpret_cast<const f32x4 *>(H + (size_t)(PADL + t_begin - (L - 1)) * K32);
        f32x4 *dst = reinterpret_cast<f32x4 *>(hc_rows);
        const int nlive = (t_end - t_begin + L - 1) * K32 / 4, nall = (R + L - 1 + 4) * K32 / 4;
        for (int e = tid; e < nall; e += 256) dst[e] = e < nlive ? src[e] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    const float *Hs = hc_rows + (size_t)(L - 1) * K32; // row t - t_begin of the workgroup's own range
    const int nsteps = (t_end - t_begin + 3) / 4;       // steps of four rows (the rows behind t_end are zero)
    for (int d0 = wave; d0 < L; d0 += 4 * HALS_CORR_LPW) { // this pass: lags d0, d0 + 4, ..., d0 + 4 (LPW - 1)
        f32x16 acc[HALS_CORR_LPW];
#pragma unroll
        for (int q = 0; q < HALS_CORR_LPW; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
        // lane (i, half h): A = H[t + h - d][a0 + i], B = H[t + h][b0 + i]; lags beyond L read lag 0's rows and are not stored
        const float *pb = Hs + h * K32 + bb * 32 + i;
        const float *pa[HALS_CORR_LPW];
#pragma unroll
        for (int q = 0; q < HALS_CORR_LPW; ++q) pa[q] = Hs + (h - ((d0 + 4 * q < L) ? d0 + 4 * q : 0)) * K32 + ab * 32 + i;
        float a0[HALS_CORR_LPW], a1[HALS_CORR_LPW], b0 = pb[0], b1;
#pragma unroll
        for (int q = 0; q < HALS_CORR_LPW; ++q) a0[q] = pa[q][0];
        for (int s = 0; s < nsteps; ++s) {
            const int o = s * 4 * K32;
            b1 = pb[o + 2 * K32];
#pragma unroll
            for (int q = 0; q < HALS_CORR_LPW; ++q) a1[q] = pa[q][o + 2 * K32];
#pragma unroll
            for (int q = 0; q < HALS_CORR_LPW; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[q], b0, acc[q], 0, 0, 0);
            b0 = pb[o + 4 * K32]; // (the last step reads the four zero rows of the read-ahead)
#pragma unroll
            for (int q = 0; q < HALS_CORR_LPW; ++q) a0[q] = pa[q][o + 4 * K32];
#pragma unroll
            for (int q = 0; q < HALS_CORR_LPW; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[q], b1, acc[q], 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < HALS_CORR_LPW; ++q) {
            const int d = d0 + 4 * q;
            if (d < L)
#pragma unroll
                for (int r = 0; r < 16; ++r) slab[((size_t)d * K32 + ab * 32 + cmf_crow(r, h)) * K32 + bb * 32 + i] = acc[q][r];
        }
    }
}
// C[d][a][b] (pitch NpC) = sum over the G slabs.  A workgroup = 64 consecutive entries x 16 waves: wave j adds the slabs g = j, j + 16, ...
// (coalesced 256-byte reads, all in flight), the 16 partial sums meet in LDS and are added in wave order -- a fixed order, the same
// result on every run.  (One thread per entry walking all G slabs was 30 us of latency for 21 MB.)   grid: L * K32 * K32 / 64
static __global__ __launch_bounds__(1024) void hals_corr_sum_kernel(const float *slabs, float *C, int G, int L, int K32, int NpC)
{
    __shared__ float part[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t n = (size_t)L * K32 * K32, idx = (size_t)blockIdx.x * 64 + lane;
    float s = 0.f;
    if (idx < n)
        for (int g = wave; g < G; g += 16) s += slabs[(size_t)g * n + idx];
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && idx < n) {
        float tot = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) tot += part[j][lane];
        const size_t da = idx / K32, b = idx - da * K32;
        C[da * NpC + b] = tot;
    }
}

// HH[(l*K32+k) * NpH + l'*K32+k'] from C [L][K32][NpC] and the last columns of H ([TP][K32]).  One thread per entry.
// sharded != 0: this handle owns one column block of a T-sharded problem and C holds the lag correlations over ITS
// columns only (with the left H halo); the entry is then this shard's additive share of HH -- C as it is, minus the cut
// terms on the shard that holds the global right edge (is_last; the columns u - d >= Tl - 2(L-1) they read are the
// shard's own or its left halo) -- and the shares are summed by the group's all-reduce.
static __global__ void hals_hh_kernel(const float *C, const float *H, float *HH, int Tl, int L, int K, int K32, int NpC, int NpH, int PADL,
                               int sharded, int is_last)
{
    const int LK = L * K32;
    const size_t total = (size_t)LK * LK;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(idx / LK), jp = (int)(idx - (size_t)j * LK);
        const int l = j / K32, k = j - l * K32, lp = jp / K32, kp = jp - lp * K32;
        float v = 0.f;
        if (k < K && kp < K) {
            // a = the row with the larger lag, b the other; cut = the smaller lag: the terms u >= T - cut are not in HH
            const int d = l >= lp ? l - lp : lp - l;
            const int a = l >= lp ? k : kp, b = l >= lp ? kp : k, cut = l >= lp ? lp : l;
            // kept terms: u in [d, T - cut).  When they are few (T of the order of L) they are summed
            // directly: C minus nearly all of itself would leave rounding noise where H_unfold has exact zeros (rows with
            // l >= T), and the sweep divides by HH[j][j] + eps
            const int kept = Tl - cut - d;
            if (sharded) {
                v = C[((size_t)d * K32 + a) * NpC + b];
                if (is_last)
                    for (int u = Tl - cut; u < Tl; ++u) v -= H[(size_t)(PADL + u - d) * K32 + a] * H[(size_t)(PADL + u) * K32 + b];
            } else if (kept <= 0) {
                v = 0.f;
            } else if (kept <= cut || kept <= 64) {
                for (int u = d; u < Tl - cut; ++u) v = fmaf(H[(size_t)(PADL + u - d) * K32 + a], H[(size_t)(PADL + u) * K32 + b], v);
            } else {
                v = C[((size_t)d * K32 + a) * NpC + b];
                for (int u = Tl - cut; u < Tl; ++u) v -= H[(size_t)(PADL + u - d) * K32 + a] * H[(size_t)(PADL + u) * K32 + b]; // (u - d >= 0 here)
            }
        }
        HH[(size_t)j * NpH + jp] = v;
    }
}

#define HALS_NG 2
// W sweep: one wave = HALS_NG units n, the projected state g[u][j] (j over the L*K32 columns
// of H_unfold) sits in registers, lane (j % 64) slot (j / 64).  A step needs one state entry per unit -- a
// v_readlane of the slot its column lives in -- and then updates every entry with one FMA; there is no LDS, no
// wave synchronisation and nothing to wait for except the prefetched HH row, so a step costs a few hundred cycles
// instead of several LDS and memory round trips.
// grid: ceil(N / (4 * HALS_NG)), block 256 (4 independent waves); dynamic LDS: 4 * K*L * HALS_NG floats.
// NQ = state slots per lane (64 * NQ >= L * K32), HALS_WD = prefetch depth in steps: 8 up to 16 slots, 4 for the long
// states (K32 * L up to 2048, e.g. K = 64, L = 20), whose prefetch ring would otherwise not fit the register file.
template <int NQ, int HALS_WD>
__global__ __launch_bounds__(256) void hals_w_sweep_reg_kernel(float *Wt, float *Wn, const float *G, const float *Gsub, const float *HH,
                                                                int N, int K, int L, int Np, int K32, int NpH, float l1, float l2)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int LK = L * K32;
    const int n0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wave) * HALS_NG);
    if (n0 >= N) return;
    extern __shared__ float wnew_all[]; // [4][K*L * HALS_NG]: new W values of the wave's units, step by step
    float *wnew = wnew_all + (size_t)wave * (K * L * HALS_NG);
    float g[HALS_NG][NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int j = lane + 64 * q;
#pragma unroll
        for (int u = 0; u < HALS_NG; ++u) { // Gsub (may be NULL): the projection as a difference, G - Gsub = denomW - numW
            const size_t at = (size_t)j * Np + n0 + u;
            g[u][q] = (j < LK && n0 + u < N) ? (Gsub ? G[at] - Gsub[at] : G[at]) : 0.f;
        }
    }
    const int nsteps = K * L;
    // HH rows, diagonal entries and old W values are prefetched HALS_WD steps ahead into a register ring (a step is
    // ~200 cycles of dependent arithmetic, an L2 round trip several times that); the step loop is unrolled by the
    // ring depth so that every ring slot is a fixed set of registers.
    float hr[HALS_WD][NQ], hpp[HALS_WD], hinv[HALS_WD], wo[HALS_WD][HALS_NG];
    // all prefetches are buffer loads with wave-uniform (scalar) offsets: no address arithmetic in vector registers,
    // and the descriptor's bound makes the reads past a row's or the array's end harmless zeros without a branch
    const __amdgpu_buffer_rsrc_t hrs = cmf_rsrc(HH, (size_t)LK * NpH * 4);
    const __amdgpu_buffer_rsrc_t wrs = cmf_rsrc(Wt, (size_t)LK * Np * 4);
    // (Scalar bookkeeping -- the step index divided by L, the offsets multiplied out -- is about a third of a step's ~130
    // instructions, but both cheaper-looking forms were measured SLOWER than this one (0.28 ms): running counters with a wrap
    // test, 0.35 ms -- the wrap is a scalar branch in a loop whose wave is alone on its SIMD; and the constant parts of the load
    // offsets folded into the instructions' immediate field, 0.39 ms.)
    auto prefetch = [&](int sidx, float (&row)[NQ], float &diag, float (&wold)[HALS_NG]) {
        const int c = sidx < nsteps ? sidx : nsteps - 1; // clamped to the last step
        const int kk = c / L, ll = c - kk * L;
        const int pi = ll * K32 + kk;
#pragma unroll
        for (int q = 0; q < NQ; ++q) row[q] = cmf_bload(hrs, lane * 4, (pi * NpH + 64 * q) * 4);
        diag = cmf_bload(hrs, 0, (pi * NpH + pi) * 4);
#pragma unroll
        for (int u = 0; u < HALS_NG; ++u) wold[u] = cmf_bload(wrs, 0, (pi * Np + n0 + u) * 4); // same address in every lane
    };
#pragma unroll
    for (int i = 0; i < HALS_WD; ++i) prefetch(i, hr[i], hpp[i], wo[i]);
    hinv[0] = 1.0f / (hpp[0] + CMF_EPS_F + l2);
    for (int s0 = 0; s0 < nsteps; s0 += HALS_WD) {
#pragma unroll
        for (int i = 0; i < HALS_WD; ++i) {
            const int sidx = s0 + i;
            if (sidx < nsteps) { // wave-uniform
                const int k = sidx / L, l = sidx - k * L;
                const int pidx = l * K32 + k;
                const int tl = pidx & 63, tq = pidx >> 6; // wave-uniform: lane and slot of column pidx
                float d[HALS_NG];
#pragma unroll
                for (int u = 0; u < HALS_NG; ++u) {
                    float gs = g[u][0];
#pragma unroll
                    for (int q = 1; q < NQ; ++q) gs = (tq == q) ? g[u][q] : gs;
                    const float gv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gs), tl));
                    const float v = gv - wo[i][u] * hpp[i];                  // hals.jl:104 projected
                    float wn = (-v - l1) * hinv[i];                          // hals.jl:110 (the reciprocal of the norm is formed a step
                                                                             // ahead, below: no division in the chain)
                    wn = fmaxf(wn, 0.f);
                    if (lane == 0) wnew[sidx * HALS_NG + u] = wn; // (one lane: 64 lanes on one address serialise in the LDS) flushed after the sweep
                    d[u] = (n0 + u < N) ? wn - wo[i][u] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int u = 0; u < HALS_NG; ++u) g[u][q] = fmaf(d[u], hr[i][q], g[u][q]); // hals.jl:106
                // the next step's reciprocal norm: its diagonal entry arrived several steps ago, and nothing in this step's chain
                // waits for it
                hinv[(i + 1) % HALS_WD] = 1.0f / (hpp[(i + 1) % HALS_WD] + CMF_EPS_F + l2);
                prefetch(sidx + HALS_WD, hr[i], hpp[i], wo[i]);
            }
        }
    }
    // flush: no global store inside the sweep -- loads and stores share the in-order vmcnt on gfx9-class hardware
    // and the compiler must then wait for *everything* (vmcnt(0)) at each use of a prefetched value
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int idx = lane; idx < nsteps * HALS_NG; idx += 64) {
        const int sidx = idx / HALS_NG, u = idx - sidx * HALS_NG;
        const int k = sidx / L, l = sidx - k * L;
        if (n0 + u < N) {
            const float wn = wnew[idx];
            Wt[(size_t)(l * K32 + k) * Np + n0 + u] = wn;
            Wn[((size_t)l * Np + n0 + u) * K32 + k] = wn;
        }
    }
}

// PT[k][t] = sum_s of the transconv slabs [S][1][Tl][K32] (of the residual), or, with `den` ([Tl][K32]) given,
// den[t][k] - that sum (the projection as denomH - numH: slabs of transconv(W, data));   grid (ceil(Tl/64), KB), block 256
// snap (may be NULL): the launch in front of the persistent row pipeline also takes the pipeline's snapshot of H and H' (own columns;
// [2][TP][K32]: two 6.4 MB blit copies of 14 + 9 us at config 5 before) and clears its nflags progress flags (a 6 us fill before).
static __global__ __launch_bounds__(256) void hals_p_init_kernel(float *PT, const float *slabs, const float *den, int S, int Tl, int K32, int TPp,
                                                                 const float *H, const float *Ht, float *snap, int TP, int PADL, int *flags, int nflags)
{
    __shared__ float tile[32][65];
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * 64, kb = blockIdx.y;
    const size_t TK = (size_t)Tl * K32;
    {
        const int kk = tid & 31;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            int tt = q * 8 + (tid >> 5), t = t0 + tt;
            float v = 0.f;
            if (t < Tl) {
                size_t idx = (size_t)t * K32 + kb * 32 + kk;
                for (int s = 0; s < S; ++s) v += slabs[(size_t)s * TK + idx];
                if (den) v = den[idx] - v;
                if (snap) snap[(size_t)(PADL + t) * K32 + kb * 32 + kk] = H[(size_t)(PADL + t) * K32 + kb * 32 + kk];
            }
            tile[kk][tt] = v;
        }
    }
    __syncthreads();
    {
        const int tt = tid & 63;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            int kk = q * 4 + (tid >> 6);
            PT[(size_t)(kb * 32 + kk) * TPp + t0 + tt] = tile[kk][tt]; // t0 + tt < TPp (padded)
            if (snap && t0 + tt < Tl) {
                const size_t at = (size_t)(kb * 32 + kk) * TP + PADL + t0 + tt;
                snap[(size_t)TP * K32 + at] = Ht[at];
            }
        }
    }
    if (flags && blockIdx.x == 0 && blockIdx.y == 0)
        for (int e = tid; e < nflags; e += 256) flags[e] = 0;
}

// PW[l][l'][k][k'] = sum_n Wn[l][n][k] * Wn[l'][n][k'] on the MFMA pipe: one WORKGROUP per (l <= l', k block, k' block),
// its PW_NW = 8 waves take every eighth batch of 32 rows of n and add their partial sums through LDS in wave order
// (deterministic); the pair (l', l) is the transpose and is written from the same result.  Both operands are 128-byte rows
// of Wn read straight from L2.  grid (L*(L+1)/2, KB*KB), block 64 * PW_NW.  (Four waves until round 4: 210 workgroups are fewer than
// the chip's SIMDs, so the launch lasts as long as ONE wave's chain of N / 8 MFMAs and its L2 round trips: 17.8 us at N = 2000.)
// (Round 2 ran one wave per ordered pair: L*L waves of N/2 dependent MFMAs each -- 27 us of MFMA issue per wave at
// N = 2000 whatever the chip does, 55 us measured, independent of T.)
#define PW_NW 8
static __global__ __launch_bounds__(64 * PW_NW) void hals_pw_kernel(const float *Wn, float *PW, int N, int L, int Np, int K32, int KB)
{
    __shared__ float part[PW_NW][16][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // (scalar: uniform loop bounds)
    const int i = lane & 31, h = lane >> 5;
    int l = 0, rem = blockIdx.x; // pair index -> (l, lp), l <= lp: row l of the upper triangle holds L - l pairs
    while (rem >= L - l) { rem -= L - l; ++l; }
    const int lp = l + rem;
    const int kb = blockIdx.y / KB, kbp = blockIdx.y % KB;
    // buffer loads with scalar row offsets (no 64-bit address per load: with plain pointers the 32 loads of a batch in flight cost
    // 64 address registers on top of their 32 destinations, more than 8 waves per workgroup can have)
    const __amdgpu_buffer_rsrc_t ar = cmf_rsrc(Wn + (size_t)l * Np * K32, (size_t)Np * K32 * 4);
    const __amdgpu_buffer_rsrc_t br = cmf_rsrc(Wn + (size_t)lp * Np * K32, (size_t)Np * K32 * 4);
    const int aoff = (h * K32 + kb * 32 + i) * 4, boff = (h * K32 + kbp * 32 + i) * 4;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    // 32 rows of n per batch, the next batch's 32 operand loads in flight under this batch's 16 MFMAs.  Rows >= N of Wn are
    // zero up to Np, a multiple of 128.
    const int NB = (N + 31) & ~31;
    float av[2][16], bv[2][16];
    auto load = [&](float (&x)[16], float (&y)[16], int n0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            x[q] = cmf_bload(ar, aoff, (n0 + 2 * q) * K32 * 4);
            y[q] = cmf_bload(br, boff, (n0 + 2 * q) * K32 * 4);
        }
    };
    auto mac = [&](const float (&x)[16], const float (&y)[16]) {
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[q], y[q], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[q + 1], y[q + 1], acc1, 0, 0, 0);
        }
    };
    const int first = 32 * wave, step = 32 * PW_NW; // batches wave, wave + PW_NW, ...
    if (first < NB) load(av[0], bv[0], first);
    for (int n0 = first; n0 < NB; n0 += 2 * step) {
        if (n0 + step < NB) load(av[1], bv[1], n0 + step);
        mac(av[0], bv[0]);
        if (n0 + step < NB) {
            if (n0 + 2 * step < NB) load(av[0], bv[0], n0 + 2 * step);
            mac(av[1], bv[1]);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][r][lane] = acc0[r] + acc1[r];
    __syncthreads();
    for (int o = threadIdx.x; o < 16 * 64; o += 64 * PW_NW) {
        const int r = o >> 6, ln = o & 63;
        float v = part[0][r][ln];
#pragma unroll
        for (int w = 1; w < PW_NW; ++w) v += part[w][r][ln];
        const int k = kb * 32 + cmf_crow(r, ln >> 5), kp = kbp * 32 + (ln & 31);
        PW[(((size_t)l * L + lp) * K32 + k) * K32 + kp] = v;
        if (l != lp) PW[(((size_t)lp * L + l) * K32 + kp) * K32 + k] = v;
    }
}

// GW[k][k'][e+L-1] = sum_l PW[l][l-e][k][k'], e in (-L, L)        (full-window taps)
// GE[k][i][k'][e+L-1]: the same with only lags l < Lt(i) = ne - i (truncated window of edge column t = t_edge0 + i,
// Lt = Tl - t), i in [0, ne).  One thread per (k, e, k'), k' fastest across threads (its L reads are 128-byte rows shared
// with its neighbours); the edge tables are the running sums of the same walk over l -- Lt = l + 1 after lag l -- so all
// ne + 1 tables cost L reads per thread.  (One thread per OUTPUT, with the tap index fastest, read every term of every
// table separately: 23 us at K = 32, L = 20, independent of T.)
// GWt (optional): the full-window taps once more as [k'][e][k] with the tap count padded to Ep = 2L (gram_h_mfma_kernel's B
// operand; a separate transposing launch until round 4).
static __global__ void hals_gw_kernel(const float *PW, float *GW, float *GE, int L, int K32, int ne, int Tl, int t_edge0, float *GWt, int Ep)
{
    const int E = 2 * L - 1;
    const size_t total = (size_t)K32 * K32 * E;
    (void)Tl; (void)t_edge0;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int kp = (int)(idx % K32);
        const int ei = (int)((idx / K32) % E);
        const int k = (int)(idx / ((size_t)E * K32));
        const int e = ei - (L - 1);
        float s = 0.f;
        for (int l0 = 0; l0 < L; l0 += 8) { // eight terms in flight, added in lag order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int l = l0 + u, lp = l - e;
                v[u] = (l < L && lp >= 0 && lp < L) ? PW[(((size_t)l * L + lp) * K32 + k) * K32 + kp] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int l = l0 + u;
                if (l < L) {
                    s += v[u];
                    const int i = ne - (l + 1); // the edge column whose window holds exactly the lags 0 .. l
                    if (i >= 0 && i < ne) GE[(((size_t)k * ne + i) * K32 + kp) * E + ei] = s;
                }
            }
        }
        GW[((size_t)k * K32 + kp) * E + ei] = s;
        if (GWt) {
            GWt[((size_t)kp * Ep + ei) * K32 + k] = s;
            if (ei == E - 1)
                for (int z = E; z < Ep; ++z) GWt[((size_t)kp * Ep + z) * K32 + k] = 0.f;
        }
    }
}

// Serial Gauss-Seidel sweep of ONE row k of H (hals.jl:121-148) by ONE wave.  Lane j holds the
// pending P value of the column t' with t' % 64 == j inside the window [t, t+64); the same-row taps
// rotate one lane per step (DPP wave_ror), so a step is: 2 readlanes, 4 dependent VALU ops, 1 FMA.
// Writes the new H row (both layouts) and the per-column change D[t] for the cross-row push.
struct HalsRowParams {
    float *PT;        // [K32][TPp]
    float *H, *Ht;    // [TP][K32], [K32][TP]
    float *D;         // [TPp]
    const float *GW;  // [K32][K32][2L-1]
    const float *GE;  // [K32][ne][K32][2L-1]
    int k, Tl, L, K32, TP, TPp, PADL, ne, t_edge0;
    int t_begin, t_end; // column segment of this launch (t_begin multiple of 64; t_end multiple of 64 or Tl)
    float l1, l2;
};

// lane i <- lane i+1; lane 63 keeps `fill`
__device__ __forceinline__ float cmf_wave_shl1(float v, float fill)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}
// lane i <- lane i+1, lane 63 <- lane 0
__device__ __forceinline__ float cmf_wave_rol1(float v)
{
    // bound_ctrl = true: every lane has a source, and with it the `old` operand is dead (no v_mov 0 per rotation)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x134, 0xf, 0xf, true));
}
// lane i <- lane i-1, lane 0 <- lane 63
__device__ __forceinline__ float cmf_wave_ror1(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x13C, 0xf, 0xf, true));
}
// a * b + c as one opaque VALU instruction: keeps the SLP vectoriser from packing it with an unrelated FMA
// (v_pk_fma_f32) whose operands arrive later
__device__ __forceinline__ float cmf_fma_opaque(float a, float b, float c)
{
    float r;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// Two FMAs in one VALU instruction (v_pk_fma_f32: 4 cycles for the wave, like one v_fma_f32): r = w * b + c for the two rows held in
// w / c, b = one half of the pair bp (HI = false: bp.x, true: bp.y; op_sel broadcasts it, so two consecutive steps share one
// register pair).  Opaque to the compiler like cmf_fma_opaque.  Checked against fmaf on the device (tools: see DESIGN.md section 4e).
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool HI>
__device__ __forceinline__ f32x2 cmf_pk_fma_opaque(f32x2 w, f32x2 bp, f32x2 c)
{
    f32x2 r;
    if (HI) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(w), "v"(bp), "v"(c));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(w), "v"(bp), "v"(c));
    return r;
}
__device__ __forceinline__ float cmf_lane0(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// ---- how a row sweep meets the other rows ----------------------------------------------------------------------------
// HalsNoSync: the stage pipeline (hals_h_stage_kernel) -- everything a segment needs was pushed by earlier launches.
// HalsFlagSync: the persistent pipeline (hals_h_persist_kernel) -- the cross-row terms of block c (64 columns) of this
// row's P are added by a puller workgroup, which raises pulled[c % P] to c + 1 when the block is in memory; the sweep
// publishes its own progress (blocks whose changes D are in memory) for the pullers of the next row.  All hand-offs
// follow MI355X_MICROARCH.md "inter-workgroup visibility": payload stored sc1 (each 128-byte line whole, by one store
// instruction), s_waitcnt vmcnt(0), then an sc1 flag store; readers poll the flag with sc1 loads and read the payload with
// sc1 loads.  Every wait is bounded: a poll loop that runs out sets the abort word (and the host's status word) and every
// other loop leaves on seeing it, so the grid always drains.
#define HALS_FLAG_STRIDE 32      // ints: one 128-byte line per flag
#define HALS_POLL_LIMIT (1 << 21) // ~2 s of polling: three orders of magnitude above the longest legitimate wait

// wait until *flag >= need; false when the pipeline was aborted (or this wait ran out and aborted it)
__device__ __forceinline__ bool hals_wait_flag(const int *flag, int need, int *abort_word, int *host_status)
{
#pragma nounroll
    for (int n = 0; n < HALS_POLL_LIMIT; ++n) {
        const int v = cmf_load_sc1(flag), a = cmf_load_sc1(abort_word); // both in flight together: one round trip per poll
        // both values are looked at before either branch: a return with the second load still pending (in the compiler's
        // books) makes it wait for EVERYTHING outstanding -- the publish store just issued, a microsecond -- at the next
        // write of that load's register in the caller's hot loop (measured: 1.36 -> 1.72 us per block, by register luck)
        const bool done = (v >= need), dead = (a != 0);
        if ((int)done | (int)dead) return done;
        __builtin_amdgcn_s_sleep(2);
    }
    cmf_store_sc1(abort_word, 1);
    __hip_atomic_store(host_status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return false;
}

struct HalsNoSync {
    static constexpr bool PERSIST = false;
    __device__ __forceinline__ float load_p(const float *p) const { return *p; }
    __device__ __forceinline__ void store_d(float *p, float v) const { *p = v; }
    __device__ __forceinline__ void store_h(float *p, float v) const { *p = v; }
    __device__ __forceinline__ bool gate(int) const { return true; }
    __device__ __forceinline__ int gate_issue(int) const { return 0; }
    __device__ __forceinline__ bool gate_check(int, int) const { return true; }
    __device__ __forceinline__ void publish(int) const {}
    __device__ __forceinline__ void stamp(int) const {}
};

struct HalsFlagSync {
    static constexpr bool PERSIST = true;
    const int *pulled; // this row's P flags, HALS_FLAG_STRIDE apart (NULL for row 0: nothing is pulled into it)
    int *prog;         // this row's progress flag
    int *abort_word, *host_status;
    int P, nblk;
    // this row's own edge taps GE[k][i][k][e >= 0] as [ne][L], then 1 / (norm + eps + l2) as [ne], in LDS
    const __attribute__((address_space(3))) float *edge;
    unsigned long long *stamps; // debug (CMF_HALS_STAMPS): s_memtime at the end of every block of this row, or NULL
    __device__ __forceinline__ void stamp(int blk) const { if (stamps && threadIdx.x == 0) stamps[blk] = __builtin_amdgcn_s_memrealtime(); }
    __device__ __forceinline__ float load_p(const float *p) const { return cmf_load_sc1(p); }
    __device__ __forceinline__ void store_d(float *p, float v) const { cmf_store_sc1(p, v); }
    // H' (the conv's operand layout) is stored agent-scope too: a conv launch on other CUs may be chasing the last row's progress flag
    // (conv3_chase_kernel) and reads the columns a sweeper has published while this kernel is still running
    __device__ __forceinline__ void store_h(float *p, float v) const { cmf_store_sc1(p, v); }
    __device__ __forceinline__ bool gate(int c) const // block c of this row's P is complete
    {
        if (!pulled || c >= nblk) return true;
        return hals_wait_flag(pulled + (c % P) * HALS_FLAG_STRIDE, c + 1, abort_word, host_status);
    }
    __device__ __forceinline__ int gate_issue(int c) const // start reading the flag of block c; its value goes to gate_check
    {
        if (!pulled || c >= nblk) return 0x7fffffff;
        return cmf_load_sc1(pulled + (c % P) * HALS_FLAG_STRIDE);
    }
    __device__ __forceinline__ bool gate_check(int c, int seen) const { return seen >= c + 1 ? true : gate(c); }
    __device__ __forceinline__ void publish(int done) const // every store of blocks < done has completed (caller drained vmcnt)
    {
        if (threadIdx.x == 0) cmf_store_sc1(prog, done);
    }
};

template <class Sync>
__device__ __forceinline__ void hals_h_row_sweep(const HalsRowParams &q, int lane, const Sync &sy)
{
    const int L = q.L, E = 2 * L - 1, k = q.k;
    float *Prow = q.PT + (size_t)k * q.TPp;
    float *Hrow = q.Ht + (size_t)k * q.TP + q.PADL;
    const float *gk = q.GW + ((size_t)k * q.K32 + k) * E + (L - 1); // gk[e], e >= 0
    const float nrm = gk[0];
    const float inv_den = 1.0f / (nrm + CMF_EPS_F + q.l2);
    const float g = (lane >= 1 && lane < L) ? gk[lane] : 0.f; // same-row taps, lane = column offset
    const int nfull = q.t_edge0;                              // columns [0, nfull) have the full lag window
    int tb = q.t_begin;
    if (!sy.gate(tb >> 6) || !sy.gate((tb >> 6) + 1)) return;
    float p = sy.load_p(Prow + tb + lane);       // pending P of columns tb + lane (PT is zero-padded beyond Tl)
    float pn = sy.load_p(Prow + tb + 64 + lane); // columns tb + 64 + lane: enter the window one per step
    float hreg = Hrow[tb + lane];    // H_old of columns tb + lane
    const int t_stop = q.t_end < q.Tl ? q.t_end : q.Tl;
    int t_first = tb;                        // first column left to the generic path
    float grot0 = g, hnew0 = 0.f, dreg0 = 0.f; // its initial tap rotation and the results of a partial block's columns
    // ---- fast path: whole 64-column blocks with the full window.  The window slides one lane per step (DPP wave
    // shift), so the active column is always lane 0 and the taps never move.  The recurrence is carried in the change
    // d_t = x_t - h_t, with the lag-1 tap taken out of the wave-wide update and the state kept in "numerator" form
    //     V_t[i] = (c - h)_{t+i} - inv_den * w_t[i]        w_t = pending P of columns t.., without d_{t-1}'s lag-1 term
    //     ps     = shl1(V_t)                                 (column t+64 enters at lane 63; ps does not need d_t)
    //     d_{t+1} = max(kappa * d_t + ps[0], -h_{t+1})       kappa = -inv_den * g[1]          <- the chain: FMA, MAX
    //     V_{t+1} = ps + d_t * gsn                           gsn[i] = -inv_den * g[i+1] (i >= 1), gsn[0] = 0
    // so the chain from one column to the next is one FMA and one MAX on lane 0; the v_readfirstlane of d, the wave-wide
    // update and the shifts trail it by a step.  The direct form (readfirstlane(p), FMA, MAX, SUB, wave-wide FMA, shift:
    // everything in the chain) cost ~88 cycles per column.
    if (tb + 64 <= nfull && tb + 64 <= t_stop) {
        const float g1 = (L > 1) ? gk[1] : 0.f;
        const float kappa = -inv_den * g1;
        const float gsn = (lane >= 1 && lane + 1 < L) ? -inv_den * gk[lane + 1] : 0.f;
        float w = p;         // complete on entry (a hand-over or the initial P): the previous column's change is in it
        float dprev = 0.f;   // lane 0: change of the previous column whose lag-1 term is still missing from the state
        // cmh = c - h with c = (h_old*nrm - l1)/(nrm+eps+l2), per column.  It enters V here and leaves it again in the
        // conversion back to pending values below; both must see the SAME float (one variable, pinned): left as two
        // expressions the compiler may contract them differently, and a column whose pending value is exactly 0 -- W_k
        // zero on the lags of a truncated window, where the update divides by eps alone -- comes back as an ulp of c - h.
        float cmh = (hreg * nrm - q.l1) * inv_den - hreg;
        asm volatile("" : "+v"(cmh));
        float V = fmaf(-inv_den, w, cmh);
        float hahead = Hrow[tb + 64 + lane]; // H_old one block ahead: loaded a whole block before its first use (a load at
                                             // the top of the block it is needed in puts an L2 round trip into every block's chain)
        // A block's results are stored half a block after its end (at the middle of the next block): stored at its end,
        // they are still in flight when the top of the next block waits for its prefetched operands -- vmcnt counts loads
        // and stores alike -- and every block pays part of a store round trip.
        float hnew_prev = 0.f, dvec_prev = 0.f;
        int tb_prev = -1;
        auto store_prev = [&]() {
            if (tb_prev < 0) return;
            sy.store_h(Hrow + tb_prev + lane, hnew_prev);
            q.H[(size_t)(q.PADL + tb_prev + lane) * q.K32 + k] = hnew_prev;
            sy.store_d(q.D + tb_prev + lane, dvec_prev);
        };
        for (; tb + 64 <= nfull && tb + 64 <= t_stop; tb += 64) {
            // persistent pipeline: the flag of P block +2 is read now and looked at half a block later, together with the
            // completion of the previous block's stores -- neither round trip sits in the column chain
            const int seen = sy.gate_issue((tb >> 6) + 2);
            bool aborted = false;
            float pn2 = 0.f;
            if (!Sync::PERSIST) pn2 = Prow[tb + 128 + lane];
            const float hreg2 = hahead;
            hahead = Hrow[tb + 128 + lane];
            float cmh2 = (hreg2 * nrm - q.l1) * inv_den - hreg2;
            asm volatile("" : "+v"(cmh2));
            float vnr = cmf_wave_rol1(fmaf(-inv_den, pn, cmh2)); // lane 63 holds the column that enters next
            float mhrot = -hreg;                                          // lane 0 = -h of the column being swept
            float dvec = 0.f;                                             // collects the changes of the block's columns
            // column 0 of the block: its rh is lane 0 of V itself
            float d;
            asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(fmaf(kappa, dprev, V)), "v"(mhrot));
#pragma unroll
            for (int j = 0; j < 64; ++j) {
                const float ps = cmf_wave_shl1(V, vnr);        // shl1(V_j): lane 0 = rh of column j+1
                const float s_d = cmf_lane0(d);
                V = fmaf(s_d, gsn, ps);                        // hals.jl:146 on the projected state, lags >= 2
                asm("v_writelane_b32 %0, %1, %2" : "+v"(dvec) : "s"(s_d), "n"(j)); // lane j <- d_j
                mhrot = cmf_wave_rol1(mhrot);
                vnr = cmf_wave_rol1(vnr);
                dprev = d;
                if (j + 1 < 64) // hals.jl:152-153 as a change: x - h_old = max(q, 0) - h_old = max(q - h_old, -h_old)
                    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(fmaf(kappa, dprev, ps)), "v"(mhrot));
                if (j == 31) {
                    if (Sync::PERSIST) {
                        cmf_drain_vmem();                           // the stores issued at the middle of the block before have completed
                        sy.publish(tb_prev >= 0 ? tb_prev >> 6 : 0); // ... and with them every block before that one
                    }
                    if (Sync::PERSIST) aborted = !sy.gate_check((tb >> 6) + 2, seen); // (no exit from inside the unrolled block)
                    store_prev(); // (after the look at the flag: a use of a loaded value behind stores waits for the stores)
                    if (Sync::PERSIST) pn2 = sy.load_p(Prow + tb + 128 + lane);
                }
            }
            if (Sync::PERSIST && aborted) return;
            hnew_prev = hreg + dvec;
            dvec_prev = dvec;
            tb_prev = tb;
            sy.stamp(tb >> 6);
            pn = pn2;
            hreg = hreg2;
            cmh = cmh2;
        }
        // ---- what is left in front of the edge (or of the segment's end) is less than a block: the same recurrence as a
        // rolled loop of m steps.  The generic path below costs ~320 cycles a column against ~50 here, and in the persistent
        // pipeline the rows' tails run strictly one after the other.  C carries c - h in window form beside V (bit copies:
        // see cmh above); afterwards the window is turned back so that lane j holds the column with t % 64 == j again.
        store_prev(); // the last whole block
        const int t_lim = nfull < t_stop ? nfull : t_stop;
        const int m = t_lim - tb; // < 64
        const float den = nrm + CMF_EPS_F + q.l2;
        // the whole blocks are published (the next row's last pulls wait for them) once their stores have had time to
        // complete: 40 columns into the partial block, or here if there is none that long
        const int pub_at = (m > 40) ? 40 : -1;
        if (Sync::PERSIST && pub_at < 0) { cmf_drain_vmem(); sy.publish(tb >> 6); }
        if (m > 0) {
            float cmh2 = (hahead * nrm - q.l1) * inv_den - hahead;
            asm volatile("" : "+v"(cmh2));
            float vnr = cmf_wave_rol1(fmaf(-inv_den, pn, cmh2));
            float c2r = cmf_wave_rol1(cmh2);
            float C = cmh;
            float mhrot = -hreg;
            float dvec = 0.f;
            float d;
            asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(fmaf(kappa, dprev, V)), "v"(mhrot));
#pragma nounroll
            for (int j = 0; j < m; ++j) {
                const float ps = cmf_wave_shl1(V, vnr);
                const float s_d = cmf_lane0(d);
                V = fmaf(s_d, gsn, ps);
                dvec = (lane == j) ? s_d : dvec;
                mhrot = cmf_wave_rol1(mhrot);
                vnr = cmf_wave_rol1(vnr);
                C = cmf_wave_shl1(C, c2r);
                c2r = cmf_wave_rol1(c2r);
                dprev = d;
                asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(cmf_fma_opaque(kappa, dprev, ps)), "v"(mhrot)); // (the last one is not used)
                if (Sync::PERSIST && j == pub_at) { cmf_drain_vmem(); sy.publish(tb >> 6); }
            }
            const float pw = (C - V) * den + ((lane == 0) ? cmf_lane0(dprev) * g1 : 0.f); // lane i: column tb + m + i
            const int src = ((lane - m) & 63) * 4;
            p = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, pw)));
            grot0 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, g)));
            hnew0 = (lane < m) ? hreg + dvec : 0.f;
            dreg0 = (lane < m) ? dvec : 0.f;
            t_first = tb + m;
        } else {
            // back to the complete pending values: w = ((c - h) - V) / inv_den; the last column's lag-1 term goes to lane 0
            p = (cmh - V) * den + ((lane == 0) ? cmf_lane0(dprev) * g1 : 0.f);
            t_first = tb;
        }
    }
    // ---- generic path (right-edge columns, hals.jl:136, and rows too short for the fast path): lane j
    // holds the column t' with t' % 64 == j; the taps rotate instead of the window.
    if (t_first >= t_stop) {
        if (t_first > tb && tb + lane < t_first) { // (L = 1: no edge) the partial block closed the row
            sy.store_h(Hrow + tb + lane, hnew0);
            q.H[(size_t)(q.PADL + tb + lane) * q.K32 + k] = hnew0;
            sy.store_d(q.D + tb + lane, dreg0);
        }
        // segment ends on a block boundary: hand the pending window (same-row pushes applied) to the
        // launch that continues this row
        if (t_stop < q.Tl) Prow[tb + lane] = p;
        if (Sync::PERSIST) { cmf_drain_vmem(); sy.publish((t_stop + 63) >> 6); }
        return;
    }
    if (Sync::PERSIST) // the rest of the row runs on complete P blocks only
        for (int c = (tb >> 6) + 2; c <= ((t_stop + 63) >> 6) + 1; ++c)
            if (!sy.gate(c)) return;
    float grot = grot0;
    float hnew = hnew0, dreg = dreg0;
    // one block ahead of their use (a load at the block boundary would sit in the chain): H_old of the next block and the
    // pending values of the block after it
    float hnext = Hrow[(t_first & ~63) + 64 + lane];
    float pnn = sy.load_p(Prow + (t_first & ~63) + 128 + lane);
    // closes column t: lane t % 64 takes the column's results and the pending value of column t + 64; whole blocks are
    // stored (and, in the persistent pipeline, published)
    auto close_column = [&](int t, float x, float d) {
        const int idx = t & 63;
        if (lane == idx) { hnew = x; dreg = d; p = pn; }
        if (idx == 63 || t == t_stop - 1) {
            const int t0 = t - idx;
            if (t0 + lane < q.Tl) {
                sy.store_h(Hrow + t0 + lane, hnew);
                q.H[(size_t)(q.PADL + t0 + lane) * q.K32 + k] = hnew;
                sy.store_d(q.D + t0 + lane, dreg);
            }
            sy.stamp(t0 >> 6);
            if (idx == 63) { // (no publish here: the row's last one follows within a few microseconds, and a drain costs one)
                pn = pnn;
                hreg = hnext;
                pnn = sy.load_p(Prow + t0 + 192 + lane);
                hnext = Hrow[t0 + 128 + lane];
            }
        }
    };
    int t = t_first;
    for (; t < t_stop && t < nfull; ++t) { // full-window columns of rows too short for the fast path
        const int idx = t & 63;
        const float s_p = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p), idx));
        const float s_h = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hreg), idx));
        const float x = fmaxf((s_h * nrm - s_p - q.l1) * inv_den, 0.f);
        const float d = x - s_h;
        p = fmaf(d, grot, p);
        close_column(t, x, d);
        grot = cmf_wave_ror1(grot);
    }
    // edge columns: truncated windows with their own norm and taps (GE), fetched one column ahead -- from LDS in the
    // persistent kernel: read from memory at their point of use they put two round trips into every edge column
    float nrm_n = 0.f, inv_n = 0.f, gel_n = 0.f;
    bool gel_ok = false; // (the select is applied where the tap is used, so that the fetch is not waited for at once)
    auto edge_fetch = [&](int tt) {
        const int e = (lane - tt) & 63;
        if constexpr (Sync::PERSIST) {
            const int i = tt - nfull;
            nrm_n = sy.edge[i * L];
            inv_n = sy.edge[q.ne * L + i];
            gel_n = sy.edge[i * L + ((e < L) ? e : 0)]; // (an address inside the table for every lane)
            gel_ok = (e >= 1 && e < L);
        } else {
            const float *ge = q.GE + (((size_t)k * q.ne + (tt - nfull)) * q.K32 + k) * E + (L - 1);
            nrm_n = ge[0];
            inv_n = 1.0f / (nrm_n + CMF_EPS_F + q.l2);
            gel_n = (e >= 1 && e < L) ? ge[e] : 0.f;
            gel_ok = true;
        }
    };
    if (t < t_stop) edge_fetch(t);
    for (; t < t_stop; ++t) {
        const int idx = t & 63;
        const float nrm_e = nrm_n, inv_e = inv_n, ge_l = gel_ok ? gel_n : 0.f;
        if (t + 1 < t_stop) edge_fetch(t + 1);
        const float s_p = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p), idx));
        const float s_h = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hreg), idx));
        const float x = fmaxf((s_h * nrm_e - s_p - q.l1) * inv_e, 0.f);
        const float d = x - s_h;
        p = fmaf(d, ge_l, p);
        close_column(t, x, d);
    }
    if (t_stop < q.Tl) Prow[t_stop + lane] = p; // (t_stop is a multiple of 64 here) window for the next launch
    if (Sync::PERSIST) { cmf_drain_vmem(); sy.publish((t_stop + 63) >> 6); }
}

// Cross-row push of row k's changes: PT[k'][t'] += sum_e D[t'-e] * taps(t'-e)[k][k'][e] for k' > k.
// sources: columns [s_begin, s_end) of row k; targets: [s_begin-(L-1), s_end+L-1) of one later row kp;
// cb = 256-column block of the target range, tid = thread in the block (all 256 threads call this).
// The block's window of D (256 + 2(L-1) columns, zero outside the source segment) and the 2L-1 full-window taps are
// staged in LDS first: read straight from memory, the tap loop is a chain of ~2(2L-1) dependent L2 round trips per
// thread, and a stage's pushes then take longer than its row sweeps (10-15 us against 12 us at 384-column segments).
// Source columns in the right edge (t >= t_edge0: truncated windows, hals.jl:136) have per-column taps in GE.
__device__ __forceinline__ void hals_h_push(float *PT, const float *D, const float *GW, const float *GE,
                                            int k, int kp, int Tl, int L, int K32, int TPp, int ne, int t_edge0,
                                            int s_begin, int s_end, int cb, int tid)
{
    __shared__ float Ds[256 + 128];
    __shared__ float taps[128];
    const int E = 2 * L - 1;
    const int tp0 = s_begin - (L - 1) + cb * 256; // first target column of the block
    const int w0 = tp0 - (L - 1);                 // first source column the block can see
    for (int j = tid; j < 256 + 2 * (L - 1); j += 256) {
        const int t = w0 + j;
        Ds[j] = (t >= s_begin && t < s_end) ? D[t] : 0.f;
    }
    if (tid < E) taps[tid] = GW[((size_t)k * K32 + kp) * E + tid]; // taps[e + L - 1]
    __syncthreads();
    const int tp = tp0 + tid;
    if (tp < 0 || tp >= Tl || tp >= s_end + L - 1) return;
    float s = 0.f;
    if (tp0 + 255 + (L - 1) < t_edge0 || s_end <= t_edge0) { // no source column of this block lies in the right edge
        for (int e = -(L - 1); e <= L - 1; ++e) s = fmaf(Ds[tid + (L - 1) - e], taps[e + L - 1], s); // t = tp - e
    } else {
        for (int e = -(L - 1); e <= L - 1; ++e) {
            const int t = tp - e;
            if (t < s_begin || t >= s_end) continue;
            const float tap = (t < t_edge0) ? taps[e + L - 1] : GE[(((size_t)k * ne + (t - t_edge0)) * K32 + kp) * E + (L - 1) + e];
            s = fmaf(Ds[tid + (L - 1) - e], tap, s);
        }
    }
    PT[(size_t)kp * TPp + tp] += s;
}


// One pipeline stage of the H sweep (hals.jl:121-154).  Row k sweeps its column segment sg = stage - lag*k, and
// the changes of the segments swept in the previous stage are pushed to the later rows.  Row k's segments are
// [sg*seg - skew*k, (sg+1)*seg - skew*k): each row's grid is shifted left by `skew` columns against the row above.
//   lag 2, skew 128 (default): a segment of row k+1 ends 128 columns before the same-numbered segment of row k, so
//     it is influenced by row k's segments <= sg only (their pushes reach L-1 <= 63 columns to the left), all pushed
//     by stage sg + 2k + 1 < sg + 2(k+1); within a launch, a row's sweep -- which also hands the 64-column pending
//     window behind its segment over to the next stage -- ends at least 128 - 64 - (L-1) >= 1 columns before the first
//     column any push of that launch touches, and pushes from different rows into one row are >= seg - 2(L-1) apart.
//   lag 3, skew 0: the round-1 schedule (unshifted grids need one more stage of distance).
// Either way everything that influences a segment has been pushed before it is swept and the sweeps and pushes of one
// launch touch disjoint columns (segments are >= 256 columns), so the update order is exactly the reference's while
// ~nseg/lag rows are in flight.
// grid: (K + K*CB, max(1, K-1)), block 256.   blockIdx.x < K: sweep of row blockIdx.x (wave 0 only).
struct HalsStageParams {
    HalsRowParams row;  // k, D, t_begin, t_end filled per block
    float *Dall;        // [K32][TPp]
    int K, seg, nseg, CB, stage;
    int lag, skew;
};

// columns [*b, *e) of segment sg of row k (clipped to [0, Tl)); false when the segment is empty
__device__ __forceinline__ bool hals_segment(const HalsStageParams &sp, int k, int sg, int *b, int *e)
{
    if (sg < 0) return false;
    long long t0 = (long long)sg * sp.seg - (long long)sp.skew * k, t1 = t0 + sp.seg;
    if (t1 <= 0 || t0 >= sp.row.Tl) return false;
    *b = t0 < 0 ? 0 : (int)t0;
    *e = t1 > sp.row.Tl ? sp.row.Tl : (int)t1;
    return true;
}

static __global__ __launch_bounds__(256) void hals_h_stage_kernel(HalsStageParams sp)
{
    const int bx = blockIdx.x;
    if (bx < sp.K) {
        if (blockIdx.y != 0 || threadIdx.x >= 64) return;
        const int k = bx;
        HalsRowParams q = sp.row;
        if (!hals_segment(sp, k, sp.stage - sp.lag * k, &q.t_begin, &q.t_end)) return;
        q.k = k;
        q.D = sp.Dall + (size_t)k * q.TPp;
        hals_h_row_sweep(q, threadIdx.x, HalsNoSync());
    } else {
        const int k = (bx - sp.K) / sp.CB, cb = (bx - sp.K) % sp.CB;
        const int kp = k + 1 + blockIdx.y;
        int s_begin, s_end;
        if (kp >= sp.K || !hals_segment(sp, k, sp.stage - 1 - sp.lag * k, &s_begin, &s_end)) return; // swept in the previous stage
        const HalsRowParams &r = sp.row;
        hals_h_push(r.PT, sp.Dall + (size_t)k * r.TPp, r.GW, r.GE, k, kp, r.Tl, r.L, r.K32, r.TPp, r.ne, r.t_edge0,
                    s_begin, s_end, cb, threadIdx.x);
    }
}


// ---------------------------------------------------------------------------------------------
// The H sweep as ONE persistent launch (round 2; replaces the ~200 stage launches when the grid fits the chip).
//   workgroup k < K          : wave 0 sweeps row k from column 0 to Tl (hals_h_row_sweep with HalsFlagSync);
//   workgroup K + (k-1)P + j : puller j of row k >= 1: for its blocks b = j, j + P, ... (64 columns each) it waits until
//       row k-1 has published blocks <= b + 1 (rows above k-1 are further ahead by induction), gathers the changes D of
//       all rows k'' < k around the block, adds  sum_k'' sum_e D[k''][t' - e] * taps(k'', k)[e]  to P[k][block] -- the
//       cross-row pushes of hals.jl:146 in pull form, so only one workgroup ever writes a given block of P -- and raises
//       its flag.  The update order is the reference's: a block is swept only after every change that reaches it
//       (rows above, columns up to L-1 to its right) has been applied.
// Row k trails row k-1 by about four blocks plus two hand-offs instead of the 640 columns and two launches of the stage
// pipeline.  The grid, K + (K-1)P workgroups, must be resident at once (the host checks it against the CU count);
// every wait is bounded (hals_wait_flag).
// ---------------------------------------------------------------------------------------------
struct HalsPersistParams {
    HalsRowParams row;  // k, D filled per workgroup; t_begin = 0, t_end = Tl
    float *Dall;        // [K32][TPp]
    int *flags;         // [K prog | K * P pulled | abort], HALS_FLAG_STRIDE ints apart (zeroed before the launch)
    int *host_status;   // pinned host word: set to 1 when a wait ran out
    int K, P, nblk;
    int debug;          // timing experiments only: 1 = no gating, 2 = pullers skip their work; tests: 3 = pullers leave at once
    unsigned long long *stamps; // debug: [K][nblk] sweeper block-end times, then [K][nblk][4] puller phase times; or NULL
};

static __global__ __launch_bounds__(1024) void hals_h_persist_kernel(HalsPersistParams pp)
{
    extern __shared__ float hp_smem[];
    const int tid = threadIdx.x;
    const int K = pp.K, P = pp.P, nblk = pp.nblk;
    int *prog = pp.flags;
    int *pulled = pp.flags + (size_t)K * HALS_FLAG_STRIDE;
    int *abort_word = pp.flags + (size_t)(K + K * P) * HALS_FLAG_STRIDE;
    const HalsRowParams &r = pp.row;
    if ((int)blockIdx.x < K) { // ---- sweeper of row k
        if (tid >= 64) return;
        const int k = blockIdx.x;
        HalsRowParams q = r;
        q.k = k;
        q.D = pp.Dall + (size_t)k * q.TPp;
        q.t_begin = 0;
        q.t_end = q.Tl;
        {   // this row's edge taps -> LDS (hp_smem is sized for the pullers; ne * L floats fit: L <= 64)
            const int E = 2 * q.L - 1;
            for (int idx = tid; idx < q.ne * q.L; idx += 64) {
                const int i = idx / q.L, e = idx - i * q.L;
                hp_smem[idx] = q.GE[(((size_t)k * q.ne + i) * q.K32 + k) * E + (q.L - 1) + e];
            }
            for (int i = tid; i < q.ne; i += 64) // and 1 / (norm + eps + l2) of every edge column
                hp_smem[q.ne * q.L + i] = 1.0f / (q.GE[(((size_t)k * q.ne + i) * q.K32 + k) * E + (q.L - 1)] + CMF_EPS_F + q.l2);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        HalsFlagSync sy;
        sy.edge = (const __attribute__((address_space(3))) float *)hp_smem;
        sy.pulled = (k > 0 && P > 0 && pp.debug != 1) ? pulled + (size_t)k * P * HALS_FLAG_STRIDE : nullptr;
        sy.prog = prog + (size_t)k * HALS_FLAG_STRIDE;
        sy.abort_word = abort_word;
        sy.host_status = pp.host_status;
        sy.P = P;
        sy.nblk = nblk;
        sy.stamps = pp.stamps ? pp.stamps + (size_t)k * nblk : nullptr;
        hals_h_row_sweep(q, tid, sy);
        return;
    }
    // ---- puller j of row k
    if (pp.debug == 1 || pp.debug == 3) return;
    const int u = blockIdx.x - K;
    const int k = 1 + u / P, j = u % P;
    const int L = r.L, E = 2 * L - 1, W = 64 + 2 * (L - 1), K32 = r.K32, Tl = r.Tl, TPp = r.TPp;
    float *Gs = hp_smem;               // [k][E]: taps of source row k'' onto row k
    float *Ds = Gs + (size_t)(K - 1) * E; // [k][W]: D of the source rows around the block
    float *part = Ds + (size_t)(K - 1) * W; // [16][64]
    __shared__ int ok_s;
    for (int idx = tid; idx < k * E; idx += 1024) {
        const int k2 = idx / E, e = idx - k2 * E;
        Gs[idx] = r.GW[((size_t)k2 * K32 + k) * E + e];
    }
    const int wave = tid >> 6, lane = tid & 63;
    const int *prog_up = prog + (size_t)(k - 1) * HALS_FLAG_STRIDE;
    int *my_flag = pulled + ((size_t)k * P + j) * HALS_FLAG_STRIDE;
    float *Prow = r.PT + (size_t)k * TPp;
    for (int b = j; b < nblk; b += P) {
        const int need = (b + 2 < nblk) ? b + 2 : nblk;
        if (tid == 0) ok_s = hals_wait_flag(prog_up, need, abort_word, pp.host_status) ? 1 : 0;
        __syncthreads(); // also: everyone is done with Ds / part of the previous block
        if (!ok_s) return;
        unsigned long long *st = pp.stamps ? pp.stamps + (size_t)K * nblk + ((size_t)k * nblk + b) * 4 : nullptr;
        if (st && tid == 0) st[0] = __builtin_amdgcn_s_memrealtime();
        if (pp.debug == 2) { if (tid == 0) cmf_store_sc1(my_flag, b + 1); continue; }
        const int w0 = b * 64 - (L - 1);
        for (int base = 0; base < k * W; base += 4 * 1024) { // four loads in flight per thread, then the LDS writes
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * 1024 + tid;
                const int k2 = idx / W, c = idx - k2 * W, t = w0 + c;
                v[u] = (idx < k * W && t >= 0 && t < Tl) ? cmf_load_sc1(pp.Dall + (size_t)k2 * TPp + t) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * 1024 + tid;
                if (idx < k * W) Ds[idx] = v[u];
            }
        }
        const int tp = b * 64 + lane;
        const float p0 = (wave == 0) ? Prow[tp] : 0.f; // written by the kernel before this one (PT is padded to whole blocks)
        __syncthreads();
        if (st && tid == 0) st[1] = __builtin_amdgcn_s_memrealtime();
        float s = 0.f;
        if (b * 64 + 64 + (L - 1) <= r.t_edge0) { // no source column of this block lies in the right edge
            for (int k2 = wave; k2 < k; k2 += 16) {
                const float *dsr = Ds + (size_t)k2 * W + lane + (L - 1);
                const float *gr = Gs + (size_t)k2 * E + (L - 1);
                // two partial sums and eight reads in flight: taken one at a time the loop is a chain of LDS round trips
                // (100 cycles a tap: the whole compute phase of a block was 1.6 us for ONE source row)
                float s1 = 0.f;
#pragma unroll 8
                for (int e = -(L - 1); e + 1 <= L - 1; e += 2) { // source column t = tp - e
                    s = fmaf(dsr[-e], gr[e], s);
                    s1 = fmaf(dsr[-e - 1], gr[e + 1], s1);
                }
                s = fmaf(dsr[-(L - 1)], gr[L - 1], s) + s1; // (2L - 1 taps: the last one is left over)
            }
        } else { // the last blocks of the row: some source columns lie in the right edge (truncated windows, hals.jl:136)
            // Sources in front of the edge take the full-window taps from LDS as above (masked), the <= L-1 edge sources
            // their own taps from GE in a loop of their own: one coalesced row of taps per (source row, edge column), several
            // in flight.  (With both kinds in one loop -- a bounds test and a dependent global load per term -- these last
            // pulls took 7-20 us, and every row's end waits for them.)
            for (int k2 = wave; k2 < k; k2 += 16) {
                const float *dsr = Ds + (size_t)k2 * W + lane + (L - 1);
                // (the LDS taps through an address-space-3 pointer: left generic, the compiler folds the two sources of taps
                // into one selected pointer whose LDS-aperture check does not assemble -- "Illegal instruction detected")
                const __attribute__((address_space(3))) float *gr = (const __attribute__((address_space(3))) float *)hp_smem + k2 * E + (L - 1);
#pragma unroll 8
                for (int e = -(L - 1); e <= L - 1; ++e) {
                    const int t = tp - e;
                    const float d = (t >= 0 && t < r.t_edge0) ? dsr[-e] : 0.f;
                    s = fmaf(d, gr[e], s);
                }
                const float *ge = r.GE + ((size_t)k2 * r.ne * K32 + k) * E + (L - 1); // + i * K32 * E + e
                for (int i0 = 0; i0 < r.ne; i0 += 8) { // eight tap rows in flight, then their FMAs (more would spill: the
                    float gv[8], dv[8];                 // 1024-thread workgroup leaves 128 registers a lane)
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i2 = i0 + u, e = tp - (r.t_edge0 + i2);
                        const bool ok = i2 < r.ne && e >= -(L - 1) && e <= L - 1;
                        gv[u] = ok ? ge[(size_t)i2 * K32 * E + e] : 0.f;
                        dv[u] = ok ? dsr[-e] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) s = fmaf(dv[u], gv[u], s);
                }
            }
        }
        part[wave * 64 + lane] = s;
        __syncthreads();
        if (st && tid == 0) st[2] = __builtin_amdgcn_s_memrealtime();
        if (wave == 0) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < 16; ++w) tot += part[w * 64 + lane]; // fixed order: reproducible
            cmf_store_sc1(Prow + tp, p0 + tot); // 64 lanes, two whole 128-byte lines in one instruction
            cmf_drain_vmem();
            if (lane == 0) cmf_store_sc1(my_flag, b + 1);
            if (st && lane == 0) st[3] = __builtin_amdgcn_s_memrealtime();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// General sweeps: the same recurrences in the same order for shapes beyond the on-chip sweeps' limits (the reference
// takes any K, L: hals.jl:90-154; its own micro-benchmark runs L = 100, notebooks/benchmarks.ipynb cell 2).  Slower --
// state in LDS / memory instead of registers and wave lanes -- and without a shape limit of their own.
// ---------------------------------------------------------------------------------------------

// W sweep for any L * Kpad: ONE workgroup per unit n, the unit's projected state g[j] (j over the L*K32 columns of
// H_unfold) in LDS.  Step (k, l) -- k outer, lag inner, hals.jl:90-97 -- reads g[j] and the diagonal HH[j][j], forms the
// new w (hals.jl:104-110) in every thread, and adds (w_new - w_old) * HH[j][j'] to the entries j' that later steps of
// this sweep still read (the HH row comes from L2: all N workgroups walk the same rows).
// grid N, block 256, dynamic LDS: L*K32 floats.
static __global__ __launch_bounds__(256) void hals_w_sweep_gen_kernel(float *Wt, float *Wn, const float *G, const float *Gsub, const float *HH,
                                                                int N, int K, int L, int Np, int K32, int NpH, float l1, float l2)
{
    extern __shared__ float gen_g[];
    const int tid = threadIdx.x, n = blockIdx.x;
    const int LK = L * K32;
    for (int j = tid; j < LK; j += 256) {
        const size_t at = (size_t)j * Np + n;
        gen_g[j] = Gsub ? G[at] - Gsub[at] : G[at];
    }
    __syncthreads();
    for (int k = 0; k < K; ++k)
        for (int l = 0; l < L; ++l) {
            const int j = l * K32 + k;
            const float hjj = HH[(size_t)j * NpH + j];
            const float w_old = Wt[(size_t)j * Np + n];
            const float v = gen_g[j] - w_old * hjj;                       // hals.jl:104 projected
            const float w_new = fmaxf((-v - l1) / (hjj + CMF_EPS_F + l2), 0.f); // hals.jl:110
            const float d = w_new - w_old;
            __syncthreads(); // everyone has read g[j] (and this thread's Wt value) before the row update touches them
            const float *row = HH + (size_t)j * NpH;
            for (int jp = tid; jp < LK; jp += 256) gen_g[jp] = fmaf(d, row[jp], gen_g[jp]); // hals.jl:106
            if (tid == 0) {
                Wt[(size_t)j * Np + n] = w_new;
                Wn[((size_t)l * Np + n) * K32 + k] = w_new;
            }
            __syncthreads();
        }
}

// H sweep of ONE row k for any L: one wave walks the row column by column (hals.jl:121-148); the pending values of the
// columns [t, t + L) live in an LDS ring of M = roundup(L, 64) + 64 entries that is refilled 64 columns at a time from PT
// (zero-padded behind Tl), H_old comes in blocks of 64 through a register.  A step: x from the pending value, the same-row
// pushes P[k][t + e] += (x - h_old) * taps[e], e = 1 .. Lt - 1 (full-window taps GW[k][k], or the edge column's own GE).
// Writes H (both layouts) and the per-column change D[t]; the cross-row terms follow in hals_h_push_gen_kernel.
// grid 1, block 64, dynamic LDS: (M + L) floats.
static __global__ __launch_bounds__(64) void hals_h_row_gen_kernel(HalsRowParams q)
{
    extern __shared__ float gen_ring[];
    const int lane = threadIdx.x;
    const int L = q.L, E = 2 * L - 1, k = q.k;
    const int M = ((L + 63) / 64) * 64 + 64;
    float *ring = gen_ring, *taps = gen_ring + M;
    float *Prow = q.PT + (size_t)k * q.TPp;
    float *Hrow = q.Ht + (size_t)k * q.TP + q.PADL;
    const float *gk = q.GW + ((size_t)k * q.K32 + k) * E + (L - 1);
    for (int e = lane; e < L; e += 64) taps[e] = gk[e];
    for (int c = lane; c < M; c += 64) ring[c] = Prow[c]; // columns [0, M)
    __syncthreads(); // (one wave: an LDS fence between its lanes)
    const float nrm_full = taps[0];
    for (int tb = 0; tb < q.Tl; tb += 64) {
        if (tb > 0) { // columns [tb + M - 64, tb + M): not yet reached by any same-row push (they reach t + L - 1 < tb + M - 64)
            ring[(tb + M - 64 + lane) % M] = Prow[tb + M - 64 + lane];
            __syncthreads();
        }
        const float hreg = (tb + lane < q.Tl) ? Hrow[tb + lane] : 0.f;
        float hnew = 0.f, dreg = 0.f;
        const int tend = (tb + 64 < q.Tl) ? tb + 64 : q.Tl;
        for (int t = tb; t < tend; ++t) {
            const float s_h = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hreg), t - tb));
            const float s_p = ring[t % M];
            float nrm = nrm_full;
            int Lt = L;
            const float *ge = nullptr;
            if (t >= q.t_edge0) { // truncated window (hals.jl:136): the column's own norm and taps
                ge = q.GE + (((size_t)k * q.ne + (t - q.t_edge0)) * q.K32 + k) * E + (L - 1);
                nrm = ge[0];
                Lt = q.Tl - t;
            }
            const float x = fmaxf((s_h * nrm - s_p - q.l1) / (nrm + CMF_EPS_F + q.l2), 0.f); // hals.jl:152-153
            const float d = x - s_h;
            for (int e = 1 + lane; e < Lt; e += 64) ring[(t + e) % M] = fmaf(d, ge ? ge[e] : taps[e], ring[(t + e) % M]); // hals.jl:146
            __syncthreads(); // the next column's pending value may have been written by another lane
            if (lane == t - tb) { hnew = x; dreg = d; }
        }
        if (tb + lane < q.Tl) {
            Hrow[tb + lane] = hnew;
            q.H[(size_t)(q.PADL + tb + lane) * q.K32 + k] = hnew;
            q.D[tb + lane] = dreg;
        }
    }
}

// Cross-row terms of row k's changes for any L: PT[kp][tp] += sum_e D[tp - e] * taps(tp - e)[k][kp][e], kp > k
// (hals.jl:146 for the other components; source columns in the right edge use their own taps GE).
// grid (ceil(Tl / 256), K - 1 - k), block 256: blockIdx.y -> kp = k + 1 + blockIdx.y
static __global__ __launch_bounds__(256) void hals_h_push_gen_kernel(float *PT, const float *D, const float *GW, const float *GE,
                                                               int k, int Tl, int L, int K32, int TPp, int ne, int t_edge0)
{
    const int tp = blockIdx.x * 256 + threadIdx.x, kp = k + 1 + blockIdx.y;
    if (tp >= Tl) return;
    const int E = 2 * L - 1;
    const float *gw = GW + ((size_t)k * K32 + kp) * E + (L - 1);
    float s = 0.f;
    for (int e = -(L - 1); e <= L - 1; ++e) {
        const int t = tp - e;
        if (t < 0 || t >= Tl) continue;
        const float tap = (t < t_edge0) ? gw[e] : GE[(((size_t)k * ne + (t - t_edge0)) * K32 + kp) * E + (L - 1) + e];
        s = fmaf(D[t], tap, s);
    }
    PT[(size_t)kp * TPp + tp] += s;
}

// =============================================================================================
// PGD rule (src/algs/pgd.jl) on the same contractions:
//   gradW = 2 * H_shift * resid' (+ penalties)   -> hxt on the stored residual      (pgd.jl:206-214)
//   gradH = 2 * transconv(W, resid) (+ penalties) -> transconv on resid^T             (pgd.jl:218-221)
//   x <- proj(x - step / (||grad|| + eps) * grad)                                      (pgd.jl:237-241)
// Pass 1 forms the gradient and its sum of squares, pass 2 applies the step.
// =============================================================================================
__device__ __forceinline__ float cmf_sign(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

// outT[n][PADL + t] = in[PADL + t][n] for t < Tl, all Np columns: the stored residual (est - data, masked, or its sign) in the
// layout tensor_transconv wants.  The H phase of a PGD iteration needs exactly the residual the W phase's closing conv has
// just stored (same W, same H; pgd.jl:245 and :230), so it is transposed (0.8 GB of traffic) instead of convolved again.
// grid (Np / 64, ceil(Tl / 64)), block 256
static __global__ __launch_bounds__(256) void transpose_rows_kernel(const float *in, float *outT, int Tl, int Np, int TP, int PADL)
{
    __shared__ float tile[64][65];
    const int tid = threadIdx.x;
    const int n0 = blockIdx.x * 64, t0 = blockIdx.y * 64;
    {
        const int nn = tid & 63;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int tt = q * 4 + (tid >> 6), t = t0 + tt;
            tile[tt][nn] = (t < Tl) ? in[(size_t)(PADL + t) * Np + n0 + nn] : 0.f;
        }
    }
    __syncthreads();
    {
        const int tt = tid & 63, t = t0 + tt;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int nn = q * 4 + (tid >> 6);
            if (t < Tl) outT[(size_t)(n0 + nn) * TP + PADL + t] = tile[tt][nn];
        }
    }
}

// ... and on a T-shard the right lag halo of that transposed residual (columns Tl .. Tl + halo - 1, which the stored [t][n]
// residual must not hold: the C2 kernel reads the rows behind Tl as zero padding): est there from W and the H halo directly,
// minus the data halo, masked / as a sign like the conv epilogue does it.  halo <= L - 1 columns: 2*K*L FMAs per output.
// grid (Np / 128, halo), block 128
static __global__ __launch_bounds__(128) void resid_halo_kernel(const float *Wt, const float *H, const float *XT, const float *MT, float *outT,
                                                          int Tl, int K, int L, int K32, int Np, int TP, int PADL, int loss_abs)
{
    const int n = blockIdx.x * 128 + threadIdx.x, t = Tl + blockIdx.y;
    float v = 0.f;
    for (int l = 0; l < L; ++l) {
        const float *hrow = H + (size_t)(PADL + t - l) * K32; // (rows in front of the shard are its left halo / zero padding)
        const float *wrow = Wt + (size_t)l * K32 * Np + n;
        for (int k = 0; k < K; ++k) v = fmaf(wrow[(size_t)k * Np], hrow[k], v);
    }
    const size_t at = (size_t)n * TP + PADL + t;
    const float dv = XT[at], m = MT ? MT[at] : 1.f;
    outT[at] = loss_abs ? ((v > dv) ? m : ((v < dv) ? -m : 0.f)) : (v - dv) * m;
}

// grad[idx] = gscale*G + 2*pen_sq*w + pen_abs*sign(w) over the valid entries of the Wt layout (gscale: 2 for SquareLoss,
// whose stored residual is est - data; 1 for AbsoluteLoss, whose stored quantity already is the gradient sign(est - data));
// block partials of sum(g^2).  grid (Np/64, KB, L), block 256
static __global__ __launch_bounds__(256) void pgd_w_grad_kernel(const float *Wt, const float *G, float *grad, double *partial,
                                                          int N, int K, int Np, int K32, float pen_sq, float pen_abs, float gscale)
{
    const int tid = threadIdx.x;
    const int n = blockIdx.x * 64 + (tid & 63), kb = blockIdx.y, l = blockIdx.z;
    double ss = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int k = kb * 32 + q * 4 + (tid >> 6);
        const size_t idx = ((size_t)l * K32 + k) * Np + n;
        float g = 0.f;
        if (k < K && n < N) {
            const float w = Wt[idx];
            g = gscale * G[idx] + 2.f * pen_sq * w + pen_abs * cmf_sign(w);
        }
        grad[idx] = g;
        ss += (double)g * (double)g;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
    __shared__ double red[4];
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    if (tid == 0) partial[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// W <- proj(W - alpha*grad), alpha = step / (sqrt(*sumsq) + eps); refreshes Wn.  grid (Np/64, KB, L), block 256
static __global__ __launch_bounds__(256) void pgd_w_apply_kernel(float *Wt, float *Wn, const float *grad, const double *sumsq,
                                                           int N, int K, int Np, int K32, float step, int nonneg)
{
    __shared__ float tile[32][65];
    const int tid = threadIdx.x;
    const int n0 = blockIdx.x * 64, kb = blockIdx.y, l = blockIdx.z;
    const float alpha = (float)((double)step / (sqrt(*sumsq) + 2.220446049250313e-16));
    {
        const int nn = tid & 63;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int kk = q * 4 + (tid >> 6), k = kb * 32 + kk, n = n0 + nn;
            const size_t idx = ((size_t)l * K32 + k) * Np + n;
            float w = 0.f;
            if (k < K && n < N) {
                w = Wt[idx] - alpha * grad[idx];
                if (nonneg) w = fmaxf(CMF_EPS_F, w);
            }
            Wt[idx] = w;
            tile[kk][nn] = w;
        }
    }
    __syncthreads();
    {
        const int kk = tid & 31;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int nn = q * 8 + (tid >> 5);
            Wn[((size_t)l * Np + n0 + nn) * K32 + kb * 32 + kk] = tile[kk][nn];
        }
    }
}

// grad[t][k] = gscale*sum_s slabs[s][t][k] + 2*pen_sq*h + pen_abs*sign(h); grid (ceil(Tl/64), KB), block 256
static __global__ __launch_bounds__(256) void pgd_h_grad_kernel(const float *H, const float *slabs, int S, float *grad, double *partial,
                                                          int Tl, int K, int K32, int PADL, float pen_sq, float pen_abs, float gscale)
{
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * 64, kb = blockIdx.y;
    const size_t TK = (size_t)Tl * K32;
    const int kk = tid & 31, k = kb * 32 + kk;
    double ss = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int t = t0 + q * 8 + (tid >> 5);
        if (t < Tl) {
            const size_t idx = (size_t)t * K32 + k;
            float g = 0.f;
            if (k < K) {
                float v = 0.f;
                for (int s = 0; s < S; ++s) v += slabs[(size_t)s * TK + idx];
                const float hv = H[(size_t)(PADL + t) * K32 + k];
                g = gscale * v + 2.f * pen_sq * hv + pen_abs * cmf_sign(hv);
            }
            grad[idx] = g;
            ss += (double)g * (double)g;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
    __shared__ double red[4];
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    if (tid == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// H <- proj(H - alpha*grad); refreshes Ht.  grid (ceil(Tl/64), KB), block 256
static __global__ __launch_bounds__(256) void pgd_h_apply_kernel(float *H, float *Ht, const float *grad, const double *sumsq,
                                                           int Tl, int K, int K32, int PADL, int TP, float step, int nonneg)
{
    __shared__ float tile[32][65];
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * 64, kb = blockIdx.y;
    const float alpha = (float)((double)step / (sqrt(*sumsq) + 2.220446049250313e-16));
    {
        const int kk = tid & 31, k = kb * 32 + kk;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int tt = q * 8 + (tid >> 5), t = t0 + tt;
            float hv = 0.f;
            if (t < Tl) {
                const size_t hidx = (size_t)(PADL + t) * K32 + k;
                if (k < K) {
                    hv = H[hidx] - alpha * grad[(size_t)t * K32 + k];
                    if (nonneg) hv = fmaxf(CMF_EPS_F, hv);
                }
                H[hidx] = hv;
            }
            tile[kk][tt] = hv;
        }
    }
    __syncthreads();
    {
        const int tt = tid & 63;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int kk = q * 4 + (tid >> 6), t = t0 + tt;
            if (t < Tl) Ht[(size_t)(kb * 32 + kk) * TP + PADL + t] = tile[kk][tt];
        }
    }
}


// UnitNormConstraint (pgd.jl:100-110): every component k whose slice (W[k, :, :] or H[k, :]) has norm > 1 is scaled to
// norm 1.  Pass 1: ss[k] = sum of squares of the slice (one block per k, fp64); pass 2: scale both layouts.
static __global__ __launch_bounds__(256) void pgd_w_knorm_kernel(const float *Wt, double *ss, int N, int L, int Np, int K32)
{
    __shared__ double red[256];
    const int k = blockIdx.x;
    double s = 0.0;
    for (int idx = threadIdx.x; idx < L * N; idx += 256) {
        const int l = idx / N, n = idx - l * N;
        const double v = Wt[((size_t)l * K32 + k) * Np + n];
        s += v * v;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) ss[k] = red[0];
}
static __global__ __launch_bounds__(256) void pgd_h_knorm_kernel(const float *Ht, double *ss, int Tl, int TP, int PADL)
{
    __shared__ double red[256];
    const int k = blockIdx.x;
    double s = 0.0;
    for (int t = threadIdx.x; t < Tl; t += 256) {
        const double v = Ht[(size_t)k * TP + PADL + t];
        s += v * v;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) ss[k] = red[0];
}
__device__ __forceinline__ float cmf_unit_scale(double ss)
{
    const double mag = sqrt(ss);
    return mag > 1.0 ? (float)(1.0 / mag) : 1.f;
}
static __global__ void pgd_w_kscale_kernel(float *Wt, float *Wn, const double *ss, int N, int K, int L, int Np, int K32)
{
    const size_t total = (size_t)L * K * N;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx % N), k = (int)((idx / N) % K), l = (int)(idx / ((size_t)N * K));
        const float sc = cmf_unit_scale(ss[k]);
        if (sc != 1.f) {
            const size_t it = ((size_t)l * K32 + k) * Np + n;
            const float v = Wt[it] * sc;
            Wt[it] = v;
            Wn[((size_t)l * Np + n) * K32 + k] = v;
        }
    }
}
static __global__ void pgd_h_kscale_kernel(float *H, float *Ht, const double *ss, int Tl, int K, int K32, int TP, int PADL)
{
    const size_t total = (size_t)Tl * K;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(idx % K), t = (int)(idx / K);
        const float sc = cmf_unit_scale(ss[k]);
        if (sc != 1.f) {
            const size_t ih = (size_t)(PADL + t) * K32 + k;
            const float v = H[ih] * sc;
            H[ih] = v;
            Ht[(size_t)k * TP + PADL + t] = v;
        }
    }
}


// =============================================================================================
// Optional Gram form of the MU denominators ("optional algebraic shortcut").
//   denomW[(l,k)][n] = sum_{(l',k')} HH[(l,k)][(l',k')] * W[(l',k')][n],  HH = H_unfold H_unfold'
//   denomH[t][k]     = sum_{k',e} taps(t)[k][k'][e] * H[t+e][k'],          taps = lag-Gram of W (as in HALS)
// Both are exact rewritings of H_shift * est' and tensor_transconv(W, est) with est = tensor_conv(W, H)
// (including the edge truncations); only the association of the sums differs.  Off by default.
// =============================================================================================

// out[p][n] = sum_{p'} HH[p'][p] * Wt[p'][n]  (HH is symmetric: row p' is read, columns p0+i -> coalesced)
// grid (Np/32, LK/(32*MB)), block 256: ONE tile of MB x 1 blocks of 32 x 32 (MB*32 rows p, 32 columns n) per workgroup; its
// four waves each take a quarter of the reduction range p' for all MB blocks (one B row of Wt and MB A rows of HH feed MB
// MFMAs: 1 + 1/MB 128-byte operand rows from L2 per MFMA), 8 row pairs in flight under the MFMAs of the batch before (two
// register sets), and add their partial sums through LDS in wave order (deterministic) behind ONE barrier.  MB is chosen
// by the launcher so that the grid is one workgroup per CU where the shape allows it (config 2: LK = 640, Np = 2048, MB = 5:
// exactly 256 workgroups of 4 x 400 MFMAs -- the 64 x 64 tiles of rounds 3-4 were 320 workgroups on 256 CUs, 38 us for
// 10.7 us of MFMA work).  Per output element the terms are added in the same order as in that kernel: bitwise the same sums.
// History: one wave per 32 x 32 block with a load, wait, MFMA loop was a chain of LK / 2 L2 round trips (99 us at LK = 640,
// Np = 2048, independent of T -- a fifth of a T/8 shard's Gram-form iteration); a 32 x 32 block per workgroup with the
// reduction split was L2-bandwidth bound (two operand rows per MFMA, 210 MB per launch: 36 us).
// dynamic LDS: 4 * MB * 16 * 64 floats.
template <int MB>
__global__ __launch_bounds__(256) void gram_w_kernel(const float *HH, const float *Wt, float *out, int LK, int NpH, int Np)
{
    extern __shared__ __attribute__((aligned(16))) float gw_part[]; // [4][MB][16][64]
    // (the wave index as a SCALAR: with a divergent-looking `wave` every guarded load below became its own exec-masked basic
    // block with a conservative wait behind it -- 26 us for 10.7 us of MFMA work)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int p0 = blockIdx.y * 32 * MB, n0 = blockIdx.x * 32;
    f32x16 acc[MB];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    // (buffer loads with scalar row offsets: no 64-bit address arithmetic per load)
    const __amdgpu_buffer_rsrc_t ar = cmf_rsrc(HH, (size_t)LK * NpH * 4);
    const __amdgpu_buffer_rsrc_t br = cmf_rsrc(Wt, (size_t)LK * Np * 4);
    const int aoff = (h * NpH + p0 + i) * 4, boff = (h * Np + n0 + i) * 4;
    const int q4 = (LK / 4 + 1) & ~1;            // this wave's share of the reduction rows (even)
    const int lo = wave * q4, hi = (lo + q4 < LK) ? lo + q4 : LK;
    constexpr int NB = 8;                        // row pairs per batch
    float av[2][NB][MB], bv[2][NB];
    auto load = [&](int s, int pp0) {
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int pp = pp0 + 2 * q;
            const bool ok = pp < hi;
            const int ppc = ok ? pp : lo; // unconditional loads (a row pair this wave owns), a zero B operand behind the range
            const float bx = cmf_bload(br, boff, ppc * Np * 4);
            bv[s][q] = ok ? bx : 0.f;
#pragma unroll
            for (int a = 0; a < MB; ++a) av[s][q][a] = cmf_bload(ar, aoff + 128 * a, ppc * NpH * 4);
        }
    };
    auto mac = [&](int s) {
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
            for (int a = 0; a < MB; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][q][a], bv[s][q], acc[a], 0, 0, 0);
    };
    if (lo < hi) load(0, lo);
    for (int pp0 = lo; pp0 < hi; pp0 += 4 * NB) {
        if (pp0 + 2 * NB < hi) load(1, pp0 + 2 * NB);
        mac(0);
        if (pp0 + 2 * NB < hi) {
            if (pp0 + 4 * NB < hi) load(0, pp0 + 4 * NB);
            mac(1);
        }
    }
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) gw_part[((wave * MB + a) * 16 + r) * 64 + lane] = acc[a][r];
    __syncthreads();
    for (int o = threadIdx.x; o < MB * 16 * 64; o += 256) {
        const int a = o >> 10, r = (o >> 6) & 15, ln = o & 63;
        const int e = (a * 16 + r) * 64 + ln;
        const float v = ((gw_part[e] + gw_part[MB * 1024 + e]) + gw_part[2 * MB * 1024 + e]) + gw_part[3 * MB * 1024 + e];
        out[(size_t)(p0 + 32 * a + cmf_crow(r, ln >> 5)) * Np + n0 + (ln & 31)] = v;
    }
}

// out[t][k] = sum_{k',e} taps(t)[k][k'][e] * Ht[k'][PADL + t + e];  taps = GW (full window) or GE (edge columns)
// grid (ceil(Tl/64), K32/4), block 256: wave w -> k = blockIdx.y*4 + w, lane -> t = t0 + lane
// dynamic LDS: K32 * (64 + 2*(L-1)) floats
static __global__ __launch_bounds__(256) void gram_h_kernel(const float *Ht, const float *GW, const float *GE, float *out,
                                                      int Tl, int K, int L, int K32, int TP, int PADL, int ne, int t_edge0, int block0)
{
    extern __shared__ __attribute__((aligned(16))) float smem_dyn[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = (blockIdx.x + block0) * 64; // block0: the 64-column blocks in front belong to gram_h_mfma_kernel
    const int k = blockIdx.y * 4 + wave;
    const int E = 2 * L - 1, WN = 64 + 2 * (L - 1);
    for (int idx = tid; idx < K32 * WN; idx += 256) {
        const int kp = idx / WN, c = idx - kp * WN;
        smem_dyn[idx] = Ht[(size_t)kp * TP + PADL + t0 - (L - 1) + c];
    }
    __syncthreads();
    const int t = t0 + lane;
    float acc = 0.f;
    if (k < K) {
        const bool full = (t0 + 64 <= t_edge0); // block-uniform: every column has the full lag window
        if (full) {
            const float *gw = GW + (size_t)k * K32 * E; // wave-uniform -> scalar loads
            for (int kp = 0; kp < K; ++kp) {
                const float *hw = smem_dyn + kp * WN + lane;
#pragma unroll 4
                for (int ei = 0; ei < E; ++ei) acc = fmaf(gw[kp * E + ei], hw[ei], acc);
            }
        } else if (t < Tl) {
            const float *taps = (t < t_edge0) ? GW + (size_t)k * K32 * E
                                               : GE + ((size_t)k * ne + (t - t_edge0)) * K32 * E;
            for (int kp = 0; kp < K; ++kp) {
                const float *hw = smem_dyn + kp * WN + lane;
#pragma unroll 8
                for (int ei = 0; ei < E; ++ei) acc = fmaf(taps[kp * E + ei], hw[ei], acc); // (per-lane tap loads: several in flight)
            }
        }
    }
    if (t < Tl && k < K32) out[(size_t)t * K32 + k] = acc;
}

// The same product on the MFMA pipe for the columns whose lag window is full (all but the last L-1):
//   out[t][k] = sum_{(k', e)} A[t][(k', e)] * B[(k', e)][k],   A = H[k'][t - (L-1) + e] (windows of the staged H rows),
//   B = the full-window taps in the layout GWt[k'][e][k] (k fastest, e padded to an even count with a zero tap).
// One wave = 32 columns t x 32 outputs k, K32 * Ep / 2 MFMAs; A comes from LDS (lane i: window column i + e), B streams
// from L2 in 128-byte rows, one k' ahead (two register sets).  The scalar kernel above spends an LDS read and a scalar
// load per FMA and restages the window for every four outputs: 317 us at config 5 against ~60 here.
// grid (tiles of 128 columns that end at or before t_edge0, KB), block 256 (4 waves = 4 x 32 columns);
// dynamic LDS: K32 * (128 + 2*(L-1)) floats.
// ... and for the few columns the tiles above leave over (the right edge with its per-column taps GE, and what does not
// fill a tile of 128): one wave per output (t, k), its lanes over the K * E terms -- the taps of a column are K32 * E
// contiguous floats -- and a DPP wave sum.  (The scalar kernel's edge path walks those terms one load at a time: 76 us
// for 80 columns.)   grid (columns from t_first on, K32 / 4), block 256: wave w -> k = blockIdx.y * 4 + w
struct GramEdge { // the edge workgroups that ride at the end of gram_h_mfma_kernel's grid (n_main = its own workgroups)
    const float *GW, *GE;
    int Tl, ne, t_edge0, t_first, n_main, kq; // kq = K32 / 4 workgroups per column
};
__device__ __forceinline__ void gram_h_edge(const float *Ht, const float *GW, const float *GE, float *out,
                                            int Tl, int K, int L, int K32, int TP, int PADL, int ne, int t_edge0, int t, int kblk)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = kblk * 4 + wave;
    if (t >= Tl) return;
    const int E = 2 * L - 1;
    float x = 0.f;
    if (k < K) {
        const float *taps = (t < t_edge0) ? GW + (size_t)k * K32 * E : GE + ((size_t)k * ne + (t - t_edge0)) * K32 * E;
        const float *hw = Ht + PADL + t - (L - 1);
        for (int idx = lane; idx < K * E; idx += 64) {
            const int kp = idx / E, ei = idx - kp * E;
            x = fmaf(taps[idx], hw[(size_t)kp * TP + ei], x);
        }
    }
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x111, 0xf, 0xf, false)); // row_shr:1
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x112, 0xf, 0xf, false)); // row_shr:2
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x114, 0xf, 0xf, false)); // row_shr:4
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x118, 0xf, 0xf, false)); // row_shr:8
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x142, 0xa, 0xf, false)); // row_bcast:15
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x143, 0xc, 0xf, false)); // row_bcast:31
    if (lane == 63 && k < K32) out[(size_t)t * K32 + k] = x;
}
// fw (1, 2 or 4): the F = K * Ep / 2 MFMAs of an output block are split over fw waves (partial sums added through LDS in
// wave order), the workgroup's tile is then 128 / fw columns: a short shard (T/8 = 6250 columns: 48 tiles of 128) gets
// fw times the waves, each with 1/fw of the dependent MFMA chain and of the tap stream (62 -> ~15 us at T = 6250).
// dynamic LDS: K32 * (128 / fw + 2*(L-1)) floats for the H window (+ 4096 floats for the partial sums when fw > 1).
// grid: n_tiles * KB workgroups of the MFMA form (tile fastest) + the edge workgroups behind them (GramEdge: one launch and
// one kernel boundary less, and the latency-bound edge waves run beside the MFMA tiles instead of after them).
static __global__ __launch_bounds__(256) void gram_h_mfma_kernel(const float *Ht, const float *GWt, float *out, int K, int L, int K32, int TP, int PADL, int Ep,
                                                           int fw, int n_tiles, GramEdge edge)
{
    extern __shared__ __attribute__((aligned(16))) float smem_dyn[];
    if ((int)blockIdx.x >= edge.n_main) {
        const int b = blockIdx.x - edge.n_main;
        gram_h_edge(Ht, edge.GW, edge.GE, out, edge.Tl, K, L, K32, TP, PADL, edge.ne, edge.t_edge0, edge.t_first + b / edge.kq, b % edge.kq);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // (scalar: see gram_w_kernel)
    const int i = lane & 31, hh = lane >> 5;
    const int cw = 4 / fw, tile = 32 * cw;
    const int cg = wave % cw, fg = wave / cw; // column group and F group of this wave
    const int t0 = ((int)blockIdx.x % n_tiles) * tile, kbo = (int)blockIdx.x / n_tiles;
    const int WN = tile + 2 * (L - 1);
    for (int base = 0; base < K32 * WN; base += 4 * 256) { // four loads in flight per thread, then the LDS writes
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * 256 + tid;
            const int kp = idx / WN, c = idx - kp * WN;
            v[u] = idx < K32 * WN ? Ht[(size_t)kp * TP + PADL + t0 - (L - 1) + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * 256 + tid;
            if (idx < K32 * WN) smem_dyn[idx] = v[u];
        }
    }
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // The K * Ep / 2 MFMAs are numbered f = kp * (Ep / 2) + s; their B rows lie 2 * K32 floats apart in f (Ep is even), so
    // the B stream is one linear walk: 32 rows per chunk, the next chunk's loads issued before this chunk's MFMAs (one
    // wave or two per SIMD: an L2 round trip must be covered from inside the wave).
    const int nstep = Ep >> 1, F = K * nstep;
    const int Fq = ((F + fw - 1) / fw + 31) & ~31; // this wave's share, in whole chunks
    const int f_lo = fg * Fq, f_hi = (f_lo + Fq < F) ? f_lo + Fq : F;
    const float *arow = smem_dyn + cg * 32 + i + hh;                 // + kp * WN + 2 * s
    const __amdgpu_buffer_rsrc_t brr = cmf_rsrc(GWt, (size_t)K32 * Ep * K32 * 4); // B rows: + 2 * f * K32 floats (scalar offsets)
    const int broff = (hh * K32 + kbo * 32 + i) * 4;
    float bb[2][32];
    auto loadb = [&](float (&x)[32], int f0) {
#pragma unroll
        for (int q = 0; q < 32; ++q) x[q] = cmf_bload(brr, broff, 2 * ((f0 + q < f_hi) ? f0 + q : f_lo) * K32 * 4); // (unconditional; A is zero behind the range)
    };
    auto mac = [&](const float (&x)[32], int f0) {
        int kp = f0 / nstep, s2 = f0 - kp * nstep;
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const float a = (f0 + q < f_hi) ? arow[(size_t)kp * WN + 2 * s2] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, x[q], acc, 0, 0, 0);
            if (++s2 == nstep) { s2 = 0; ++kp; }
        }
    };
    if (f_lo < f_hi) loadb(bb[0], f_lo);
    for (int f0 = f_lo; f0 < f_hi; f0 += 64) {
        if (f0 + 32 < f_hi) loadb(bb[1], f0 + 32);
        mac(bb[0], f0);
        if (f0 + 32 < f_hi) {
            if (f0 + 64 < f_hi) loadb(bb[0], f0 + 64);
            mac(bb[1], f0 + 32);
        }
    }
    if (fw > 1) { // add the F groups' partial sums in group order (deterministic)
        float *part = smem_dyn + (size_t)K32 * WN; // [fw][cw][16][64]
#pragma unroll
        for (int r = 0; r < 16; ++r) part[((fg * cw + cg) * 16 + r) * 64 + lane] = acc[r];
        __syncthreads();
        if (fg != 0) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[r];
            for (int g2 = 1; g2 < fw; ++g2) v += part[((g2 * cw + cg) * 16 + r) * 64 + lane];
            acc[r] = v;
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int t = t0 + cg * 32 + cmf_crow(r, hh);
        out[(size_t)t * K32 + kbo * 32 + i] = acc[r];
    }
}

// partial[b] = sum H*den, partial[nb + b] = sum H*num over the block's (t, k); grid ceil(Tl*K32/1024), block 256
static __global__ __launch_bounds__(256) void gram_dot_kernel(const float *H, const float *num, const float *den, double *partial,
                                                        int Tl, int K, int K32, int PADL, int nb)
{
    const size_t total = (size_t)Tl * K32;
    double sd = 0.0, sn = 0.0;
    for (size_t idx = (size_t)blockIdx.x * 1024 + threadIdx.x; idx < (size_t)(blockIdx.x + 1) * 1024 && idx < total; idx += 256) {
        const int k = (int)(idx % K32);
        const size_t t = idx / K32;
        if (k < K) {
            const double hv = H[(PADL + t) * K32 + k];
            sd += hv * (double)den[idx];
            sn += hv * (double)num[idx];
        }
    }
    __shared__ double red[2][256];
    red[0][threadIdx.x] = sd;
    red[1][threadIdx.x] = sn;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            red[0][threadIdx.x] += red[0][threadIdx.x + off];
            red[1][threadIdx.x] += red[1][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = red[0][0];
        partial[nb + blockIdx.x] = red[1][0];
    }
}

// ---------------------------------------------------------------------------------------------
// Layout conversion (fp64 Julia order on the host side <-> padded fp32 device layouts)
// ---------------------------------------------------------------------------------------------
// in: cols [tc, tc+ncols) of an N x * column-major fp64 matrix (in[n + N*(t - tc)]).
// Writes X[(PADL+t)][n] and XT[n][PADL+t].  grid (ceil(N/32), ceil(ncols/32)), block (32, 8)
static __global__ void pack_cols_kernel(const double *in, int N, int tc, int ncols, float *X, float *XT, int Np, int TP, int PADL)
{
    __shared__ float tile[32][33];
    const int n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    for (int q = threadIdx.y; q < 32; q += 8) {
        int c = c0 + q, n = n0 + threadIdx.x;
        float v = 0.f;
        if (c < ncols && n < N) {
            v = (float)in[(size_t)c * N + n];
            if (X) X[(size_t)(PADL + tc + c) * Np + n] = v;
        }
        tile[q][threadIdx.x] = v;
    }
    __syncthreads();
    if (XT) {
        for (int q = threadIdx.y; q < 32; q += 8) {
            int n = n0 + q, c = c0 + threadIdx.x;
            if (c < ncols && n < N) XT[(size_t)n * TP + PADL + tc + c] = tile[threadIdx.x][q];
        }
    }
}

// W fp64 [L][N][K] (k fastest) -> Wt[l][k][n], Wn[l][n][k]
static __global__ void pack_W_kernel(const double *in, int N, int K, int L, float *Wt, float *Wn, int Np, int K32)
{
    size_t total = (size_t)L * N * K;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        int k = (int)(idx % K);
        size_t r = idx / K;
        int n = (int)(r % N), l = (int)(r / N);
        float v = (float)in[idx];
        Wt[((size_t)l * K32 + k) * Np + n] = v;
        Wn[((size_t)l * Np + n) * K32 + k] = v;
    }
}
template <typename O>
__global__ void unpack_W_kernel(O *out, int N, int K, int L, const float *Wn, int Np, int K32)
{
    size_t total = (size_t)L * N * K;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        int k = (int)(idx % K);
        size_t r = idx / K;
        int n = (int)(r % N), l = (int)(r / N);
        out[idx] = (O)Wn[((size_t)l * Np + n) * K32 + k];
    }
}
// H fp64 [T][K] (k fastest) -> H[(PADL+t)][k], Ht[k][PADL+t]
static __global__ void pack_H_kernel(const double *in, int Tl, int K, float *H, float *Ht, int K32, int TP, int PADL)
{
    size_t total = (size_t)Tl * K;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        int k = (int)(idx % K);
        int t = (int)(idx / K);
        float v = (float)in[idx];
        H[(size_t)(PADL + t) * K32 + k] = v;
        Ht[(size_t)k * TP + PADL + t] = v;
    }
}
template <typename O>
__global__ void unpack_H_kernel(O *out, int Tl, int K, const float *H, int K32, int PADL)
{
    size_t total = (size_t)Tl * K;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        int k = (int)(idx % K);
        int t = (int)(idx / K);
        out[idx] = (O)H[(size_t)(PADL + t) * K32 + k];
    }
}
// rows [tc, tc+ncols) of a [*][stride] fp32 row-major buffer (row offset row0) -> fp64 out[c*width + j]
static __global__ void unpack_rows_kernel(double *out, const float *in, int row0, int tc, int ncols, int width, int stride)
{
    size_t total = (size_t)ncols * width;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        int j = (int)(idx % width);
        int c = (int)(idx / width);
        out[idx] = (double)in[(size_t)(row0 + tc + c) * stride + j];
    }
}

// H halo rows <-> contiguous staging buffers ((L-1) x K32 floats)
// dir 0: buf <- H rows [r0, r0+rows)   dir 1: H rows <- buf (and Ht columns)
static __global__ void halo_copy_kernel(float *H, float *Ht, float *buf, int r0, int rows, int K32, int TP, int dir)
{
    int total = rows * K32;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        int k = idx % K32, r = idx / K32;
        if (dir == 0) buf[idx] = H[(size_t)(r0 + r) * K32 + k];
        else {
            float v = buf[idx];
            H[(size_t)(r0 + r) * K32 + k] = v;
            Ht[(size_t)k * TP + r0 + r] = v;
        }
    }
}

// init_rand's scale factor (model.jl:120): per-block partial sums of <data, est> and <est, est> over the padded layouts
// (padding is zero in both).  partial[b] = dot, partial[gridDim.x + b] = norm^2; fp64 accumulation of fp32 products.
static __global__ __launch_bounds__(256) void init_dot_kernel(const float *est, const float *X, size_t n4, double *partial)
{
    __shared__ double red[2][256];
    double sd = 0.0, sn = 0.0;
    for (size_t idx = blockIdx.x * (size_t)256 + threadIdx.x; idx < n4; idx += (size_t)gridDim.x * 256) {
        const float4 e = reinterpret_cast<const float4 *>(est)[idx];
        const float4 x = reinterpret_cast<const float4 *>(X)[idx];
        sd += (double)e.x * x.x + (double)e.y * x.y + (double)e.z * x.z + (double)e.w * x.w;
        sn += (double)e.x * e.x + (double)e.y * e.y + (double)e.z * e.z + (double)e.w * e.w;
    }
    red[0][threadIdx.x] = sd;
    red[1][threadIdx.x] = sn;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            red[0][threadIdx.x] += red[0][threadIdx.x + off];
            red[1][threadIdx.x] += red[1][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = red[0][0];
        partial[gridDim.x + blockIdx.x] = red[1][0];
    }
}

// sum of squares of a fp64 array -> *out (one block; used for data_norm on a staged chunk)
static __global__ __launch_bounds__(256) void sumsq_f64_kernel(const double *in, size_t n, double *out_accum)
{
    __shared__ double red[256];
    double s = 0.0;
    for (size_t idx = blockIdx.x * (size_t)256 + threadIdx.x; idx < n; idx += (size_t)gridDim.x * 256) s += in[idx] * in[idx];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out_accum[blockIdx.x] = red[0];
}

// =============================================================================================
// T-sharded groups: small kernels around the collectives
// =============================================================================================
// Both H halo send blocks in one launch: buf = [own first `rows` rows | own last `rows` rows] of H
static __global__ void halo_pack2_kernel(const float *H, float *buf, int r_first, int r_last, int rows, int K32)
{
    const int total = rows * K32;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < 2 * total; idx += gridDim.x * blockDim.x) {
        const int w = idx >= total, j = idx - w * total;
        buf[idx] = H[(size_t)(w ? r_last : r_first) * K32 + j];
    }
}

// Both receive halos in one launch: left (rows [r_left, r_left+rows)) from `left`, right from `right`; a NULL
// source (no neighbour: the global edge) leaves the zeros in place.  Writes H and its transposed copy Ht.
static __global__ void halo_unpack2_kernel(float *H, float *Ht, const float *left, const float *right, int r_left, int r_right,
                                    int rows, int K32, int TP)
{
    const int total = rows * K32;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < 2 * total; idx += gridDim.x * blockDim.x) {
        const int w = idx >= total, j = idx - w * total;
        const float *src = w ? right : left;
        if (!src) continue;
        const int k = j % K32, r = (w ? r_right : r_left) + j / K32;
        const float v = src[j];
        H[(size_t)r * K32 + k] = v;
        Ht[(size_t)k * TP + r] = v;
    }
}

// The halo of H in the tail of the W-phase all-reduce (cmf_groups.hip, "halo in the all-reduce"): every rank writes, into its OWN
// slot of `slots` ([nranks][3 * rows * K32]), its last 2*rows columns of H followed by its first `rows` columns, and zeros into every
// other rank's slot -- the sum over the ranks is then every rank's columns, exactly (x + 0 + ...).
static __global__ void halo_pack3_kernel(const float *H, float *slots, int r_own0, int Tl, int rows, int K32, int rank, int nranks)
{
    const int per = 3 * rows * K32;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < nranks * per; idx += gridDim.x * blockDim.x) {
        const int r = idx / per, j = idx - r * per;
        float v = 0.f;
        if (r == rank) {
            const int c = j / K32, k = j - c * K32; // columns 0 .. 2 rows - 1: the last 2 rows own columns; then the first `rows`
            const int t = c < 2 * rows ? Tl - 2 * rows + c : c - 2 * rows;
            v = (t >= 0 && t < Tl) ? H[(size_t)(r_own0 + t) * K32 + k] : 0.f;
        }
        slots[idx] = v;
    }
}
// ... and back: H[-2 rows, 0) from the left neighbour's last columns, H[Tl, Tl + rows) from the right neighbour's first ones (a rank
// without that neighbour -- the global edge -- keeps its zeros); H and its transposed copy.
static __global__ void halo_unpack3_kernel(float *H, float *Ht, const float *slots, int r_own0, int Tl, int rows, int K32, int TP, int rank, int nranks)
{
    const int per = 3 * rows * K32;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < per; idx += gridDim.x * blockDim.x) {
        const int c = idx / K32, k = idx - c * K32;
        int src_rank, r;
        if (c < 2 * rows) { src_rank = rank - 1; r = r_own0 - 2 * rows + c; }
        else { src_rank = rank + 1; r = r_own0 + Tl + (c - 2 * rows); }
        if (src_rank < 0 || src_rank >= nranks) continue;
        const float v = slots[(size_t)src_rank * per + idx];
        H[(size_t)r * K32 + k] = v;
        Ht[(size_t)k * TP + r] = v;
    }
}

// loss_reduce_kernel for a shard of a group: *out = sum(partial[0..n)) as before, and the sum is also posted in the
// tail of the [numW | denomW] all-reduce buffer as two floats (hi, lo: hi + lo reproduces the double to ~2^-48) in
// this rank's own slots, zeros in every other rank's slots -- the sum over ranks of the tail is then exact (x + 0 + ...)
// and every rank can add the per-rank doubles in rank order after the next all-reduce (the loss scalar rides on the
// single bulk collective).
static __global__ __launch_bounds__(256) void loss_tail_kernel(const double *partial, int n, double *out, float *tail, int tail_len, int rank)
{
    __shared__ double red[256];
    red[threadIdx.x] = cmf_thread_loss_sum(partial, n);
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    const double tot = red[0];
    if (threadIdx.x == 0) *out = tot;
    const float hi = (float)tot, lo = (float)(tot - (double)hi);
    for (int j = threadIdx.x; j < tail_len; j += 256) tail[j] = (j == 2 * rank) ? hi : (j == 2 * rank + 1) ? lo : 0.f;
}

// Loopback transport (several shards of one group living on ONE device, e.g. a middle-rank shard test on a one-GPU
// box): the collectives are plain kernels over the shards' buffers.  Sums are taken in rank order.
#define CMF_MAX_LOCAL 16
struct CmfPtrTable { float *p[CMF_MAX_LOCAL]; };
static __global__ __launch_bounds__(256) void loopback_allreduce_kernel(CmfPtrTable bufs, int R, size_t count)
{
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < count; idx += (size_t)gridDim.x * blockDim.x) {
        float s = bufs.p[0][idx];
        for (int r = 1; r < R; ++r) s += bufs.p[r][idx];
        for (int r = 0; r < R; ++r) bufs.p[r][idx] = s;
    }
}
static __global__ __launch_bounds__(256) void loopback_allgather_kernel(CmfPtrTable send, CmfPtrTable recv, int R, int count)
{
    const int total = R * count;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const float v = send.p[idx / count][idx % count];
        for (int r = 0; r < R; ++r) recv.p[r][idx] = v;
    }
}

// "Peer" transport (all shards of a one-process group, each on its own stream, reading and writing each other's buffers
// directly: over xGMI with peer access between distinct devices, trivially on one device).  Visibility is by kernel
// boundaries only -- the host fences every stream against every other one with events around these kernels -- never by
// flags inside a kernel.  All-reduce as ONE kernel per shard: shard `me` owns the `me`-th of R equal slices, sums it over
// the R buffers in rank order (bitwise the loopback kernel's sum) and stores the result into ALL R buffers: a
// reduce-scatter by reading the peers and an all-gather by writing them, 1/R of the payload per link and direction
// (1.3 MB of the 10.5 MB at R = 8) where a ring moves (R-1)/R of it through every link, twice.  No two shards touch the
// same slice, so the R kernels need no order among themselves.
static __global__ __launch_bounds__(256) void peer_allreduce_kernel(CmfPtrTable bufs, int R, int me, size_t count, size_t per, int vec4)
{
    const size_t lo = (size_t)me * per, hi = lo + per < count ? lo + per : count;
    if (lo >= hi) return;
    if (vec4) { // every buffer 16-byte aligned, per a multiple of 4: the slice in float4 words + a scalar rest at the end of the payload
        const size_t n4 = (hi - lo) / 4;
        for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < n4; q += (size_t)gridDim.x * blockDim.x) {
            float4 v[CMF_MAX_LOCAL];
#pragma unroll
            for (int r = 0; r < CMF_MAX_LOCAL; ++r)
                if (r < R) v[r] = reinterpret_cast<const float4 *>(bufs.p[r] + lo)[q];
            float4 s = v[0];
#pragma unroll
            for (int r = 1; r < CMF_MAX_LOCAL; ++r)
                if (r < R) { s.x += v[r].x; s.y += v[r].y; s.z += v[r].z; s.w += v[r].w; }
#pragma unroll
            for (int r = 0; r < CMF_MAX_LOCAL; ++r)
                if (r < R) reinterpret_cast<float4 *>(bufs.p[r] + lo)[q] = s;
        }
        const size_t done = lo + 4 * n4;
        for (size_t idx = done + blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < hi; idx += (size_t)gridDim.x * blockDim.x) {
            float s = bufs.p[0][idx];
            for (int r = 1; r < R; ++r) s += bufs.p[r][idx];
            for (int r = 0; r < R; ++r) bufs.p[r][idx] = s;
        }
        return;
    }
    for (size_t idx = lo + blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < hi; idx += (size_t)gridDim.x * blockDim.x) {
        float s = bufs.p[0][idx];
        for (int r = 1; r < R; ++r) s += bufs.p[r][idx];
        for (int r = 0; r < R; ++r) bufs.p[r][idx] = s;
    }
}
// all-gather of the peer transport: shard `me` pulls every rank's send block into its own receive buffer
static __global__ __launch_bounds__(256) void peer_allgather_kernel(CmfPtrTable send, float *recv, int R, int count)
{
    const int total = R * count;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x)
        recv[idx] = send.p[idx / count][idx % count];
}

#include "cmf_small_k.h"
