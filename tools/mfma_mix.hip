// Microbenchmark (not part of the library): MFMA issue rate of one wave's instruction stream as the conv main loop's
// ingredients are added -- v_mfma_f32_32x32x2_f32 alone, + the LDS operand reads, + the W buffer loads, + the real
// conv2_lag / conv2_load_w code -- at 1, 2 and 3 waves per SIMD.  Prints cycles per MFMA (64 = the pipe's rate).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -Icmf.jl_amd/csrc tools/mfma_mix.hip -o tools/bin/mfma_mix
#include <hip/hip_runtime.h>
#include "cmf_kernels.h"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// V: 0 MFMA only; 1 + ds_read pairs (software pipelined like conv2_lag); 2 + 32 buffer loads per 64 MFMAs; 3 real code
template <int V>
__global__ __launch_bounds__(64, 3) void mix_kernel(const float *W, float *out, unsigned long long *cyc, int iters, int Np)
{
    __shared__ float Hs[32 * CONV3_STRIDE];
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    for (int q = lane; q < 32 * CONV3_STRIDE; q += 64) Hs[q] = 0.001f * (float)(q & 255);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int rowbytes = Np * 4, lagbytes = 32 * Np * 4;
    const __amdgpu_buffer_rsrc_t wr = cmf_rsrc(W, (size_t)20 * lagbytes);
    const int woff = (h * Np + (blockIdx.x % 32) * 64 + i) * 4;
    float wA[16][2], wB[16][2];
    for (int kp = 0; kp < 16; ++kp) { wA[kp][0] = wB[kp][0] = 1.f + kp; wA[kp][1] = wB[kp][1] = 2.f + kp; }
    const float *hsb = Hs + h * CONV3_STRIDE + 32 + i;
    if (V >= 2) conv2_load_w(wA, wr, woff, 0, lagbytes, rowbytes);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) { // one iteration = a lag pair = 128 MFMAs
        const int l0 = (2 * it) % 18;
        if (V == 3) {
            conv2_load_w(wB, wr, woff, l0 + 1, lagbytes, rowbytes);
            __builtin_amdgcn_sched_barrier(0);
            conv2_lag<0, CONV3_STRIDE>(acc, hsb - l0, wA);
            conv2_load_w(wA, wr, woff, l0 + 2, lagbytes, rowbytes);
            __builtin_amdgcn_sched_barrier(0);
            conv2_lag<0, CONV3_STRIDE>(acc, hsb - l0 - 1, wB);
        } else {
            for (int half = 0; half < 2; ++half) {
                if (V == 2) {
                    if (half == 0) conv2_load_w(wB, wr, woff, l0 + 1, lagbytes, rowbytes);
                    else conv2_load_w(wA, wr, woff, l0 + 2, lagbytes, rowbytes);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const float (&w)[16][2] = half ? wB : wA;
                const float *hb = hsb - l0 - half;
                float a0 = (V >= 1) ? hb[0] : 1.f, a1 = (V >= 1) ? hb[32] : 2.f;
#pragma unroll
                for (int kp = 0; kp < 16; ++kp) {
                    float na0 = a0, na1 = a1;
                    if (V >= 1 && kp + 1 < 16) { na0 = hb[(kp + 1) * 2 * CONV3_STRIDE]; na1 = hb[(kp + 1) * 2 * CONV3_STRIDE + 32]; }
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, w[kp][0], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, w[kp][1], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, w[kp][0], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, w[kp][1], acc[1][1], 0, 0, 0);
                    a0 = na0; a1 = na1;
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[blockIdx.x * 64 + lane] = s;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V> static int run(const float *W, float *out, unsigned long long *cyc, int waves, int iters, int Np, const char *name)
{
    hipLaunchKernelGGL((mix_kernel<V>), dim3(waves), dim3(64), 0, 0, W, out, cyc, iters, Np);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((mix_kernel<V>), dim3(waves), dim3(64), 0, 0, W, out, cyc, iters, Np);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> c(waves);
    CK(hipMemcpy(c.data(), cyc, waves * 8, hipMemcpyDeviceToHost));
    double mean = 0; for (auto v : c) mean += (double)v; mean /= waves;
    const double per_mfma_wave = mean / (iters * 128.0);
    const double wps = waves / 1024.0;
    printf("%-28s waves/SIMD %.0f: %.1f cycles per MFMA per wave -> %.1f per MFMA per SIMD (pipe util %.3f), %.3f ms\n", name, wps,
           per_mfma_wave, per_mfma_wave / wps, 64.0 * wps / per_mfma_wave, ms);
    return 0;
}

int main()
{
    const int Np = 2048, iters = 400;
    float *W, *out; unsigned long long *cyc;
    CK(hipMalloc(&W, (size_t)20 * 32 * Np * 4)); CK(hipMemset(W, 0, (size_t)20 * 32 * Np * 4));
    CK(hipMalloc(&out, (size_t)3072 * 64 * 4)); CK(hipMalloc(&cyc, 3072 * 8));
    for (int waves : {1024, 2048, 3072}) {
        if (run<0>(W, out, cyc, waves, iters, Np, "MFMA only")) return 1;
        if (run<1>(W, out, cyc, waves, iters, Np, "+ LDS operand reads")) return 1;
        if (run<2>(W, out, cyc, waves, iters, Np, "+ W buffer loads")) return 1;
        if (run<3>(W, out, cyc, waves, iters, Np, "conv2_lag + conv2_load_w")) return 1;
    }
    return 0;
}
