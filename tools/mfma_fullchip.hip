// Full-chip fp32 MFMA rate under the launch geometries of the conv kernels, with NO memory traffic: how much of the
// conv kernels' distance to the roof is launch structure (one-wave workgroups, rounds, the tail) and clock, and how much
// is theirs.   hipcc -O3 --offload-arch=gfx950 tools/mfma_fullchip.hip -o tools/bin/mfma_fullchip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// one wave: `n` groups of 4 independent 32x32x2 MFMAs (a 64 x 64 tile's 2 x 2 blocks)
template <int WPB>
__global__ __launch_bounds__(64 * WPB, 3) void mfma_tiles(float *out, int n, int tiles_per_wave)
{
    // occupancy as in the conv kernels (3 waves per SIMD): 13 KB of LDS per wave -> 12 waves per CU
    __shared__ float pad[WPB * 13 * 256];
    if (n < 0) pad[threadIdx.x] = 1.f;
    f32x16 acc[4];
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-6f;
    if (n < 0) a = pad[threadIdx.x ^ 1];
    for (int t = 0; t < tiles_per_wave; ++t) {
        for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
        }
        float s = 0.f;
        for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
        if (s == 12345.f) out[blockIdx.x * blockDim.x + threadIdx.x] = s; // never true: keeps the MFMAs alive
        a += 1e-3f;
    }
}

template <typename F> static double time_ms(F f, int reps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    std::vector<float> ms;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0, 0); f(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float m; hipEventElapsedTime(&m, e0, e1); ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    return ms[0];
}

int main()
{
    float *out; hipMalloc(&out, 64 << 20);
    const int n = 320; // 1280 MFMAs per tile = 20 lags x 16 k pairs x 4
    struct { const char *name; int grid, wpb, tpw; } cases[] = {
        {"25024 one-wave workgroups x 1 tile (conv3 at config 2)", 25024, 1, 1},
        {"24576 one-wave workgroups x 1 tile (8 exact rounds)", 24576, 1, 1},
        {"6256 four-wave workgroups x 1 tile per wave (conv2)", 6256, 4, 1},
        {"6144 four-wave workgroups x 1 tile per wave (8 exact rounds)", 6144, 4, 1},
        {"3072 one-wave workgroups x 8 tiles (persistent, 8 rounds)", 3072, 1, 8},
        {"2048 one-wave workgroups x 12 tiles (2 waves / SIMD)", 2048, 1, 12},
        {"1024 one-wave workgroups x 24 tiles (1 wave / SIMD)", 1024, 1, 24},
    };
    for (auto &c : cases) {
        double ms = c.wpb == 1 ? time_ms([&] { hipLaunchKernelGGL(mfma_tiles<1>, dim3(c.grid), dim3(64), 0, 0, out, n, c.tpw); }, 20)
                               : time_ms([&] { hipLaunchKernelGGL(mfma_tiles<4>, dim3(c.grid), dim3(256), 0, 0, out, n, c.tpw); }, 20);
        const double mfmas = (double)c.grid * c.wpb * c.tpw * 4 * n;
        const double tf = mfmas * 32 * 32 * 2 * 2 / (ms * 1e-3) / 1e12;
        const double cyc = mfmas * 64 / 1024.0; // per SIMD
        printf("%-62s %.4f ms  %.1f TFLOP/s (%.3f of 157.3)  implied clock at 100%% pipe %.3f GHz\n", c.name, ms, tf, tf / 157.3, cyc / (ms * 1e-3) / 1e9);
    }
    return 0;
}
