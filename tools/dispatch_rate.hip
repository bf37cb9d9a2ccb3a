// How long does the chip take to START a grid of one-wave workgroups?  (a T/8 shard's conv launch is 3136 one-wave
// workgroups that each live ~105 us; the launch takes ~120.)  Every wave spins for `work_us` microseconds of s_memrealtime and
// leaves; the kernel's duration minus work_us is the time the dispatcher needed to bring the last wave up (plus the drain).
//   hipcc -O3 --offload-arch=gfx950 tools/dispatch_rate.hip -o tools/bin/dispatch_rate && tools/bin/dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NW, int LDS_FLOATS, int VGPRS>
__global__ __launch_bounds__(64 * NW) void spin_kernel(float *out, unsigned long long ticks)
{
    __shared__ float lds[LDS_FLOATS * NW];
    float v[VGPRS];
#pragma unroll
    for (int i = 0; i < VGPRS; ++i) v[i] = (float)(threadIdx.x + i);
    lds[threadIdx.x] = v[0];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < VGPRS; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v[i]));
    }
    float s = lds[(threadIdx.x * 7) % (64 * NW)];
#pragma unroll
    for (int i = 0; i < VGPRS; ++i) s += v[i];
    if (s == 123.456f) out[0] = s;
}

template <int NW, int LDS_FLOATS, int VGPRS>
static void run(const char *name, int nwg, double work_us, float *out)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const unsigned long long ticks = (unsigned long long)(work_us * 100.0);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((spin_kernel<NW, LDS_FLOATS, VGPRS>), dim3(nwg), dim3(64 * NW), 0, 0, out, ticks);
    hipEventRecord(a, 0);
    const int reps = 20;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((spin_kernel<NW, LDS_FLOATS, VGPRS>), dim3(nwg), dim3(64 * NW), 0, 0, out, ticks);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    printf("%-64s %5d workgroups x %d waves, work %.0f us: %.2f us per launch (+%.2f)\n", name, nwg, NW, work_us, 1e3 * ms / reps, 1e3 * ms / reps - work_us);
}

int main()
{
    float *out;
    hipMalloc(&out, 1024);
    for (double work : {0.0, 20.0, 100.0}) {
        run<1, 3072, 120>("one-wave workgroups, 12 KB LDS, ~128 VGPRs (3 per SIMD)", 3072, work, out);
        run<1, 3072, 120>("one-wave workgroups, 12 KB LDS, ~128 VGPRs (3 per SIMD)", 3136, work, out);
        run<1, 3072, 120>("one-wave workgroups, 12 KB LDS, ~128 VGPRs (3 per SIMD)", 1536, work, out);
        run<4, 3072, 120>("four-wave workgroups, 48 KB LDS, ~128 VGPRs", 768, work, out);
        run<4, 3072, 120>("four-wave workgroups, 48 KB LDS, ~128 VGPRs", 784, work, out);
        run<1, 64, 120>("one-wave workgroups, 256 B LDS, ~128 VGPRs", 3072, work, out);
        run<1, 3072, 24>("one-wave workgroups, 12 KB LDS, few VGPRs", 3072, work, out);
        run<4, 64, 200>("four-wave workgroups, ~208 VGPRs (2 per SIMD), hxt-like", 512, work, out);
    }
    return 0;
}
