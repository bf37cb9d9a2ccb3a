// What a handle's allocations cost: N separate hipMalloc + hipMemset + null-stream synchronisation (and hipFree each) against one arena.
//   hipcc --offload-arch=gfx950 -O2 tools/alloc_cost.hip -o /tmp/alloc_cost && /tmp/alloc_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipFree(nullptr);
    const size_t sizes[] = {1u << 20, 1u << 20, 1u << 18, 1u << 18, 4u << 20, 4u << 20, 4u << 20, 4u << 20, 8u << 20, 1u << 20, 2u << 20, 2u << 20, 1u << 18,
                            4096, 4096, 4096, 4096, 1u << 16, 64, 1u << 14, 1u << 14};
    const int n = sizeof(sizes) / sizeof(sizes[0]);
    for (int rep = 0; rep < 3; ++rep) {
        std::vector<void *> p(n);
        double t0 = now();
        for (int i = 0; i < n; ++i) { hipMalloc(&p[i], sizes[i]); hipMemset(p[i], 0, sizes[i]); hipStreamSynchronize(nullptr); }
        double t1 = now();
        for (int i = 0; i < n; ++i) hipFree(p[i]);
        double t2 = now();
        size_t tot = 0;
        for (int i = 0; i < n; ++i) tot += (sizes[i] + 255) & ~size_t(255);
        void *a;
        hipMalloc(&a, tot); hipMemset(a, 0, tot); hipStreamSynchronize(nullptr);
        double t3 = now();
        hipFree(a);
        double t4 = now();
        hipStream_t s; hipEvent_t e0, e1; void *hp;
        hipStreamCreateWithFlags(&s, hipStreamNonBlocking); hipEventCreate(&e0); hipEventCreate(&e1);
        double t5 = now();
        hipHostMalloc(&hp, 32);
        double t6 = now();
        hipHostFree(hp);
        double t7 = now();
        hipEventDestroy(e0); hipEventDestroy(e1); hipStreamDestroy(s);
        double t8 = now();
        printf("%d buffers: malloc+memset+sync %.0f us, free %.0f us | one arena: alloc %.0f us, free %.0f us | stream+2 events %.0f us, destroy %.0f us | hipHostMalloc %.0f us, hipHostFree %.0f us\n",
               n, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t8 - t7, t6 - t5, t7 - t6);
    }
    return 0;
}
