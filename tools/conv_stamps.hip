// Debug tool (not part of the library): per-workgroup s_memtime stamps of conv2_kernel at the config-2 shape.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -DCMF_CONV_STAMPS -Icmf.jl_amd/csrc tools/conv_stamps.hip -o gpurun_out/conv_stamps
#include <hip/hip_runtime.h>
#include "cmf_kernels.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int MODE> static int run(ConvParams p, dim3 grid, unsigned long long *d_st, int nwg, const char *name)
{
    std::vector<unsigned long long> st((size_t)nwg * 8);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((conv2_kernel<MODE>), grid, dim3(256), 0, 0, p);
        CK(hipDeviceSynchronize());
    }
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL((conv2_kernel<MODE>), grid, dim3(256), 0, 0, p);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: %.4f ms per launch (10 back-to-back launches, stamps on)\n", name, ms / 10);
    }
    CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    {   // raw dump for tools/conv_stamps_analyze.py
        char fn[256]; snprintf(fn, sizeof fn, "gpurun_out/stamps_%s.bin", name);
        FILE *f = fopen(fn, "wb"); if (f) { fwrite(st.data(), 8, st.size(), f); fclose(f); }
    }
    unsigned long long t0 = ~0ull, t1 = 0;
    double pro = 0, mainl = 0, epi = 0;
    for (int w = 0; w < nwg; ++w) {
        t0 = std::min(t0, st[8 * w]); t1 = std::max(t1, st[8 * w + 3]);
        pro += (double)(st[8 * w + 1] - st[8 * w]); mainl += (double)(st[8 * w + 2] - st[8 * w + 1]); epi += (double)(st[8 * w + 3] - st[8 * w + 2]);
    }
    double p5 = 0, p6 = 0, p7 = 0;
    for (int w = 0; w < nwg; ++w) { p5 += (double)(st[8 * w + 5] - st[8 * w]); p6 += (double)(st[8 * w + 6] - st[8 * w]); p7 += (double)(st[8 * w + 7] - st[8 * w]); }
    printf("%s: start -> before W loads %.0f -> W loads issued %.0f -> after 1st barrier %.0f -> H strip staged (main loop starts) %.0f ticks\n", name, p5 / nwg, p6 / nwg, p7 / nwg, pro / nwg);
    printf("%s: span %.1f kticks; per workgroup mean: prologue %.0f  main loop %.0f  epilogue %.0f ticks\n", name, (t1 - t0) / 1e3, pro / nwg, mainl / nwg, epi / nwg);
    // start-time histogram: how many workgroups started in each 5%% slice of the span
    int hist[20] = {0}, hend[20] = {0};
    for (int w = 0; w < nwg; ++w) {
        hist[std::min<int>(19, (int)((st[8 * w] - t0) * 20 / (t1 - t0 + 1)))]++;
        hend[std::min<int>(19, (int)((st[8 * w + 3] - t0) * 20 / (t1 - t0 + 1)))]++;
    }
    printf("  starts per 5%% slice:"); for (int b = 0; b < 20; ++b) printf(" %d", hist[b]); printf("\n");
    printf("  ends   per 5%% slice:"); for (int b = 0; b < 20; ++b) printf(" %d", hend[b]); printf("\n");
    return 0;
}
int main(int argc, char **argv)
{
    const int N = 2000, T = argc > 1 ? atoi(argv[1]) : 50000, K = 32, L = 20;
    const int Np = 2048, K32 = 32, PADL = 64, TP = PADL + ((T + L + 511) / 512) * 512 + 256, Lp = 20;
    float *Ht, *Wt, *out, *X; double *partial; unsigned long long *d_st;
    const int gx = Np / 128, gy = (T + 127) / 128, nwg = gx * gy;
    CK(hipMalloc(&Ht, (size_t)K32 * TP * 4)); CK(hipMalloc(&Wt, (size_t)Lp * K32 * Np * 4));
    CK(hipMalloc(&out, (size_t)TP * Np * 4)); CK(hipMalloc(&X, (size_t)TP * Np * 4));
    CK(hipMalloc(&partial, (size_t)nwg * 8)); CK(hipMalloc(&d_st, (size_t)nwg * 8 * 8));
    std::vector<float> hbuf((size_t)TP * Np);
    for (auto &v : hbuf) v = (float)rand() / RAND_MAX;
    CK(hipMemcpy(Ht, hbuf.data(), (size_t)K32 * TP * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(Wt, hbuf.data(), (size_t)Lp * K32 * Np * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(X, hbuf.data(), (size_t)TP * Np * 4, hipMemcpyHostToDevice));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(cmf_stamps), &d_st, sizeof(d_st)));
    ConvParams p;
    p.Ht = Ht; p.Wt = Wt; p.out = out; p.data = X; p.mask = X; p.partial = partial;
    p.Np = Np; p.TP = TP; p.PADL = PADL; p.K = K; p.KB = 1; p.L = L; p.T_store = T;
    (void)N;
    if (run<0>(p, dim3(gx, gy), d_st, nwg, "mode0")) return 1;
    if (run<1>(p, dim3(gx, gy), d_st, nwg, "mode1")) return 1;
    if (run<2>(p, dim3(gx, gy), d_st, nwg, "mode2")) return 1;
    if (run<3>(p, dim3(gx, gy), d_st, nwg, "mode3")) return 1;
    return 0;
}
