// valu_latency.hip -- issue-to-issue latency of dependent single-wave instruction chains on gfx950 (one wave per
// SIMD, nothing to hide latency behind): what bounds the HALS row sweep's step.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_latency.hip -o tools/bin/valu_latency && tools/bin/valu_latency
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x
#define N_OUTER 64

// the walking step unrolled over 16 lanes with a tap register of its own per step (what hals_h_row_sweep would hold), J as an immediate
template <int J>
__device__ __forceinline__ void walk16(float &v, float w, float &U, const float (&g)[16], float mh, float &dacc, float &t)
{
    asm volatile("v_max_f32 %0, %5, %3\n\t"
                 "s_nop 0\n\t"
                 "v_readlane_b32 s20, %0, %6\n\t"
                 "s_nop 1\n\t"
                 "v_fma_f32 %5, s20, %1, %2\n\t"
                 "v_fmac_f32 %2, s20, %7\n\t"
                 "v_writelane_b32 %4, s20, %6"
                 : "+v"(v), "+v"(w), "+v"(U), "+v"(mh), "+v"(dacc), "+v"(t) : "n"(J), "v"(g[J]) : "s20");
    if constexpr (J + 1 < 16) walk16<J + 1>(v, w, U, g, mh, dacc, t);
}

template <int KIND>
__global__ void chain(float *out, unsigned long long *cycles, float a, float b)
{
    float v = a + threadIdx.x, w = b;
    float sidx = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < N_OUTER; ++it) {
        if (KIND == 0) { // dependent v_fma
            REP16(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(w));)
        } else if (KIND == 1) { // v_readfirstlane -> v_fma with the SGPR as an operand
            REP16(asm volatile("v_readfirstlane_b32 s20, %0\n\tv_fma_f32 %0, s20, %1, %0" : "+v"(v) : "v"(w) : "s20");)
        } else if (KIND == 2) { // dependent DPP wave_shl
            REP16(asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(v));)
        } else if (KIND == 3) { // fma -> dpp -> fma -> dpp
            REP16(asm volatile("v_fma_f32 %0, %0, %1, %1\n\ts_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(v) : "v"(w));)
        } else if (KIND == 4) { // fma -> max (the sweep's chain)
            REP16(asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_max_f32 %0, %0, %1" : "+v"(v) : "v"(w));)
        } else if (KIND == 5) { // independent fmas (issue rate)
            float x0 = v, x1 = v + 1, x2 = v + 2, x3 = v + 3;
            REP16(asm volatile("v_fma_f32 %0, %0, %4, %4\n\tv_fma_f32 %1, %1, %4, %4\n\tv_fma_f32 %2, %2, %4, %4\n\tv_fma_f32 %3, %3, %4, %4"
                               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(w));)
            v = x0 + x1 + x2 + x3;
        } else if (KIND == 6) { // readfirstlane -> readfirstlane dependent through a v_mov from the SGPR
            REP16(asm volatile("v_readfirstlane_b32 s20, %0\n\tv_mov_b32 %0, s20" : "+v"(v) : : "s20");)
        } else if (KIND == 8) { // the HALS row sweep's step in its V form (hals_h_row_sweep fast path), d in %0, V in %2
            float V = w + threadIdx.x, vnr = w, mh = -w, dv = 0.f;
            REP16(asm volatile("v_readfirstlane_b32 s20, %0\n\t"
                               "v_mov_b32_dpp %3, %3 wave_rol:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                               "v_mov_b32 %5, %3\n\t"
                               "v_mov_b32_dpp %5, %2 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                               "v_fma_f32 %2, s20, %1, %5\n\t"
                               "v_fma_f32 %0, %1, %0, %5\n\t"
                               "v_mov_b32_dpp %4, %4 wave_rol:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                               "v_max_f32 %0, %0, %4\n\t"
                               "v_writelane_b32 %6, s20, 3"
                               : "+v"(v), "+v"(w), "+v"(V), "+v"(vnr), "+v"(mh), "=&v"(sidx), "+v"(dv) : : "s20");)
            v += V + dv;
        } else if (KIND == 9) { // a sweep step that WALKS THE LANES instead of shifting the state: column j lives in lane j; the chain is
            // readlane(d, j-1) -> fma(c1 * d_{j-1} + U) -> max(., -h); off the chain: keep lane j of d, U += d_{j-1} * taps, shift the taps
            float U = w + threadIdx.x, g = w, mh = -w, dacc = 0.f, t = 0.f;
            REP16(asm volatile("v_readlane_b32 s20, %0, 5\n\t"
                               "v_fma_f32 %6, s20, %1, %2\n\t"
                               "v_max_f32 %0, %6, %4\n\t"
                               "v_cndmask_b32 %5, %5, %0, vcc\n\t"
                               "v_fma_f32 %2, s20, %3, %2\n\t"
                               "s_nop 0\n\t"
                               "v_mov_b32_dpp %3, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                               : "+v"(v), "+v"(w), "+v"(U), "+v"(g), "+v"(mh), "+v"(dacc), "+v"(t) : : "s20", "vcc");)
            v += U + dacc + g;
        } else if (KIND == 10) { // the same with two columns' worth of taps per DPP shift left out (pre-shifted tap registers): 5 instructions
            float U = w + threadIdx.x, g = w, mh = -w, dacc = 0.f, t = 0.f;
            REP16(asm volatile("v_readlane_b32 s20, %0, 5\n\t"
                               "v_fma_f32 %6, s20, %1, %2\n\t"
                               "v_max_f32 %0, %6, %4\n\t"
                               "v_cndmask_b32 %5, %5, %0, vcc\n\t"
                               "v_fma_f32 %2, s20, %3, %2"
                               : "+v"(v), "+v"(w), "+v"(U), "+v"(g), "+v"(mh), "+v"(dacc), "+v"(t) : : "s20", "vcc");)
            v += U + dacc + g;
        } else if (KIND == 11) { // ... the taps of the step read from an LDS table (ds_read_b32 three steps ahead) instead of held in 64 registers
            __shared__ float tab[64 * 8];
            tab[threadIdx.x] = w;
            float U = w + threadIdx.x, g = w, g2 = w, mh = -w, dacc = 0.f, t = 0.f;
            unsigned addr = threadIdx.x * 4;
            REP16(asm volatile("ds_read_b32 %7, %8 offset:256\n\t"
                               "v_readlane_b32 s20, %0, 5\n\t"
                               "v_fma_f32 %6, s20, %1, %2\n\t"
                               "v_max_f32 %0, %6, %4\n\t"
                               "v_cndmask_b32 %5, %5, %0, vcc\n\t"
                               "s_waitcnt lgkmcnt(3)\n\t"
                               "v_fma_f32 %2, s20, %3, %2"
                               : "+v"(v), "+v"(w), "+v"(U), "+v"(g), "+v"(mh), "+v"(dacc), "+v"(t), "=v"(g2) : "v"(addr) : "s20", "vcc");)
            asm volatile("s_waitcnt lgkmcnt(0)");
            v += U + dacc + g + g2;
        } else if (KIND == 12) { // the walking step as the compiler emits it for gfx950 (hazard nops: VALU -> v_readlane, SGPR written by VALU -> VALU)
            float U = w + threadIdx.x, g = w, mh = -w, dacc = 0.f, t = v;
            REP16(asm volatile("v_max_f32 %0, %6, %4\n\t"
                               "s_nop 0\n\t"
                               "v_readlane_b32 s20, %0, 5\n\t"
                               "v_writelane_b32 %5, s20, 5\n\t"
                               "s_nop 1\n\t"
                               "v_fma_f32 %6, s20, %1, %2\n\t"
                               "v_fmac_f32 %2, s20, %3"
                               : "+v"(v), "+v"(w), "+v"(U), "+v"(g), "+v"(mh), "+v"(dacc), "+v"(t) : : "s20");)
            v += U + dacc + g + t;
        } else if (KIND == 13) { // the same, lanes 0..15 and 16 tap registers
            float U = w + threadIdx.x, mh = -w, dacc = 0.f, t = v;
            float g[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) g[j] = w + j;
            walk16<0>(v, w, U, g, mh, dacc, t);
            v += U + dacc + t;
        } else if (KIND == 14) { // the window form without the per-step entry of a column (zero-fill shift, the next 32 columns put in at once
            // every 32 steps: L <= 33): the v_mov and the rotation of the entering values drop out -- 7 instructions
            float V = w + threadIdx.x, mh = -w, dv = 0.f;
            REP16(asm volatile("v_readfirstlane_b32 s20, %0\n\t"
                               "v_mov_b32_dpp %4, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                               "v_fma_f32 %2, s20, %1, %4\n\t"
                               "v_fma_f32 %0, %1, %0, %4\n\t"
                               "v_mov_b32_dpp %3, %3 wave_rol:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                               "v_max_f32 %0, %0, %3\n\t"
                               "v_writelane_b32 %5, s20, 3"
                               : "+v"(v), "+v"(w), "+v"(V), "+v"(mh), "=&v"(sidx), "+v"(dv) : : "s20");)
            v += V + dv;
        } else if (KIND == 7) { // v_readlane with immediate -> fma
            REP16(asm volatile("v_readlane_b32 s20, %0, 5\n\tv_fma_f32 %0, s20, %1, %0" : "+v"(v) : "v"(w) : "s20");)
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = v + sidx;
    if (threadIdx.x == 0) cycles[0] = t1 - t0;
}

template <int KIND>
static void run(const char *name, int instr_per_rep)
{
    float *out;
    unsigned long long *cyc, h = 0;
    hipMalloc(&out, 64 * 4);
    hipMalloc(&cyc, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0.f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(chain<KIND>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0f, 0.5f);
        hipEventRecord(e1, 0);
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-46s %8llu ticks  %.3f ticks/rep (%d instr per rep)   kernel %.1f us\n", name, h, (double)h / (N_OUTER * 16), instr_per_rep, ms * 1e3);
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    run<5>("4 independent v_fma (issue rate)", 4);
    run<0>("dependent v_fma", 1);
    run<4>("v_fma -> v_max (dependent pair)", 2);
    run<2>("dependent v_mov_dpp wave_shl (+s_nop 1)", 1);
    run<3>("v_fma -> dpp -> ...", 2);
    run<1>("v_readfirstlane -> v_fma(sgpr)", 2);
    run<7>("v_readlane imm -> v_fma(sgpr)", 2);
    run<6>("v_readfirstlane -> v_mov(sgpr)", 2);
    run<8>("HALS sweep step, V form (9 instr)", 9);
    run<9>("sweep step walking the lanes (6 instr)", 6);
    run<10>("... with pre-shifted taps (5 instr)", 5);
    run<11>("... the taps from an LDS table (5 + ds_read)", 6);
    run<12>("walking step as compiled (5 instr + hazard nops)", 5);
    run<13>("... lanes 0..15, a tap register per step", 5);
    run<14>("window form, columns entering 32 at a time (7 instr)", 7);
    return 0;
}
