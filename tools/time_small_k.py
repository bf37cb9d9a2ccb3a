#!/usr/bin/env python3
"""Few-component shapes (K <= 16): per-contraction and per-iteration times with the few-component kernels (csrc/cmf_small_k.h,
option small_k = 1, the default) and with the general kernels (small_k = 0), and the fraction of the fp32 MFMA roof on USEFUL
flops (2*K*N*S per contraction, S = L*T - L(L-1)/2; 6 executed per iteration with est reuse).
    python3 tools/time_small_k.py [N T K L]..."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cmf_jl_amd as cmf  # noqa: E402

PEAK = 157.3e12
shapes = [(250, 50000, 5, 20), (500, 2000, 5, 10), (2000, 50000, 5, 20), (2000, 50000, 16, 20), (250, 50000, 12, 20)]
if len(sys.argv) > 4:
    a = [int(x) for x in sys.argv[1:]]
    shapes = [tuple(a[i:i + 4]) for i in range(0, len(a), 4)]
for N, T, K, L in shapes:
    data = cmf.gen_synthetic(N=N, T=T, seed=1234)
    W0, H0 = cmf.init_rand(data, L=L, K=K, seed=0)
    S = L * T - L * (L - 1) / 2
    f1 = 2.0 * K * N * S
    for small in (1, 0):
        rule = cmf.MultUpdate(data, W0, H0)
        rule.set_option("small_k", small)
        rule.iterate(20)  # (steady state: the first ten milliseconds of a loop of such short iterations run ~4 % slower)
        rule.synchronize()
        n = 200
        t0 = time.perf_counter()
        rule.iterate(n)
        rule.synchronize()
        dt = (time.perf_counter() - t0) / n
        ks = {nm: rule.time_kernel(nm, reps=10)[0] for nm in ("conv_t", "conv_loss_store", "hxt", "transconv")}
        rule.close()
        print(f"N={N} T={T} K={K} L={L} small_k={small}: {1e3 * dt:8.3f} ms/iter ({1 / dt:7.1f} iter/s), useful MFMA frac {6 * f1 / dt / PEAK:6.3f}; "
              + "  ".join(f"{k} {1e3 * v:7.1f} us ({(2 if k in ('hxt', 'transconv') else 1) * f1 / (v * 1e-3) / PEAK:5.3f})" for k, v in ks.items()), flush=True)
