#!/usr/bin/env python3
"""HALS and PGD at the reference's protocol shape (N=250 T=50000 K=5 L=20) with the few-component kernels on (default) and off:
ms per iteration (wall clock over 20 iterations, rule methods called like the reference's loop does)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cmf_jl_amd as cmf  # noqa: E402

N, T, K, L = 250, 50000, 5, 20
data = cmf.gen_synthetic(N=N, T=T, seed=1234)
W0, H0 = cmf.init_rand(data, L=L, K=K, seed=0)
for name, cls in (("hals", cmf.HALSUpdate), ("pgd", cmf.PGDUpdate), ("mult", cmf.MultUpdate)):
    for small in (1, 0):
        rule = cls(data, W0, H0)
        rule.set_option("small_k", small)
        for _ in range(3):
            rule.update_motifs(); rule.update_feature_maps()
        t0 = time.perf_counter()
        for _ in range(20):
            rule.update_motifs()
            loss = rule.update_feature_maps()
        dt = (time.perf_counter() - t0) / 20
        rule.close()
        print(f"{name:5s} small_k={small}: {1e3 * dt:7.3f} ms per iteration (call by call, loss read back every iteration), loss {loss:.4f}", flush=True)
