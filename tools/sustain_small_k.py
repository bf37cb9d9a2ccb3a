#!/usr/bin/env python3
"""Does a few-component iteration get faster when the loop runs for seconds instead of milliseconds (clock ramp)?
    python3 tools/sustain_small_k.py [N T K L]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cmf_jl_amd as cmf  # noqa: E402

N, T, K, L = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (250, 50000, 5, 20)
data = cmf.gen_synthetic(N=N, T=T, seed=1234)
W0, H0 = cmf.init_rand(data, L=L, K=K, seed=0)
rule = cmf.MultUpdate(data, W0, H0)
rule.iterate(3)
rule.synchronize()
for n in (50, 50, 500, 5000, 20000, 50, 500):
    t0 = time.perf_counter()
    rule.iterate(n)
    rule.synchronize()
    dt = time.perf_counter() - t0
    print(f"N={N} T={T} K={K} L={L}: {n:6d} iterations in {dt:7.3f} s: {1e3 * dt / n:.4f} ms per iteration", flush=True)
rule.close()
