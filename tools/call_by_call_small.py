#!/usr/bin/env python3
"""Call-by-call MU iterations (the reference's loop: update_motifs!, loss = update_feature_maps!) on small problems, where the host
round trip per iteration matters: ms per iteration against the pipelined cmf_iterate.
    python3 tools/call_by_call_small.py [speculate=1]      (0: option "speculate" off; the polled loss word has no switch any more --
round 5's comparison of both is profiles/r05_call_by_call_small.txt)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cmf_jl_amd as cmf  # noqa: E402

for N, T, K, L in [(500, 2000, 5, 10), (250, 6250, 5, 20), (250, 50000, 5, 20), (2000, 6250, 32, 20)]:
    data = cmf.gen_synthetic(N=N, T=T, seed=1234)
    W0, H0 = cmf.init_rand(data, L=L, K=K, seed=0)
    rule = cmf.MultUpdate(data, W0, H0)
    rule.set_option("speculate", int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    rule.iterate(20)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(200):
            rule.update_motifs()
            rule.update_feature_maps()
        best = min(best, (time.perf_counter() - t0) / 200)
    rule.synchronize()
    t0 = time.perf_counter()
    rule.iterate(400)
    pipe = (time.perf_counter() - t0) / 400
    print(f"N={N} T={T} K={K} L={L}: call by call {1e3 * best:7.4f} ms, cmf_iterate {1e3 * pipe:7.4f} ms per iteration ({best / pipe:5.3f} x)", flush=True)
    rule.close()
