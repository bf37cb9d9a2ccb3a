#!/usr/bin/env python3
"""PGD iterations (pgd.jl:158-255, default loss / penalties / constraints) at config 2's size, one GPU and as loopback shards:
    python3 tools/time_pgd.py [T=50000] [iters=10]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cmf_jl_amd as cmf  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
data = cmf.gen_synthetic(N=2000, T=T, seed=1234)
W0, H0 = cmf.init_rand(data, L=20, K=32, seed=0)
for devices in (None, [0, 0]):
    rule = cmf.PGDUpdate(data, W0, H0, devices=devices)
    rule.update_motifs(); rule.update_feature_maps()
    t0 = time.perf_counter()
    for _ in range(iters):
        rule.update_motifs()
        loss = rule.update_feature_maps()
    dt = (time.perf_counter() - t0) / iters
    print(f"T={T} devices={devices}: {1e3 * dt:.3f} ms per PGD iteration, loss {loss:.5f}, steps {rule.steps}", flush=True)
    rule.close()
