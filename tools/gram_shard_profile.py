#!/usr/bin/env python3
"""Iterations of the Gram form (option gram = 1) at a shard's size, for `rocprofv3 --kernel-trace --stats`:
    python3 tools/gram_shard_profile.py [T=6250] [iters=50] [gram=1]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cmf_jl_amd as cmf  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 6250
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
gram = int(sys.argv[3]) if len(sys.argv) > 3 else 1
data = cmf.gen_synthetic(N=2000, T=T, seed=1234)
W0, H0 = cmf.init_rand(data, L=20, K=32, seed=0)
rule = cmf.MultUpdate(data, W0, H0)
rule.set_option("gram", gram)
rule.iterate(3)
t0 = time.perf_counter()
rule.iterate(iters)
print(f"T={T} gram={gram}: {1e3 * (time.perf_counter() - t0) / iters:.4f} ms per iteration")
rule.close()
