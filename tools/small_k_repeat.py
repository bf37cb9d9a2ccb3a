#!/usr/bin/env python3
"""Few-component kernels: the same fit twice (and a third time after other work on the device) must agree bit for bit -- a strip read
before its LDS-DMA has landed, or any other race, shows as a difference.   python3 tools/small_k_repeat.py [iterations=60]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cmf_jl_amd as cmf  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
for N, T, K, L in [(250, 50000, 5, 20), (250, 20000, 12, 20), (2000, 10000, 16, 20), (64, 5000, 9, 33), (500, 2000, 5, 10), (100, 30000, 3, 64)]:
    data = cmf.gen_synthetic(N=N, T=T, seed=11)
    W0, H0 = cmf.init_rand(data, L=L, K=K, seed=2)
    outs = []
    for rep in range(3):
        rule = cmf.MultUpdate(data, W0, H0)
        if rep == 2:  # a different neighbourhood in time: another handle busy on the same device
            other = cmf.MultUpdate(data, W0, H0)
            other.iterate(5)
        losses = rule.iterate(iters)
        outs.append((np.asarray(losses),) + rule.download())
        rule.close()
        if rep == 2:
            other.close()
    same = all(np.array_equal(outs[0][j], outs[r][j]) for r in (1, 2) for j in range(3))
    bad += not same
    print(f"N={N} T={T} K={K} L={L}: {iters} iterations three times: {'bitwise equal' if same else 'DIFFERENT'}; loss {outs[0][0][-1]:.6f}", flush=True)
sys.exit(1 if bad else 0)
