#!/usr/bin/env python3
"""Two independent contractions of an MU iteration on two streams at once vs one after the other:
    python3 tools/pair_experiment.py [T=6250 ...]          (N = 2000, K = 32, L = 20)
    python3 tools/pair_experiment.py shape N T K L         (any shape, e.g. the few-component protocol shape 250 50000 5 20)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cmf_jl_amd as cmf  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] == "shape":
    shapes = [tuple(int(a) for a in sys.argv[2:6])]
else:
    shapes = [(2000, int(a), 32, 20) for a in sys.argv[1:]] or [(2000, 6250, 32, 20)]
for N, T, K, L in shapes:
    data = cmf.gen_synthetic(N=N, T=T, seed=1234)
    W0, H0 = cmf.init_rand(data, L=L, K=K, seed=0)
    rule = cmf.MultUpdate(data, W0, H0)
    rule.iterate(2)
    for name in ("seq_conv_tc", "pair_conv_tc", "seq_loss_hxt", "pair_loss_hxt", "conv_t", "conv_loss_store", "hxt", "transconv"):
        ms = [rule.time_kernel(name, reps=20)[0] for _ in range(3)]
        print(f"N={N} T={T} K={K} L={L} {name:16s} {min(ms)*1e3:8.1f} us", flush=True)
    rule.close()
