#!/usr/bin/env python3
"""What the reference's in-place semantics cost on a T-sharded group driven call by call (alternating.jl:51-59): an 8-shard group of
config 2 on ONE GPU (loopback-streams transport with enqueue workers; the shards share the device, so the iteration itself is
~8 shard iterations long -- what matters here is the ABSOLUTE extra time per iteration of each way to deliver W and H).
    python3 tools/group_writeback_cost.py [T=50000] [shards=8] [steps=20]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cmf_jl_amd as cmf  # noqa: E402
from cmf_jl_amd._lib import check, ptr  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
N, K, L = 2000, 32, 20
data = cmf.gen_synthetic(N=N, T=T, seed=1234)
W0, H0 = cmf.init_rand(data, L=L, K=K, seed=0)
rule = cmf.MultUpdate(data, W0, H0, devices=[0] * R, transport=3)
W = np.zeros((K, N, L), order="F")
H = np.zeros((K, T), order="F")


def loop(n, after=None):
    for _ in range(n):
        rule.update_motifs(data, W, H)
        rule.update_feature_maps(data, W, H)
        if after:
            after()


def timed(after=None):
    loop(2, after)
    rule.synchronize()
    t0 = time.perf_counter()
    loop(steps, after)
    rule.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


rule.sync_every_call = False
base = timed()
rule.sync_every_call = True
rule.verify_args = "none"  # (timing of the write-back alone)
wb = timed()
rule.sync_every_call = False
Wd, Hd = rule.download()
same = bool(np.array_equal(W, Wd) and np.array_equal(H, Hd))
gf = timed(lambda: check(rule._lib.cmf_get_factors(rule._h, ptr(W), ptr(H))))
print(f"{R} loopback shards of N={N} T={T} K={K} L={L} on one GPU, call by call: {base:.3f} ms per iteration; with write-back "
      f"(cmf_arm_writeback) {wb:.3f} ms (+{wb - base:.3f}); with cmf_get_factors after every call {gf:.3f} ms (+{gf - base:.3f}); "
      f"arrays equal cmf_get_factors: {same}; overlapped calls {rule.counter('writeback_overlapped')}")
rule.close()
