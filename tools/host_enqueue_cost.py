#!/usr/bin/env python3
"""Host time per iteration of a one-process group of R shards when the device work is negligible (tiny T): what the single
host thread of cmf_create_multi spends enqueueing one sharded iteration (launches, hipSetDevice, events).
    python3 tools/host_enqueue_cost.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cmf_jl_amd as cmf  # noqa: E402
from cmf_jl_amd import _lib  # noqa: E402

N, K, L = 64, 32, 20
for R in (1, 2, 4, 8):
    T = 64 * R
    data = cmf.gen_synthetic(N=N, T=T, seed=1)
    W0, H0 = cmf.init_rand(data, L=L, K=K, seed=0)
    for tr, name in ((_lib.CMF_COMM_LOOPBACK, "shared stream"), (_lib.CMF_COMM_LOOPBACK_STREAMS, "stream per shard")):
        rule = cmf.MultUpdate(data, W0, H0, devices=[0] * R, transport=tr)
        rule.iterate(20)
        t0 = time.perf_counter()
        n = 300
        rule.iterate(n)
        dt = (time.perf_counter() - t0) / n
        print(f"R={R} {name:16s}: {1e6 * dt:8.1f} us per iteration ({1e6 * dt / R:6.1f} per shard)", flush=True)
        rule.close()
