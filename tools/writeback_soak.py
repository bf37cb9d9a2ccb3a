#!/usr/bin/env python3
"""Soak of the armed write-back (cmf_arm_writeback): thousands of call-by-call iterations, the caller's arrays compared bit for bit
with cmf_get_factors after EVERY update_feature_maps -- a stale or torn delivery (a helper reading the staging too early, an event
recorded late, a flag seen before its copy) shows as a mismatch.  A single handle, then 8 loopback shards with an enqueue thread
per shard, with K a multiple of 32 (plain DMA copies) and not (pack kernels on the copy stream).
    python3 tools/writeback_soak.py [iterations=3000]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cmf_jl_amd as cmf  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
bad = 0
for K in (32, 5):
    for shards in (1, 8):
        N, T, L = 96, 2400, 8
        data = cmf.gen_synthetic(N=N, T=T, seed=7)
        W0, H0 = cmf.init_rand(data, L=L, K=K, seed=1)
        kw = dict(devices=[0] * shards, transport=3) if shards > 1 else {}
        rule = cmf.MultUpdate(data, W0, H0, **kw)
        if shards > 1:
            rule.set_option("enqueue_threads", 1)
        rule.sync_every_call = True
        rule.verify_args = "none"
        W = np.asfortranarray(W0.copy())
        H = np.asfortranarray(H0.copy())
        t0 = time.time()
        last = time.time()
        for it in range(iters):
            rule.update_motifs(data, W, H)
            rule.update_feature_maps(data, W, H)
            Wd, Hd = rule.download()
            if not (np.array_equal(W, Wd) and np.array_equal(H, Hd)):
                bad += 1
                print(f"MISMATCH K={K} shards={shards} iteration {it}: W {np.abs(W - Wd).max():.3e} H {np.abs(H - Hd).max():.3e}", flush=True)
                if bad > 5:
                    sys.exit(1)
            if time.time() - last > 30:
                print(f"  ... K={K} shards={shards} iteration {it}", flush=True)
                last = time.time()
        print(f"K={K:2d} shards={shards}: {iters} armed iterations, every delivery bitwise cmf_get_factors; "
              f"{(time.time() - t0) / iters * 1e3:.3f} ms per iteration incl. the check; overlapped {rule.counter('writeback_overlapped')}", flush=True)
        rule.close()
sys.exit(1 if bad else 0)
