#!/usr/bin/env python3
"""Soak of the in-launch hand-off of the few-component C3 kernel (option small_k_fuse: the workgroup whose ticket completes a block's
slabs updates the block): thousands of H updates on shapes with 1 .. 8 pieces per tile, WHILE another handle keeps its own launches
running on the device from a second thread (uneven load, warm caches), each fit compared bit for bit with the separate-launch form.
A stale slab line or a lost ticket shows as a difference (or as an H update that never happens: the loss stops moving).
    python3 tools/small_k_fuse_soak.py [iterations=400] [rounds=3]"""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cmf_jl_amd as cmf  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
stop = False


def neighbour():
    data = cmf.gen_synthetic(N=300, T=20000, seed=5)
    W0, H0 = cmf.init_rand(data, L=12, K=32, seed=3)
    rule = cmf.MultUpdate(data, W0, H0)
    small = cmf.MultUpdate(data[:, :3000], *cmf.init_rand(data[:, :3000], L=7, K=4, seed=4))
    while not stop:
        rule.iterate(3)
        small.iterate(20)
    rule.close()
    small.close()


th = threading.Thread(target=neighbour)
th.start()
bad = 0
total = 0
try:
    for N, T, K, L in [(250, 50000, 5, 20), (500, 2000, 5, 10), (250, 8000, 5, 20), (64, 5000, 8, 33), (100, 30000, 3, 64), (130, 700, 7, 19), (33, 400, 4, 8)]:
        data = cmf.gen_synthetic(N=N, T=T, seed=11)
        W0, H0 = cmf.init_rand(data, L=L, K=K, seed=2)
        ref = None
        same = True
        for r in range(rounds + 1):
            rule = cmf.MultUpdate(data, W0, H0)
            rule.set_option("small_k", 2)
            rule.set_option("small_k_fuse", 0 if r == 0 else 2)
            losses = rule.iterate(iters, l1H=0.01, l2W=0.01)
            out = (np.asarray(losses),) + rule.download()
            total += 0 if r == 0 else rule.counter("small_k_fused_h_updates")
            rule.close()
            if ref is None:
                ref = out
            else:
                same &= all(np.array_equal(a, b) for a, b in zip(ref, out))
        bad += not same
        print(f"N={N} T={T} K={K} L={L}: {rounds} x {iters} fused iterations against the separate launch: {'bitwise equal' if same else 'DIFFERENT'}; loss {ref[0][0]:.5f} -> {ref[0][-1]:.5f}", flush=True)
finally:
    stop = True
    th.join()
print(f"{total} H updates ran inside the C3 launch; {bad} shapes differ")
sys.exit(1 if bad else 0)
