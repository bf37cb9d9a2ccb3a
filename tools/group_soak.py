#!/usr/bin/env python3
"""Soak of the group engine on one GPU: a small problem as 8 shards, thousands of pipelined iterations (cmf_iterate batches of
random lengths, now and then a call-by-call stretch, the overlap form switched on and off) on every transport that has a
stream per shard, with the enqueue workers -- the factors and every loss compared BITWISE with the shared-stream loopback group
run through the same schedule.  A hand-off that ever delivered stale data, a lost wake-up or a mis-ordered collective would show.
    python3 tools/group_soak.py [total_iterations=6000]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cmf_jl_amd as cmf  # noqa: E402
from cmf_jl_amd import _lib  # noqa: E402

total = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
N, T, K, L, R = 96, 1100, 32, 20, 8
data = cmf.gen_synthetic(N=N, T=T, seed=1234)
W0, H0 = cmf.init_rand(data, L=L, K=K, seed=0)
rng = np.random.default_rng(7)
schedule, left = [], total
while left > 0:
    kind = rng.choice(["iterate", "iterate", "iterate", "calls", "overlap"])
    n = int(min(left, rng.integers(1, 60)))
    schedule.append((kind, n))
    left -= n if kind != "overlap" else 0


def run(tr):
    rule = cmf.MultUpdate(data, W0, H0, devices=[0] * R, transport=tr)
    losses, overlap = [], False
    t0 = time.perf_counter()
    for kind, n in schedule:
        if kind == "overlap":
            overlap = not overlap
            rule.set_overlap(overlap)
        elif kind == "calls":
            for _ in range(min(n, 5)):
                rule.update_motifs(l1W=0.01)
                losses.append(rule.update_feature_maps(l1H=0.01))
        else:
            losses += list(rule.iterate(n, l1W=0.01, l1H=0.01))
        if len(losses) % 997 < 60:  # keep the factors alive over thousands of iterations: re-seed them now and then
            rule.upload(W0, H0)
    W, H = rule.download()
    info = rule.comm_info()
    rule.close()
    return np.asarray(losses), W, H, time.perf_counter() - t0, info


ref = run(_lib.CMF_COMM_LOOPBACK)
print(f"reference: shared-stream loopback, {len(ref[0])} iterations in {ref[3]:.1f} s", flush=True)
for tr, name in ((_lib.CMF_COMM_LOOPBACK_STREAMS, "loopback-streams"), (_lib.CMF_COMM_PEER, "peer")):
    got = run(tr)
    same = all(np.array_equal(a, b) for a, b in zip(ref[:3], got[:3]))
    print(f"{name:17s}: {len(got[0])} iterations in {got[3]:.1f} s, bitwise equal to the reference: {same}   [{got[4]}]", flush=True)
    if not same:
        sys.exit(1)
print("ok")
