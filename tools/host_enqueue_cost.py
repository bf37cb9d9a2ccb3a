#!/usr/bin/env python3
"""Host cost per iteration of a one-process group of R shards when the device work is negligible (tiny T): how long the
host needs to get one sharded iteration enqueued (launches, hipSetDevice, events, collective calls) -- with the calling
thread enqueueing every shard itself ("caller"), and with one enqueue worker per shard ("threads", the default; the calling
thread then only posts and polls the loss words).  Wall time per iteration of a long pipelined cmf_iterate batch, so it
INCLUDES the kernels' minimum durations on the one GPU all shards share here; `calling thread` is what the library itself
measured for the thread that drives the fit (counters enqueue_ns / enqueue_iters: building and enqueueing -- or posting --
one iteration), `busiest worker` the time the slowest enqueue worker spent inside its jobs per iteration (worker_ns).
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
    for tr, name in ((_lib.CMF_COMM_LOOPBACK, "loopback, shared stream"), (_lib.CMF_COMM_LOOPBACK_STREAMS, "loopback, stream per shard"),
                     (_lib.CMF_COMM_PEER, "peer, stream per shard")):
        for threads in ((0,) if tr == _lib.CMF_COMM_LOOPBACK or R == 1 else (0, 1)):
            rule = cmf.MultUpdate(data, W0, H0, devices=[0] * R, transport=tr)
            rule.set_option("enqueue_threads", threads)
            rule.iterate(20)
            n = 300
            c0 = [rule.counter(k) for k in ("enqueue_ns", "enqueue_iters", "worker_ns")]
            t0 = time.perf_counter()
            rule.iterate(n)
            dt = (time.perf_counter() - t0) / n
            c1 = [rule.counter(k) for k in ("enqueue_ns", "enqueue_iters", "worker_ns")]
            its = max(1, c1[1] - c0[1])
            print(f"R={R} {name:27s} enqueue={'threads' if threads else 'caller ':7s}: {1e6 * dt:8.1f} us per iteration "
                  f"({1e6 * dt / R:6.1f} per shard); calling thread {1e-3 * (c1[0] - c0[0]) / its:7.1f} us, "
                  f"busiest worker {1e-3 * (c1[2] - c0[2]) / n:7.1f} us", flush=True)
            rule.close()
