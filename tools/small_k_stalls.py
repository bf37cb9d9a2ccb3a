"""Where does a loop of very short iterations lose time?  cmf_iterate's host stamps (the moment each loss reached the host) at
configs[0]'s shape: the largest gaps between consecutive iterations, batch by batch.    python tools/small_k_stalls.py [N T K L]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmf_jl_amd as cmf  # noqa: E402

N, T, K, L = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (500, 2000, 5, 10)
opts = dict(a.split("=") for a in sys.argv[5:])  # option=value ...
data = cmf.gen_synthetic(N=N, T=T, seed=1234)
W0, H0 = cmf.init_rand(data, L=L, K=K, seed=0)
for h in range(2):
    rule = cmf.MultUpdate(data, W0, H0)
    for k, v in opts.items():
        rule.set_option(k, int(v))
    rule.iterate(30)
    rule.synchronize()
    for rep in range(4):
        t0 = time.perf_counter()
        _, st = rule.iterate(1000, stamps=True)
        rule.synchronize()
        dt = time.perf_counter() - t0
        d = np.diff(np.asarray(st))
        worst = np.argsort(d)[-3:][::-1]
        print(f"handle {h} batch {rep}: {1e3 * dt:.2f} ms for 1000; median step {1e6 * np.median(d):.1f} us; first stamp {1e6 * st[0]:.0f} us; "
              + "largest steps " + ", ".join(f"#{i + 1}: {1e6 * d[i]:.0f} us" for i in worst), flush=True)
    rule.close()
