"""Few-component shapes: ms per iteration of the MU rule in its default form and in the Gram forms (option gram = 1 | 2).
    python tools/small_k_forms.py [N T K L gram,gram,..]..."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmf_jl_amd as cmf  # noqa: E402

jobs = [((250, 50000, 5, 20), (0, 1, 2)), ((500, 2000, 5, 10), (0, 1, 2))]
if len(sys.argv) > 5:
    a = sys.argv[1:]
    jobs = [(tuple(int(x) for x in a[i:i + 4]), tuple(int(x) for x in a[i + 4].split(","))) for i in range(0, len(a), 5)]
for (N, T, K, L), grams in jobs:
    data = cmf.gen_synthetic(N=N, T=T, seed=1234)
    W0, H0 = cmf.init_rand(data, L=L, K=K, seed=0)
    ref = None
    for gram in grams:
        rule = cmf.MultUpdate(data, W0, H0)
        if gram >= 0:
            rule.set_option("gram", gram)
        ls = rule.iterate(30)
        rule.synchronize()
        res = []
        for rep in range(3):
            n = 300
            t0 = time.perf_counter()
            rule.iterate(n)
            rule.synchronize()
            res.append(1e3 * (time.perf_counter() - t0) / n)
        W, H = rule.download()
        if ref is None:
            ref = (W, H, ls)
        print((N, T, K, L), "gram", gram, " ".join(f"{r:.4f}" for r in res), "ms/iter; rel W to the first form", np.linalg.norm(W - ref[0]) / np.linalg.norm(ref[0]), flush=True)
        rule.close()
