#!/usr/bin/env python3
"""How far the Gram-sum loss (option gram = 2) is from the oracle's loss_hist, per shape (GPU box): decides the bar the
tests can state for it."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cmf_jl_amd as cmf  # noqa: E402
from oracle import cmf_oracle as oracle  # noqa: E402

reg = dict(l1W=0.1, l2W=0.5, l1H=0.1, l2H=0.2)
for (N, T, K, L) in [(48, 300, 4, 8), (130, 700, 32, 20), (37, 150, 33, 7), (6, 3, 2, 5), (20, 200, 6, 40), (500, 2000, 5, 10)]:
    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=min(L, 20), seed=1234)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=0)
    Wr, Hr, lr, _ = oracle.fit_mult(data, W0, H0, max_itr=10, check_convergence=False, **reg)
    for gram in (1, 2):
        rule = cmf.MultUpdate(data, W0, H0)
        rule.set_option("gram", gram)
        lg = []
        for _ in range(10):
            rule.update_motifs(l1W=reg["l1W"], l2W=reg["l2W"])
            lg.append(rule.update_feature_maps(l1H=reg["l1H"], l2H=reg["l2H"]))
        rule.close()
        err = np.abs(np.asarray(lg) - lr[1:]) / lr[1:]
        print(f"N={N} T={T} K={K} L={L} gram={gram}: loss {lr[-1]:.4f} max rel err {err.max():.2e}  (1e-6/loss^2 = {1e-6 / lr[-1] ** 2:.1e})", flush=True)
