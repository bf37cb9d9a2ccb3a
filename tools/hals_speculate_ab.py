"""HALS call by call with and without the speculated W-phase contraction (option "speculate"), alternating in one process:
ms per iteration at config 5 and at two of the reference's own small shapes.    python tools/hals_speculate_ab.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmf_jl_amd as cmf  # noqa: E402

for (N, T, K, L) in [(2000, 50000, 32, 20), (500, 2000, 5, 10), (250, 50000, 5, 20)]:
    data = cmf.gen_synthetic(N=N, T=T, seed=1234)
    W0, H0 = cmf.init_rand(data, L=L, K=K, seed=0)
    rule = cmf.HALSUpdate(data, W0, H0)
    for _ in range(30):
        rule.update_motifs()
        rule.update_feature_maps()
    res = []
    for rep in range(3):
        for spec in (1, 0):
            rule.set_option("speculate", spec)
            rule.update_motifs()
            rule.update_feature_maps()
            n = 40 if T > 10000 else 300
            t0 = time.perf_counter()
            for _ in range(n):
                rule.update_motifs()
                rule.update_feature_maps()
            res.append((spec, 1e3 * (time.perf_counter() - t0) / n))
    print((N, T, K, L), " ".join(f"speculate={s}: {ms:.4f} ms" for s, ms in res), "hits", rule.counter("speculated_contractions"), flush=True)
    rule.close()
