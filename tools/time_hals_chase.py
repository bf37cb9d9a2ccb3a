"""HALS iterations at config-5 sizes with the residual conv chasing the H row pipeline (option "hals_chase" = per cent of its tile
rows; 0 = off): ms per iteration, the pipeline's and the conv launches' own durations, and the results against chase = 0 (the same
arithmetic per tile; the two launches cut other tiles of their grids' tails into pieces, so sums agree to rounding).

    python tools/time_hals_chase.py [pct ...]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmf_jl_amd as cmf  # noqa: E402

N, T, K, L = 2000, 50000, 32, 20
data = cmf.gen_synthetic(N=N, T=T, seed=1234)
W0, H0 = cmf.init_rand(data, L=L, K=K, seed=0)
ref = None
for arg in sys.argv[1:] or ["0", "30", "40", "45", "50", "55", "60"]:
    pct, _, pullers = arg.partition(":")  # "45" or "45:3" (at most 3 puller workgroups per row: more CUs for the chasing launch)
    pct = int(pct)
    rule = cmf.HALSUpdate(data, W0, H0)
    if pullers:
        rule.set_option("hals_persist", int(pullers))
    rule.set_option("hals_chase", pct)
    losses = []
    for _ in range(3):
        rule.update_motifs()
        losses.append(rule.update_feature_maps())
    Wd, Hd = rule.download()
    if ref is None:
        ref = (losses, Wd, Hd)
    dl = max(abs(a / b - 1) for a, b in zip(losses, ref[0]))
    dh = np.linalg.norm(Hd - ref[2]) / np.linalg.norm(ref[2])
    dw = np.linalg.norm(Wd - ref[1]) / np.linalg.norm(ref[1])
    for _ in range(40):  # (a cold chip runs the first tenths of a second at a lower clock)
        rule.update_motifs()
        rule.update_feature_maps()
    n = 40
    t0 = time.perf_counter()
    for _ in range(n):
        rule.update_motifs()
        loss = rule.update_feature_maps()
    dt = (time.perf_counter() - t0) / n
    rule.set_option("profile", 1)
    for _ in range(4):
        rule.update_motifs()
        rule.update_feature_maps()
    pipe, _ = rule.kernel_times("hals_h_pipeline")
    conv, nconv = rule.kernel_times("conv_resid")
    wsw, _ = rule.kernel_times("hals_w_sweep")
    hh, _ = rule.kernel_times("hxt_hh")
    print(f"hals_chase={arg:>5s}: {1e3 * dt:.3f} ms per HALS iteration; lag correlations {hh:.3f} ms, W sweep {wsw:.3f} ms, pipeline {pipe:.3f} ms, conv launches {conv:.3f} ms (mean of {nconv}); "
          f"first 3 iterations against the first run: loss {dl:.1e}, W {dw:.1e}, H {dh:.1e}; reruns {rule.counter('hals_pipeline_reruns')}; loss {loss:.6f}",
          flush=True)
    rule.close()
