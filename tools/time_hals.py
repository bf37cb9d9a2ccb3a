"""Wall time of HALS iterations at config-5 sizes with the conv kernel variant forced (0 = per mode, 2 = 128x128 tiles,
3 = one-wave tiles): python tools/time_hals.py [variant ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmf_jl_amd as cmf

N, T, K, L = 2000, 50000, 32, 20
data = cmf.gen_synthetic(N=N, T=T, seed=1234)
W0, H0 = cmf.init_rand(data, L=L, K=K, seed=0)
for variant in [int(v) for v in sys.argv[1:]] or [0, 2, 3]:
    rule = cmf.HALSUpdate(data, W0, H0)
    rule.set_option("conv_kernel", variant)
    for _ in range(2):
        rule.update_motifs()
        rule.update_feature_maps()
    t0 = time.perf_counter()
    n = 8
    for _ in range(n):
        rule.update_motifs()
        loss = rule.update_feature_maps()
    dt = (time.perf_counter() - t0) / n
    print(f"conv_kernel={variant}: {1e3 * dt:.3f} ms per HALS iteration, loss {loss:.6f}", flush=True)
    rule.close()
