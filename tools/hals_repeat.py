"""Bitwise repeatability of the persistent HALS H pipeline: many identical H sweeps at config-5 sizes, every H compared with
the first (a hand-off that ever delivered stale data would show as a mismatch): python tools/hals_repeat.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cmf_jl_amd as cmf

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
N, T, K, L = 2000, 50000, 32, 20
data = cmf.gen_synthetic(N=N, T=T, seed=1234)
W0, H0 = cmf.init_rand(data, L=L, K=K, seed=0)
rule = cmf.HALSUpdate(data, W0, H0)
rule.update_motifs()
W1, _ = rule.download()
ref = None
bad = 0
for it in range(n):
    rule.upload(W1, H0)
    loss = rule.update_feature_maps()
    _, H = rule.download()
    rule.update_motifs()  # reads the residual the conv behind the sweep stored -- most of it by tiles CHASING the sweep (option
    W2, _ = rule.download()  # hals_chase): a tile that ever read a column of H before it was final shows up in the loss and here
    if ref is None:
        ref, ref_loss, ref_W2 = H.copy(), loss, W2.copy()
    elif not (np.array_equal(H, ref) and loss == ref_loss and np.array_equal(W2, ref_W2)):
        bad += 1
        d = np.argwhere(H != ref)
        print(f"run {it}: {len(d)} entries differ, first at {d[:3].tolist()}, max |diff| {np.abs(H - ref).max():.3e}, loss {loss} vs {ref_loss}", flush=True)
print(f"{n} H sweeps (+ the chasing residual conv and the W sweep that reads its residual) at N={N}, T={T}, K={K}, L={L}: {bad} differ from the first", flush=True)

# the small shapes of the parity tests, whole iterations (W and H phases), 60 repeats each
shapes = [(12, 40, 3, 6), (48, 300, 4, 8), (30, 70, 2, 1), (9, 5, 2, 8), (130, 700, 32, 20), (37, 200, 33, 7), (200, 1500, 5, 10),
          (150, 900, 64, 20), (64, 400, 40, 30), (2000, 1200, 32, 20)]
rng = np.random.default_rng(0)
for (N, T, K, L) in shapes:
    data = rng.random((N, T)); W0 = rng.random((K, N, L)); H0 = rng.random((K, T))
    ref = None; bad = 0
    for it in range(60):
        rule = cmf.HALSUpdate(data, W0, H0)
        rule.update_motifs(l1W=0.1, l2W=0.5); loss = rule.update_feature_maps(l1H=0.1, l2H=0.2)
        rule.update_motifs(); loss2 = rule.update_feature_maps()
        W, H = rule.download(); rule.close()
        if ref is None:
            ref = (W.copy(), H.copy(), loss, loss2)
        elif not (np.array_equal(W, ref[0]) and np.array_equal(H, ref[1]) and loss == ref[2] and loss2 == ref[3]):
            bad += 1
            print(f"  shape {(N, T, K, L)} run {it}: differs (W {np.abs(W - ref[0]).max():.2e}, H {np.abs(H - ref[1]).max():.2e}, loss {loss} {loss2} vs {ref[2]} {ref[3]})", flush=True)
    print(f"shape {(N, T, K, L)}: {bad} of 60 repeats differ")
