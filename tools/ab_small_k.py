"""A / B of two builds of the library in ONE process run after the other on the same GPU box (boxes differ by a few per cent):
    python tools/ab_small_k.py <other tree root> [N T K L]...
Each tree is imported in a child process (a process loads one libcmf_hip.so); the median cmf_iterate step of either, alternating."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import cmf_jl_amd as cmf
N, T, K, L = (int(a) for a in sys.argv[2:6])
data = cmf.gen_synthetic(N=N, T=T, seed=1234)
W0, H0 = cmf.init_rand(data, L=L, K=K, seed=0)
rule = cmf.MultUpdate(data, W0, H0)
rule.iterate(50); rule.synchronize()
meds = []
for rep in range(4):
    _, st = rule.iterate(1000, stamps=True)
    meds.append(1e6 * float(np.median(np.diff(np.asarray(st)))))
ks = {nm: min(rule.time_kernel(nm, reps=40)[0] for _ in range(3)) for nm in ("hxt", "transconv", "conv_t", "conv_loss_store")}
print(" ".join(f"{m:.1f}" for m in meds) + " | kernels (us, best of 3 x 40): " + " ".join(f"{k} {1e3 * v:.2f}" for k, v in ks.items()))
rule.close()
'''
other = os.path.abspath(sys.argv[1])
a = [int(x) for x in sys.argv[2:]] or [250, 50000, 5, 20, 500, 2000, 5, 10]
for i in range(0, len(a), 4):
    shape = [str(x) for x in a[i:i + 4]]
    for rep in range(3):
        for name, root in (("this tree", HERE), ("other tree", other)):
            out = subprocess.run([sys.executable, "-c", CHILD, root] + shape, capture_output=True, text=True)
            print(f"N,T,K,L={','.join(shape)} {name:10s}: median step (us) {out.stdout.strip() or out.stderr[-300:]}", flush=True)
