"""Stand-alone kernel timings (cmf_time_kernel) at a given T: python tools/time_kernels.py [T] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmf_jl_amd as cmf
T = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
N, K, L = 2000, 32, 20
data = cmf.gen_synthetic(N=N, T=T, seed=1234)
W0, H0 = cmf.init_rand(data, L=L, K=K, seed=0)
rule = cmf.MultUpdate(data, W0, H0)
for variant in (2, 3):
    rule.set_option("conv_kernel", variant)
    for name in ("conv", "conv_t", "conv_loss", "conv_loss_store"):
        ms = sorted(rule.time_kernel(name, reps)[0] for _ in range(5))
        print(f"T={T} conv_kernel={variant} {name:16s} min {ms[0]:.4f} median {ms[2]:.4f} ms", flush=True)
for name in ("hxt", "transconv"):
    ms = sorted(rule.time_kernel(name, reps)[0] for _ in range(5))
    print(f"T={T} {name:16s} min {ms[0]:.4f} median {ms[2]:.4f} ms", flush=True)
