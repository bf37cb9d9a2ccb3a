"""The two forms of the H halo exchange on a T-sharded group, rehearsed on ONE GPU (8 loopback shards of config 2 share the device, so
the absolute times are not a node's; the comparison is what one GPU can give): option "halo_in_allreduce" = 1 (round 6: the halos in
the tail of the W-phase all-reduce, every shard with a left neighbour updating the L-1 columns in front of its own: ONE collective per
iteration) against 0 (the all-gather of rounds 1-5).  Prints ms per iteration, the collectives issued per iteration, and the
per-kernel means of shard 0 (the redundant work shows in conv_t and transconv).

    python tools/halo_in_allreduce_cost.py [shards=8] [T=50000] [steps=30]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmf_jl_amd as cmf  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
N, K, L = 2000, 32, 20
data = cmf.gen_synthetic(N=N, T=T, seed=1234)
W0, H0 = cmf.init_rand(data, L=L, K=K, seed=0)
for transport, tname in ((2, "loopback, one stream"), (3, "loopback, a stream per shard")):
    for form in (1, 0, 1, 0):
        rule = cmf.MultUpdate(data, W0, H0, devices=[0] * R, transport=transport)
        rule.set_option("halo_in_allreduce", form)
        rule.iterate(10)
        rule.synchronize()
        c0 = (rule.counter("allreduce_calls"), rule.counter("allgather_calls"))
        t0 = time.perf_counter()
        ls = rule.iterate(steps)
        rule.synchronize()
        dt = (time.perf_counter() - t0) / steps
        c1 = (rule.counter("allreduce_calls"), rule.counter("allgather_calls"))
        rule.set_option("profile", 1)
        rule.iterate(5)
        k = {nm: rule.kernel_times(nm)[0] for nm in ("conv_t", "transconv", "conv_loss_store", "hxt")}
        print(f"{R} shards of T/{R} = {T // R} columns, {tname}, halo_in_allreduce = {form}: {1e3 * dt:.4f} ms per iteration; per iteration "
              f"{(c1[0] - c0[0]) / steps:.2f} all-reduces, {(c1[1] - c0[1] - 1) / steps:.2f} all-gathers (+ 1 to flush the last loss); shard 0: "
              + ", ".join(f"{nm} {v:.4f}" for nm, v in k.items()) + f" ms; loss {ls[-1]:.6f}", flush=True)
        rule.close()
