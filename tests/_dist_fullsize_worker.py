"""Worker: config-2-size MU iterations on the T-sharded path, one process per shard (gloo ranks sharing GPU 0, the
library's host-callback transport): python _dist_fullsize_worker.py <out.npz> <iters>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out, iters = sys.argv[1], int(sys.argv[2])
    import torch.distributed as dist

    dist.init_process_group("gloo")
    import cmf_jl_amd as cmf
    from cmf_jl_amd.sharded import ShardedMultUpdate

    N, T, K, L = 2000, 50000, 32, 20
    data = cmf.gen_synthetic(N=N, T=T, seed=1234, device=0)
    W0, H0 = cmf.init_rand(data, L=L, K=K, seed=0, device=0)
    rule = ShardedMultUpdate(data, W0, H0, device=0)
    losses = [rule.compute_loss()] + list(rule.iterate(iters))
    W, H = rule.download()
    rule.close()
    if dist.get_rank() == 0:
        np.savez(out, W=W, H=H, loss_hist=np.asarray(losses))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
