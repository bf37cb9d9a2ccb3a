"""Worker for the distributed parameter_sweep test: python _sweep_worker.py <out.npz>  (gloo ranks sharing GPU 0)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist

    dist.init_process_group("gloo")
    import cmf_jl_amd as cmf

    data = cmf.gen_synthetic(N=40, T=300, seed=1234)
    res = cmf.parameter_sweep(data, L_vals=(5, 8), K_vals=(2, 3), alg_vals=(":mult",), max_itr=6, seed=0,
                              check_convergence=False, device=0)
    if dist.get_rank() == 0:
        np.savez(sys.argv[1], keys=np.array([[L, K] for (L, K, _) in res]),
                 **{f"loss_{L}_{K}": r.loss_hist for (L, K, _), r in res.items()},
                 **{f"W_{L}_{K}": r.W for (L, K, _), r in res.items()})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
