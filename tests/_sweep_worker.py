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

    # replicas: rank r fits on GPU r when the box has one per rank, else the ranks share GPU 0
    ndev = cmf.load_library().cmf_device_count()
    device = dist.get_rank() if ndev >= dist.get_world_size() else 0
    data = cmf.gen_synthetic(N=40, T=300, seed=1234, device=device)
    algs = tuple(sys.argv[2].split(",")) if len(sys.argv) > 2 else (":mult",)
    res = cmf.parameter_sweep(data, L_vals=(5, 8), K_vals=(2, 3), alg_vals=algs, max_itr=6, seed=0,
                              check_convergence=False, device=device)
    if dist.get_rank() == 0:
        tag = lambda L, K, a: f"{L}_{K}_{a.lstrip(':')}"  # noqa: E731
        np.savez(sys.argv[1], keys=np.array([tag(L, K, a) for (L, K, a) in res]), device=np.asarray(device),
                 **{f"loss_{tag(L, K, a)}": r.loss_hist for (L, K, a), r in res.items()},
                 **{f"W_{tag(L, K, a)}": r.W for (L, K, a), r in res.items()},
                 **{f"H_{tag(L, K, a)}": r.H for (L, K, a), r in res.items()})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
