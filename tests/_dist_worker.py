"""Worker for the multi-process sharding tests (launched by tests/test_sharded.py).

usage: python _dist_worker.py <engine: cpu|hip> <out.npz> <N> <T> <K> <L> <iters> <reg:0|1>
Env: RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT (rendezvous on 127.0.0.1); CMF_TEST_BACKEND=gloo|nccl.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    engine, out = sys.argv[1], sys.argv[2]
    N, T, K, L, iters, reg = (int(x) for x in sys.argv[3:9])
    import torch.distributed as dist

    backend = os.environ.get("CMF_TEST_BACKEND", "gloo")
    if backend == "nccl":  # RCCL: one rank per GPU (the single-rank case is what a one-GPU box can run)
        import torch

        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    from oracle import cmf_oracle as oracle
    import cmf_jl_amd as cmf
    from cmf_jl_amd.sharded import HipShardEngine, ShardedMultUpdate

    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=min(L, 20), seed=1234)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=0)
    kw = dict(l1W=0.1, l2W=0.5, l1H=0.1, l2H=0.2) if reg else {}
    if engine == "cpu":
        from shard_engine_cpu import OracleShardEngine as Eng
    else:
        Eng = HipShardEngine
    rule = ShardedMultUpdate(data, W0, H0, device=0, engine_cls=Eng, overlap=os.environ.get("CMF_TEST_OVERLAP", "0") == "1")
    opt = cmf.AlternatingOptimizer(rule, iters, np.inf)
    res = cmf.fit(opt, data, L, K, W0, H0, check_convergence=False, **kw)
    rule.close()
    if rank == 0:
        np.savez(out, W=res.W, H=res.H, loss_hist=res.loss_hist, bounds=np.asarray(rule.bounds))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
