"""Worker for the multi-process sharding tests (launched by tests/test_sharded.py).

usage: python _dist_worker.py <engine: cpu|hip> <out.npz> <N> <T> <K> <L> <iters> <reg:0|1>
Env: RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT (rendezvous on 127.0.0.1); CMF_TEST_BACKEND=gloo|nccl;
CMF_TEST_OVERLAP=0|1; CMF_TEST_MODE=calls (update_motifs! / update_feature_maps! per iteration) | iterate
(one cmf_iterate batch) | fit (the host's fit loop / cmf_fit).
  cpu: the Python mirror of the library's group protocol on the numpy engine (no GPU needed)
  hip: cmf_jl_amd.sharded.ShardedMultUpdate = the library's own group iteration, one process per shard
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
    mode = os.environ.get("CMF_TEST_MODE", "calls")
    overlap = os.environ.get("CMF_TEST_OVERLAP", "0") == "1"
    if backend == "nccl":  # RCCL: one rank per GPU (the single-rank case is what a one-GPU box can run)
        import torch

        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend)
    rank = dist.get_rank()
    from oracle import cmf_oracle as oracle

    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=min(L, 20), seed=1234)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=0)
    kw = dict(l1W=0.1, l2W=0.5, l1H=0.1, l2H=0.2) if reg else dict(l1W=0.0, l2W=0.0, l1H=0.0, l2H=0.0)
    info = ""
    if engine.startswith("cpu_pgd"):  # the CPU mirror of the PGD rule on a T-sharded group (tests/shard_rules_cpu.py)
        from shard_rules_cpu import ProtocolShardedPGD

        mask = (np.random.default_rng(5).uniform(size=data.shape) > 0.25).astype(float) if "masked" in engine else None
        rule = ProtocolShardedPGD(data, W0, H0, mask=mask, loss="abs" if "abs" in engine else "square")
        cW = cH = "unitnorm" if "unitnorm" in engine else "nonneg"
        losses = [rule.compute_loss()]
        for _ in range(iters):
            rule.update_motifs(constr=cW)
            losses.append(rule.update_feature_maps(constr=cH))
        W, H = rule.download()
        if rank == 0:
            np.savez(out, W=W, H=H, loss_hist=np.asarray(losses), steps=np.asarray([rule.stepW, rule.stepH]), bounds=np.asarray(rule.bounds),
                     info=np.asarray(""))
        dist.barrier()
        dist.destroy_process_group()
        return
    if engine == "cpu_gram":
        from shard_rules_cpu import ProtocolShardedGram

        rule = ProtocolShardedGram(data, W0, H0)
    elif engine == "cpu":
        from shard_engine_cpu import OracleShardEngine
        from shard_protocol_cpu import ProtocolShardedMultUpdate

        rule = ProtocolShardedMultUpdate(data, W0, H0, OracleShardEngine, overlap=overlap,
                                         halo_in_allreduce=os.environ.get("CMF_TEST_HALO_IN_AR", "1") == "1")
    elif engine.startswith("hip_pgd"):
        import cmf_jl_amd as cmf
        from cmf_jl_amd.sharded import ShardedPGDUpdate

        rule = ShardedPGDUpdate(data, W0, H0, device=int(os.environ.get("LOCAL_RANK", "0")),
                                transport=os.environ.get("CMF_TEST_TRANSPORT") or None)
        lf = None
        if engine.endswith("masked"):
            lf = cmf.MaskedLoss(cmf.SquareLoss(), (np.random.default_rng(5).uniform(size=data.shape) > 0.25).astype(float))
        kwW, kwH = {}, {}
        if engine.endswith("unitnorm"):  # the component norms of H run over ALL shards (K doubles gathered per H phase)
            kwW, kwH = dict(constrW=cmf.UnitNormConstraint()), dict(constrH=cmf.UnitNormConstraint())
        losses = [rule.compute_loss()]
        for _ in range(iters):
            rule.update_motifs(loss_func=lf, **kwW)
            losses.append(rule.update_feature_maps(loss_func=lf, **kwH))
        W, H = rule.download()
        steps = rule.steps
        rule.close()
        if rank == 0:
            np.savez(out, W=W, H=H, loss_hist=np.asarray(losses), steps=np.asarray(steps), bounds=np.asarray(rule.bounds), info=np.asarray(""))
        dist.barrier()
        dist.destroy_process_group()
        return
    else:
        from cmf_jl_amd.sharded import ShardedMultUpdate

        rule = ShardedMultUpdate(data, W0, H0, device=int(os.environ.get("LOCAL_RANK", "0")), overlap=overlap,
                                 transport=os.environ.get("CMF_TEST_TRANSPORT") or None,
                                 fallback_to_host=os.environ.get("CMF_TEST_FALLBACK", "0") == "1")
        info = rule.comm_info() + (" FALLBACK" if rule.transport_fallback else "") + f" halo_in_allreduce={rule.counter('halo_in_allreduce')}"
        want = float(np.linalg.norm(data))
        assert abs(rule.data_norm - want) <= 1e-9 * want, f"rank {rank}: data_norm {rule.data_norm} != {want}"
    losses = [rule.compute_loss()]
    coll0 = dict(getattr(rule, "collectives", {}))  # (the CPU mirror counts the collectives it issues)
    if mode == "iterate":
        losses += list(rule.iterate(iters, **kw))
    elif mode == "fit_timed" and engine == "hip":
        # a finite max_time takes cmf_fit's synchronous loop, in which every rank follows rank 0's clock (an all-gather
        # of the iteration's duration) so that all ranks leave the loop together
        lh, th, _ = rule.fit_native(iters, 1e6, False, 3, 1e-4, False, **kw)
        assert len(lh) == iters + 1 and np.all(np.diff(th) > 0)
        allth = [None] * dist.get_world_size()
        dist.all_gather_object(allth, [float(x) for x in th])
        assert all(t == allth[0] for t in allth), "ranks disagree on time_hist"
        np.testing.assert_allclose(lh[0], losses[0], rtol=1e-12)
        losses = list(lh)
    elif mode == "fit" and engine == "hip":
        lh, th, _ = rule.fit_native(iters, np.inf, False, 3, 1e-4, False, **kw)
        np.testing.assert_allclose(lh[0], losses[0], rtol=1e-12)
        losses = list(lh)
    else:
        for _ in range(iters):
            rule.update_motifs(l1W=kw["l1W"], l2W=kw["l2W"])
            losses.append(rule.update_feature_maps(l1H=kw["l1H"], l2H=kw["l2H"]))
    coll1 = dict(getattr(rule, "collectives", {}))
    W, H = rule.download()
    rule.close()
    if rank == 0:
        np.savez(out, W=W, H=H, loss_hist=np.asarray(losses), bounds=np.asarray(rule.bounds), info=np.asarray(info),
                 all_reduces=np.asarray(coll1.get("all_reduce", 0) - coll0.get("all_reduce", 0)),
                 all_gathers=np.asarray(coll1.get("all_gather", 0) - coll0.get("all_gather", 0)))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
