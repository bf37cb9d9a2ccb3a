"""RCCL across DISTINCT devices: the tests a one-GPU box cannot run.

Every test here is gated on cmf_device_count() >= n and skips otherwise, so on the one-GPU box of the round-end run they
skip and on an N-GPU node they need nobody's help: the one-process group (cmf_create_multi: ncclCommInitAll, a stream per
device; an enqueue worker thread per device issuing its own RCCL calls, or the calling thread with grouped calls; the
opt-in peer transport) and the one-process-per-GPU group (cmf_create_shard + cmf_comm_init_rccl:
ncclCommInitRank with LOCAL_RANK = device) against the fp64 oracle and against the same partition on loopback shards of
GPU 0, call by call and as a cmf_iterate batch, with the all-reduce overlap form off and on.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from test_sharded import REG, frob_rel, oracle_fit, run_ranks

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _need_devices(n):
    import cmf_jl_amd as cmf

    have = cmf.load_library().cmf_device_count()
    if have < n:
        pytest.skip(f"needs {n} HIP devices, this box has {have}")


def _case(form, transport, who, what, n, *values):
    """A device-count-gated case.  Its id names, in this order, the launch form (one-process | per-process | bench-plain-launch),
    the transport (rccl | peer), who enqueues (workers | caller | ranks), what runs, and the device count -- so that the first
    run on a node reads, failure by failure, as "which form on which transport" (tests/test_bench_routing.py checks the ids)."""
    return pytest.param(*values, id=f"{form}.{transport}.{who}.{what}.n{n}")


def _run(rule, mode, iters, kw):
    losses = [rule.compute_loss()]
    if mode == "calls":
        for _ in range(iters):
            rule.update_motifs(l1W=kw["l1W"], l2W=kw["l2W"])
            losses.append(rule.update_feature_maps(l1H=kw["l1H"], l2H=kw["l2H"]))
    else:
        losses += list(rule.iterate(iters, **kw))
    W, H = rule.download()
    return np.asarray(losses), W, H


@pytest.mark.gpu
@pytest.mark.parametrize("n,overlap", [_case("one-process", "rccl", "workers", "overlap" if ov else "plain", n, n, ov)
                                       for ov in (False, True) for n in (2, 4, 8)])
def test_one_process_group_over_rccl(oracle, n, overlap):
    """MultUpdate(devices=range(n)) = cmf_create_multi over RCCL, vs the oracle (1e-4) and vs the same n-shard partition
    on loopback shards of GPU 0 (1e-5: same kernels, same shard shapes -- only the all-reduce's summation order differs)."""
    _need_devices(n)
    import cmf_jl_amd as cmf

    N, T, K, L, iters, reg = 130, 1800, 32, 20, 6, 1
    data, W0, H0, Wr, Hr, lr = oracle_fit(oracle, N, T, K, L, iters, reg)
    kw = REG
    ref = {}
    for mode in ("calls", "iterate"):
        rule = cmf.MultUpdate(data, W0, H0, devices=[0] * n)
        ref[mode] = _run(rule, mode, iters, kw)
        rule.close()
    for mode in ("calls", "iterate"):
        rule = cmf.MultUpdate(data, W0, H0, devices=list(range(n)))
        info = rule.comm_info()
        assert "transport=rccl" in info and f"nranks={n}" in info and f"local={n}" in info
        assert all(f"{r}@dev{r}" in info for r in range(n))
        if overlap:
            rule.set_overlap(True)
        losses, W, H = _run(rule, mode, iters, kw)
        rule.synchronize()
        rule.close()
        np.testing.assert_allclose(losses, lr, rtol=1e-4)
        assert frob_rel(W, Wr) < 1e-4 and frob_rel(H, Hr) < 1e-4
        np.testing.assert_allclose(losses, ref[mode][0], rtol=1e-5)
        assert frob_rel(W, ref[mode][1]) < 1e-5 and frob_rel(H, ref[mode][2]) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("n,transport,threads,overlap", [
    _case("one-process", tr, "workers" if th else "caller", "overlap" if ov else "plain", n, n, tr, th, ov)
    for tr, th, ov in (("rccl", 0, False), ("rccl", 0, True), ("peer", 1, False), ("peer", 0, True), ("peer", 1, True)) for n in (2, 4, 8)])
def test_one_process_group_forms_on_distinct_devices(oracle, n, transport, threads, overlap):
    """The forms of a one-process group that round 4 added, on DISTINCT devices: the calling thread enqueueing every shard
    with grouped RCCL calls (enqueue_threads = 0; the default -- a worker thread per device, RCCL's one-thread-per-device mode --
    is what test_one_process_group_over_rccl runs), the overlap form on its own communicators, and the peer transport (direct
    xGMI reads / writes between event fences) whose sums are in rank order: bitwise the loopback partition on GPU 0."""
    _need_devices(n)
    import cmf_jl_amd as cmf
    from cmf_jl_amd import _lib

    N, T, K, L, iters = 130, 1800, 32, 20, 6
    data, W0, H0, Wr, Hr, lr = oracle_fit(oracle, N, T, K, L, iters, 1)
    base = cmf.MultUpdate(data, W0, H0, devices=[0] * n)
    base.set_overlap(overlap)
    ref = _run(base, "iterate", iters, REG)
    base.close()
    rule = cmf.MultUpdate(data, W0, H0, devices=list(range(n)), transport=_lib.CMF_COMM_PEER if transport == "peer" else _lib.CMF_COMM_RCCL)
    rule.set_option("enqueue_threads", threads)
    rule.set_overlap(overlap)
    info = rule.comm_info()
    assert f"transport={transport}" in info and f"enqueue={'threads' if threads else 'caller'}" in info
    if transport == "rccl" and overlap:
        assert "lanes=2" in info
    losses, W, H = _run(rule, "iterate", iters, REG)
    ms, nbytes = rule.time_kernel("allreduce", reps=5)
    rule.close()
    print(f"{transport} n={n} threads={threads} overlap={overlap}: all-reduce of {nbytes / 1e6:.1f} MB alone {1e3 * ms:.0f} us")
    np.testing.assert_allclose(losses, lr, rtol=1e-4)
    assert frob_rel(W, Wr) < 1e-4 and frob_rel(H, Hr) < 1e-4
    if transport == "peer":
        np.testing.assert_array_equal(losses, ref[0])
        np.testing.assert_array_equal(W, ref[1])
        np.testing.assert_array_equal(H, ref[2])
    else:
        np.testing.assert_allclose(losses, ref[0], rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [_case("one-process", "rccl", "workers", "fit-and-shard-shapes", n, n) for n in (2, 4, 8)])
def test_one_process_group_fit_and_shard_shapes(oracle, n):
    """cmf_fit (pipelined and with the convergence test) on an RCCL group of distinct devices, a T that does not divide
    evenly, K not a multiple of 32."""
    _need_devices(n)
    import cmf_jl_amd as cmf

    N, T, K, L, iters = 70, 997, 5, 10, 10
    data, W0, H0, Wr, Hr, lr = oracle_fit(oracle, N, T, K, L, iters, 0)
    rule = cmf.MultUpdate(data, W0, H0, devices=list(range(n)))
    lh, th, early = rule.fit_native(iters, np.inf, False, 3, 1e-4, False)
    W, H = rule.download()
    rule.close()
    assert not early and len(lh) == iters + 1
    np.testing.assert_allclose(lh, lr, rtol=1e-4)
    assert frob_rel(W, Wr) < 1e-4 and frob_rel(H, Hr) < 1e-4
    _, _, lr2, _ = oracle.fit_mult(data, W0, H0, max_itr=200, check_convergence=True, patience=3, tol=2e-3)
    res = cmf.fit_cnmf(data, L=L, K=K, alg=":mult", max_itr=200, check_convergence=True, patience=3, tol=2e-3,
                       W_init=W0, H_init=H0, devices=list(range(n)))
    assert len(res.loss_hist) == len(lr2)
    np.testing.assert_allclose(res.loss_hist, lr2, rtol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("n,mode,overlap", [_case("per-process", "rccl", "ranks", mode + ("-overlap" if ov else ""), n, n, mode, ov)
                                            for n, mode, ov in ((2, "calls", False), (2, "iterate", True), (4, "iterate", False),
                                                                (8, "iterate", False), (2, "fit_timed", False))])
def test_one_process_per_gpu_over_rccl(oracle, tmp_path, n, mode, overlap):
    """n processes, rank r on GPU r (LOCAL_RANK=r), torch.distributed `nccl` as the rendezvous, the library's own
    ncclCommInitRank communicator for the data path: ShardedMultUpdate against the oracle."""
    _need_devices(n)
    N, T, K, L, iters = 130, 1800, 32, 20, 6
    out = str(tmp_path / "res.npz")
    got = run_ranks(n, "hip", out, N, T, K, L, iters, 1, backend="nccl", overlap=overlap, mode=mode, transport="rccl",
                    distinct_devices=True, timeout=600)
    info = str(got["info"])
    assert "transport=rccl" in info and f"nranks={n}" in info and "FALLBACK" not in info
    _, _, _, Wr, Hr, lr = oracle_fit(oracle, N, T, K, L, iters, 1)
    np.testing.assert_allclose(got["loss_hist"], lr, rtol=1e-4)
    assert frob_rel(got["W"], Wr) < 1e-4 and frob_rel(got["H"], Hr) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("n", [_case("bench-plain-launch", "rccl", "workers", "supervised-ladder", n, n) for n in (2, 8)])
def test_bench_plain_multi_gpu_launch(n):
    """`python bench.py --gpus n` with no launcher: the one-process form runs, RCCL reports n ranks in `comm`."""
    _need_devices(n)
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1",
                        "--sustain", "0", "--T", "8000"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == n and line["config"]["launch"] == "multi"
    assert line["comm"]["transport"] == "rccl" and line["comm"]["nranks"] == n and line["comm"]["devices"] == list(range(n))
    assert line["value"] > 0 and np.isfinite(line["config"]["loss_last"])


@pytest.mark.gpu
def test_bench_plain_multi_gpu_launch_reports_missing_devices():
    """More GPUs asked for than the box has: a non-zero exit with the reason on stderr, not a traceback or a hang."""
    import cmf_jl_amd as cmf

    have = cmf.load_library().cmf_device_count()
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 1), "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 2
    assert f"only {have} HIP device" in p.stderr.decode()
    assert p.stdout.decode().strip() == ""


@pytest.mark.gpu
@pytest.mark.parametrize("n,config,extras", [(4, 2, "1"), (2, 2, "gram"), (2, 5, "0")])
def test_bench_multi_form_rehearsal_on_one_gpu(n, config, extras):
    """The one-process form of `bench.py --gpus n` on THIS box: CMF_BENCH_DEVICES lists GPU 0 n times (loopback transport),
    so every line of the multi-GPU branch of bench.py runs -- routing, group construction, timing over all shards' streams,
    the comm record, the opt-in Gram side measurement, HALS replicas -- except RCCL itself."""
    env = dict(os.environ, CMF_BENCH_DEVICES=",".join(["0"] * n), CMF_BENCH_GROUP_EXTRAS=extras)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    args = ["--gpus", str(n), "--steps", "3", "--warmup", "1", "--sustain", "0", "--config", str(config)]
    if config == 2:
        args += ["--T", "8000"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == n and line["config"]["launch"] == "multi" and line["value"] > 0
    if config == 2:
        assert line["comm"]["transport"] == "loopback" and line["comm"]["nranks"] == n and line["comm"]["devices"] == [0] * n
        assert line["scaling"] == "strong" and line["roofline"]["frac"] > 0
        assert line["comm"]["allreduce"]["avg_ms"] > 0 and line["comm"]["allreduce"]["bytes"] > 1e6
        ge = line["group_extras"]
        assert "error" not in ge and ge["ms_per_step_gram"] > 0
        if extras == "1":
            assert ge["ms_per_step_gram_overlap"] > 0
            assert abs(ge["loss_last_gram"] - ge["loss_last_gram_overlap"]) <= 1e-6 * ge["loss_last_gram"]
        else:
            assert "ms_per_step_gram_overlap" not in ge
    else:
        assert line["comm"]["mode"] == "replicas" and line["comm"]["nranks"] == n and line["scaling"] == "weak"
        assert line["roofline"]["bound"] == "dependency-latency"


@pytest.mark.gpu
@pytest.mark.parametrize("transport", ["host", "rccl"])
def test_bench_ranks_form_rehearsal_on_one_gpu(transport):
    """The launcher form (`torch.distributed.run --nproc-per-node 2 bench.py --gpus 2`) on THIS box: two gloo ranks share
    GPU 0.  transport = host: the library's collectives go through the callbacks; transport = rccl: RCCL cannot put two ranks
    of a communicator on one device, every rank learns of it in the agreed attach and all fall back together -- the run
    still delivers its line, and `comm` says what happened."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, CMF_DIST_BACKEND="gloo", CMF_TRANSPORT=transport, OMP_NUM_THREADS="4")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--sustain", "0",
           "--T", "8000"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # rank 0 prints ONE line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["launch"] == "ranks" and line["value"] > 0
    assert line["comm"]["mode"] == "one-process-per-gpu" and line["comm"]["transport"] == "callbacks" and line["comm"]["nranks"] == 2
    assert line["comm"]["allreduce"]["avg_ms"] > 0
    assert ("fallback_from_rccl" in line["comm"]) == (transport == "rccl")


@pytest.mark.gpu
@pytest.mark.parametrize("transport,threads", [("peer", "1"), ("peer", "0"), ("loopback-streams", "1")])
def test_bench_multi_form_rehearsal_peer_transport_and_enqueue_threads(transport, threads):
    """`bench.py --gpus 4` on THIS box with the transports that give every shard its own stream (CMF_BENCH_TRANSPORT): the
    supervisor starts the measurement in a child process, the first form of the ladder delivers, `comm` says which transport
    carried the collectives and who enqueued, and the probe of the overlap form (default from 4 GPUs on) records both times."""
    env = dict(os.environ, CMF_BENCH_DEVICES="0,0,0,0", CMF_BENCH_TRANSPORT=transport, CMF_ENQUEUE_THREADS=threads)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1", "--sustain", "0",
                        "--T", "8000"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["value"] > 0 and line["n_gpus"] == 4 and [a["ok"] for a in line["attempts"]] == [True]
    assert line["comm"]["transport"] == transport and line["comm"]["enqueue"] == ("threads" if threads == "1" else "caller")
    probe = line["allreduce_overlap_probe_ms"]
    assert probe["single"] > 0 and probe["overlap"] > 0 and line["allreduce_overlap"] == (probe["overlap"] < probe["single"])
    assert line["group_extras"] is None  # opt-in since round 4
    c = line["comm"]["collectives"]
    assert c["allgather_halo"]["avg_ms"] > 0 and c["allreduce_gram_payload"]["bytes"] < line["comm"]["allreduce"]["bytes"]


@pytest.mark.gpu
def test_bench_failure_after_group_creation_still_prints_the_line():
    """First-contact diagnostics: the measurement itself (`--child multi`, what the supervisor starts) fails in its first
    all-reduce -- provoked with the library's test hook on a 3-shard peer group -- and still prints ONE JSON line with
    value null, the failing phase, cmf_last_error and `comm`, and exits non-zero (what the supervisor does with such a child
    is covered on the CPU: tests/test_bench_supervisor.py)."""
    env = dict(os.environ, CMF_BENCH_DEVICES="0,0,0", CMF_BENCH_TRANSPORT="peer", CMF_TEST_HOOKS="1", CMF_TEST_FAIL_SHARD="1", CMF_ALLREDUCE_OVERLAP="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    args = ["--gpus", "3", "--steps", "3", "--warmup", "1", "--sustain", "0", "--T", "6000"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--child", "multi"] + args, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 3, p.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["value"] is None and line["failed_phase"] == "warm-up steps" and "CMF_TEST_FAIL_SHARD" in line["cmf_last_error"]
    assert line["comm"]["transport"] == "peer" and line["comm"]["nranks"] == 3 and line["n_gpus"] == 3
    assert "failed in phase 'warm-up steps'" in p.stderr.decode()
