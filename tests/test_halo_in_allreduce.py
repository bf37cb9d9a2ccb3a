"""ONE collective per iteration on a T-sharded group: the halo of H in the tail of the W-phase all-reduce (csrc/cmf_groups.hip).

north_star: "sharding the T axis of data and H, with a single RCCL all-reduce over xGMI on the W-update numerator/denominator per
iteration".  Until round 5 the (L-1)-column halos of the new H travelled in an all-gather of their own between the H update and the
loss conv.  Now every shard with a left neighbour updates the L-1 columns in front of its own itself (mult.jl:44-52 on those columns,
from an H that is valid 2(L-1) columns out), so the loss conv (mult.jl:55-57) and the next W phase (mult.jl:28-34) need no exchange,
and every rank's outer columns of the new H ride in per-rank slots behind the loss tail of the NEXT all-reduce.  What these tests pin,
on loopback / stream-per-shard / peer groups on one GPU: the oracle's results at the north star's 1e-4, agreement with the
all-gather form to rounding, the collectives actually issued (cmf_get_counter), and every way out of the steady state (losses and
factors read in between, factors set, the form switched, eval_mode, the Gram form, shapes the form does not take).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cmf():
    import cmf_jl_amd as m

    assert m.load_library().cmf_device_count() >= 1, "no HIP device: the gpu tests need a real MI355X"
    return m


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b)


def problem(oracle, N, T, K, L, seed=11):
    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=min(L, 12), seed=seed)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=seed + 1)
    return data, W0, H0


@pytest.mark.parametrize("R,transport,N,T,K,L,threads", [(2, 2, 70, 900, 32, 8, 1), (3, 3, 40, 1000, 32, 20, 1), (8, 2, 33, 2100, 64, 7, 1), (4, 4, 50, 1300, 32, 33, 1),
                                                         (5, 3, 24, 700, 32, 2, 1), (3, 3, 40, 1000, 32, 20, 0), (4, 4, 50, 1300, 32, 12, 0)])
def test_one_collective_per_iteration_and_the_oracles_results(cmf, oracle, R, transport, N, T, K, L, threads):
    data, W0, H0 = problem(oracle, N, T, K, L)
    reg = dict(l1W=0.05, l2W=0.1, l1H=0.05, l2H=0.1)
    n = 6
    Wr, Hr, lr, _ = oracle.fit_mult(data, W0, H0, max_itr=n, check_convergence=False, **reg)
    out = {}
    for form in (1, 0):
        rule = cmf.MultUpdate(data, W0, H0, devices=[0] * R, transport=transport)
        try:
            rule.set_option("halo_in_allreduce", form)
            assert rule.counter("halo_in_allreduce") == form
            if transport != 2:
                rule.set_option("enqueue_threads", threads)  # an enqueue worker per shard, or the calling thread for all of them
            rule.iterate(1, **reg)  # (the first iteration starts from the set-up exchange)
            ar0, ag0 = rule.counter("allreduce_calls"), rule.counter("allgather_calls")
            ls = list(rule.iterate(n - 2, **reg))
            ar1, ag1 = rule.counter("allreduce_calls"), rule.counter("allgather_calls")
            # steady state: one all-reduce per iteration; the all-gathers are the halo exchange of the old form (one per iteration) and
            # the flush of the batch's last loss (one per cmf_iterate call)
            assert ar1 - ar0 == n - 2
            assert ag1 - ag0 == (1 if form else (n - 2) + 1)
            # call by call (what the reference's fit does): update_motifs!, then update_feature_maps! with its synchronous loss
            rule.update_motifs(**reg)
            ls.append(rule.update_feature_maps(**reg))
            out[form] = (np.array(ls),) + rule.download()
        finally:
            rule.close()
    for form in (1, 0):
        np.testing.assert_allclose(out[form][0], lr[2:], rtol=1e-4)
        assert rel(out[form][1], Wr) < 1e-4 and rel(out[form][2], Hr) < 1e-4
    # the two forms differ only in WHO computed the L-1 columns in front of a shard between an H update and the next all-reduce
    np.testing.assert_allclose(out[1][0], out[0][0], rtol=2e-6)
    assert rel(out[1][1], out[0][1]) < 2e-6 and rel(out[1][2], out[0][2]) < 2e-6


def test_every_way_out_of_the_steady_state(cmf, oracle):
    """Between an H update and the next all-reduce only the L-1 columns in front of a shard are valid: whatever else may come next --
    the loss, the factors, new factors, another H update (eval_mode), the other form, the Gram form -- must find (or make) the halos
    it needs.  The same sequence of calls on a group in either form and on ONE handle."""
    data, W0, H0 = problem(oracle, 60, 1500, 32, 12)

    def drive(rule):
        seq = []
        rule.update_motifs()
        seq.append(rule.update_feature_maps())
        seq.append(rule.compute_loss())                   # a loss between the phases (reads the columns in front)
        rule.update_motifs()
        seq.append(rule.update_feature_maps())
        W, H = rule.download()                            # factors read in the pending state
        seq.append(rule.update_feature_maps())            # two H updates in a row (what eval_mode does): needs whole halos again
        seq += list(rule.iterate(2, eval_mode=True))
        seq += list(rule.iterate(2))
        rule.upload(W, 0.5 * H)                           # new factors
        seq += list(rule.iterate(2))
        rule.set_option("gram", 1)                        # the Gram form keeps the all-gather
        seq += list(rule.iterate(2))
        rule.set_option("gram", 0)
        seq += list(rule.iterate(2))
        return np.array(seq), rule.download()

    one = cmf.MultUpdate(data, W0, H0)
    want, (Ww, Hw) = drive(one)
    one.close()
    for form, switch in ((1, False), (0, False), (1, True)):
        rule = cmf.MultUpdate(data, W0, H0, devices=[0, 0, 0, 0], transport=3)
        try:
            rule.set_option("halo_in_allreduce", form)
            if switch:  # the form switched in mid-run, in the pending state
                rule.update_motifs()
                rule.update_feature_maps()
                rule.set_option("halo_in_allreduce", 0)
                rule.upload(W0, H0)
                rule.set_option("halo_in_allreduce", 1)
            got, (Wg, Hg) = drive(rule)
        finally:
            rule.close()
        np.testing.assert_allclose(got, want, rtol=2e-5, err_msg=f"form {form} switch {switch}")
        assert rel(Wg, Ww) < 2e-5 and rel(Hg, Hw) < 2e-5


@pytest.mark.parametrize("R,N,T,K,L,why", [(3, 40, 600, 33, 8, "K is not a multiple of 32"), (3, 40, 600, 5, 8, "few components"),
                                           (4, 30, 40, 32, 8, "shards shorter than 2 (L-1)"), (2, 30, 400, 32, 1, "L = 1: no halo at all")])
def test_shapes_the_form_does_not_take_keep_the_all_gather(cmf, oracle, R, N, T, K, L, why):
    data, W0, H0 = problem(oracle, N, T, K, L)
    Wr, Hr, lr, _ = oracle.fit_mult(data, W0, H0, max_itr=3, check_convergence=False)
    rule = cmf.MultUpdate(data, W0, H0, devices=[0] * R)
    try:
        assert rule.counter("halo_in_allreduce") == 0, why
        ls = rule.iterate(3)
        W, H = rule.download()
    finally:
        rule.close()
    np.testing.assert_allclose(ls, lr[1:], rtol=1e-4)
    assert rel(W, Wr) < 1e-4 and rel(H, Hr) < 1e-4
