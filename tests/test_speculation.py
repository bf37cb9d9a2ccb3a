"""The C2 contraction of the next update_motifs! enqueued behind the loss conv of update_feature_maps! (option "speculate",
include/cmf_hip.h): identical results whatever the caller does between the two calls."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cmf():
    import cmf_jl_amd as m

    return m


def problem(cmf, N, T, K, L, seed=3):
    data = cmf.gen_synthetic(N=N, T=T, seed=seed)
    W0, H0 = cmf.init_rand(data, L=L, K=K, seed=1)
    return data, W0, H0


def run(cmf, data, W0, H0, script, speculate):
    """script: a list of steps; returns every loss and the factors after every step that produces them."""
    rule = cmf.MultUpdate(data, W0, H0)
    rule.set_option("speculate", speculate)
    out = []
    try:
        for step in script:
            if step == "W":
                rule.update_motifs(l1W=0.05, l2W=0.1)
            elif step == "H":
                out.append(rule.update_feature_maps(l1H=0.02, l2H=0.3))
            elif step == "W2":  # other regularisation than the call before: the contraction does not depend on it
                rule.update_motifs(l1W=0.5, l2W=0.0)
            elif step == "get":
                out.extend(rule.download())
            elif step == "set":
                W, H = rule.download()
                rule.upload(np.asfortranarray(W * 0.5 + 0.01), np.asfortranarray(H * 2.0))
            elif step == "loss":
                out.append(rule.compute_loss())
            elif step == "iterate":
                out.extend(rule.iterate(3))
            elif step == "gram":
                rule.set_option("gram", 1)
            elif step == "nogram":
                rule.set_option("gram", 0)
            elif step == "small0":
                rule.set_option("small_k", 0)
            elif step == "noreuse":
                rule.set_option("reuse_est", 0)
            elif step == "conv":
                rule.time_kernel("conv", reps=1)  # est = tensor_conv(W, H) written again
            else:
                raise AssertionError(step)
        out.extend(rule.download())
        hits = rule.counter("speculated_contractions")
    finally:
        rule.close()
    return out, hits


SCRIPTS = [
    (["W", "H"] * 6, 5),                                  # the reference's loop: every update_motifs! but the first finds its contraction done
    (["W", "H", "W2", "H", "W", "H"], 2),                 # l1W / l2W may change from call to call
    (["W", "H", "H", "W", "H", "W", "H"], 1),             # H twice in a row (a refit of H): the first speculation is stale, no second one (no alternation)
    (["W", "H", "get", "W", "H", "get", "W", "H"], 2),    # reading the factors in between changes nothing
    (["W", "H", "set", "W", "H", "W", "H"], 1),           # new factors: discarded
    (["W", "H", "loss", "W", "H", "conv", "W", "H"], 0),  # a loss / a conv of its own rewrites est (same values, new generation): discarded
    (["W", "H", "iterate", "W", "H", "W", "H"], 2),       # the pipelined loop consumes the speculation (first W phase), then call by call again
    (["W", "H", "gram", "W", "H", "nogram", "W", "H", "W", "H"], 1),
    (["W", "H", "noreuse", "W", "H", "W", "H"], 0),
]


@pytest.mark.parametrize("N,T,K,L", [(96, 1500, 32, 8), (70, 2600, 5, 12), (130, 900, 12, 10)])
@pytest.mark.parametrize("case", range(len(SCRIPTS)))
def test_speculated_contraction_changes_nothing(cmf, N, T, K, L, case):
    script, want_hits = SCRIPTS[case]
    data, W0, H0 = problem(cmf, N, T, K, L)
    ref, hits0 = run(cmf, data, W0, H0, script, 0)
    got, hits1 = run(cmf, data, W0, H0, script, 1)
    assert hits0 == 0
    assert hits1 == want_hits, (script, hits1)
    assert len(ref) == len(got)
    for a, b in zip(ref, got):
        assert np.array_equal(np.asarray(a), np.asarray(b))


def test_small_k_off_in_between(cmf):
    """Switching the kernel family between the two calls: the slabs the speculation filled belong to the other family."""
    data, W0, H0 = problem(cmf, 70, 2600, 5, 12)
    script = ["W", "H", "small0", "W", "H", "W", "H"]
    ref, _ = run(cmf, data, W0, H0, script, 0)
    got, hits = run(cmf, data, W0, H0, script, 1)
    assert hits == 1
    for a, b in zip(ref, got):
        assert np.array_equal(np.asarray(a), np.asarray(b))


def test_what_the_other_rules_speculate(cmf):
    data, W0, H0 = problem(cmf, 64, 1200, 8, 6)
    for cls in (cmf.HALSUpdate, cmf.PGDUpdate):
        rule = cls(data, W0, H0)
        for _ in range(3):
            rule.update_motifs()
            rule.update_feature_maps()
        # (round 6: the HALS rule speculates its own W-phase contraction -- test_hals_speculates_the_next_w_phase_too; PGD's depends on
        # the step it has just accepted or rejected: nothing to send ahead)
        assert rule.counter("speculated_contractions") == (2 if cls is cmf.HALSUpdate else 0)
        rule.close()


def test_stream_switch_between_the_two_calls(cmf):
    """cmf_set_stream while a speculated contraction may still be in flight on the old stream: the switch waits for it and nothing
    speculated carries over to the new stream (include/cmf_hip.h: cmf_set_stream)."""
    import ctypes

    from cmf_jl_amd._lib import check

    hip = ctypes.CDLL("libamdhip64.so")  # the runtime the library itself is linked to (a second one -- torch's -- would not see the device)
    data, W0, H0 = problem(cmf, 200, 6000, 32, 10)
    ref = cmf.MultUpdate(data, W0, H0)
    rule = cmf.MultUpdate(data, W0, H0)
    streams = []
    for _ in range(2):
        st = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0  # hipStreamNonBlocking
        streams.append(st)
    try:
        for it in range(4):
            ref.update_motifs()
            want = ref.update_feature_maps()
            rule.update_motifs()
            got = rule.update_feature_maps()  # returns with the next contraction enqueued behind the loss conv
            assert got == want
            check(rule._lib.cmf_set_stream(rule._h, streams[it % 2]))
        Wa, Ha = ref.download()
        Wb, Hb = rule.download()
        assert np.array_equal(Wa, Wb) and np.array_equal(Ha, Hb)
        assert rule.counter("speculated_contractions") == 0 and ref.counter("speculated_contractions") == 3
    finally:
        ref.close()
        rule.close()
        for st in streams:
            hip.hipStreamDestroy(st)


@pytest.mark.parametrize("N,T,K,L", [(60, 700, 5, 10), (130, 9000, 32, 20)])
def test_hals_speculates_the_next_w_phase_too(cmf, N, T, K, L):
    """The HALS rule (round 6): G = resid * H_unfold' and the lag correlations of the next update_motifs! (hals.jl:56-60, 104-110) go
    out behind the loss reduction of update_feature_maps! when the caller alternates the two calls -- identical results whatever happens
    in between, and the work is taken only when nothing has touched H, W or the residual."""
    data, W0, H0 = problem(cmf, N, T, K, L)
    script = ["W", "H", "W", "H", "get", "W", "H", "set", "W", "H", "H", "W", "loss", "W2", "H", "W", "H"]
    out = {}
    for spec in (1, 0):
        rule = cmf.HALSUpdate(data, W0, H0)
        rule.set_option("speculate", spec)
        res = []
        try:
            for step in script:
                if step == "W":
                    rule.update_motifs(l1W=0.05, l2W=0.1)
                elif step == "W2":
                    rule.update_motifs(l1W=0.5, l2W=0.0)
                elif step == "H":
                    res.append(rule.update_feature_maps(l1H=0.02, l2H=0.3))
                elif step == "get":
                    res.extend(rule.download())
                elif step == "set":
                    W, H = rule.download()
                    rule.upload(np.asfortranarray(W * 0.5 + 0.01), np.asfortranarray(H * 2.0))
                elif step == "loss":
                    res.append(rule.compute_loss())
            res.extend(rule.download())
            out[spec] = (res, rule.counter("speculated_contractions"))
        finally:
            rule.close()
    for a, b in zip(out[1][0], out[0][0]):
        assert np.array_equal(np.asarray(a), np.asarray(b))
    # taken: after W-H (x2: "get" reads only), after the H that follows "set"+W ... not after "set" (new factors), not after H-H
    assert out[0][1] == 0 and out[1][1] >= 3
