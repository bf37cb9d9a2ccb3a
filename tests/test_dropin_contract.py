"""The rule READS the W and H it is handed (src/algs/mult.jl:23,42): the drop-in contract of INTEGRATION.md section 3b.

The reference's rules take W and H as arguments of every call and mutate them in place; the GPU rule's working copies are
device-resident.  Under sync_every_call the binding keeps the two equivalent: it fingerprints its arguments (cmf_fingerprint)
and uploads arrays it has not seen in that state.  Every case drives the oracle's rule (oracle.update_motifs /
update_feature_maps, line for line mult.jl:23-58) and the GPU rule with the SAME host-side edits and compares at the north
star's 1e-4.  The CPU-only cases pin the fingerprint itself.
"""
import ctypes

import numpy as np
import pytest


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def fingerprint(lib, a, stride):
    fp = ctypes.c_uint64()
    assert lib.cmf_fingerprint(a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), a.size, stride, ctypes.byref(fp)) == 0
    return fp.value


def test_fingerprint_is_a_function_of_the_contents():
    """Host arithmetic, no device: equal arrays give equal values wherever they live, the full form sees every element, the sampled
    form every bulk edit and always the last line and the length."""
    import cmf_jl_amd as m

    lib = m.load_library()
    rng = np.random.default_rng(0)
    a = rng.random(100003)
    b = a.copy()
    for stride in (1, 64):
        assert fingerprint(lib, a, stride) == fingerprint(lib, b, stride)
        assert fingerprint(lib, a, stride) != fingerprint(lib, a[:-1].copy(), stride)  # the length counts
        c = a.copy()
        c[-1] += 1e-9  # the last line always counts
        assert fingerprint(lib, a, stride) != fingerprint(lib, c, stride)
        assert fingerprint(lib, a, stride) != fingerprint(lib, 2.0 * a, stride)  # a bulk edit
    c = a.copy()
    c[12345] = np.nextafter(c[12345], 2.0)  # one ulp of one element: the full form's business
    assert fingerprint(lib, a, 1) != fingerprint(lib, c, 1)
    z = np.zeros(0)
    assert fingerprint(lib, z, 1) == fingerprint(lib, z, 64)
    big = rng.random(1 << 20)  # (the threaded path of the full form: same value as a copy)
    assert fingerprint(lib, big, 1) == fingerprint(lib, big.copy(), 1)
    assert lib.cmf_fingerprint(None, 5, 1, ctypes.byref(ctypes.c_uint64())) != 0


@pytest.fixture(scope="module")
def cmf():
    import cmf_jl_amd as m

    assert m.load_library().cmf_device_count() >= 1, "no HIP device: the gpu tests need a real MI355X"
    return m


def problem(oracle, N=70, T=400, K=32, L=8, seed=5):
    data, _, _ = oracle.c_gen_synthetic(N=N, T=T, K=3, L=min(L, 8), seed=seed)
    W0, H0 = oracle.c_init_rand(data, L=L, K=K, seed=seed + 1)
    return data, W0, H0


def make(cmf, oracle, data, W0, H0, **kw):
    rule = cmf.MultUpdate(data, W0, H0, sync_every_call=True, **kw)  # (CMFHip.jl's constructor keywords)
    W, H = np.array(W0, order="F", copy=True), np.array(H0, order="F", copy=True)
    Wo, Ho = W0.copy(), H0.copy()
    return rule, W, H, oracle.MultUpdate(data, Wo, Ho), Wo, Ho


@pytest.mark.gpu
@pytest.mark.parametrize("K", [32, 5])
def test_h_edited_between_iterations_is_read(cmf, oracle, K):
    data, W0, H0 = problem(oracle, K=K)
    rule, W, H, orule, Wo, Ho = make(cmf, oracle, data, W0, H0)
    try:
        for it in range(3):
            rule.update_motifs(data, W, H)
            oracle.update_motifs(orule, data, Wo, Ho)
            got = rule.update_feature_maps(data, W, H)
            want = oracle.update_feature_maps(orule, data, Wo, Ho)
            assert abs(got - want) <= 1e-4 * want and rel(W, Wo) < 1e-4 and rel(H, Ho) < 1e-4
            H *= 0.5  # the caller rescales H in place between iterations, like a normalisation step would
            Ho *= 0.5
        assert rule.reuploads == 2  # the edits after iterations 0 and 1 were seen (the last is never read)
    finally:
        rule.close()


@pytest.mark.gpu
def test_other_arrays_in_the_second_iteration_are_read(cmf, oracle):
    """One rule object handed different arrays (a second fit from new initial factors): the reference computes from what it is
    handed."""
    data, W0, H0 = problem(oracle)
    rule, W, H, orule, Wo, Ho = make(cmf, oracle, data, W0, H0)
    try:
        rule.update_motifs(data, W, H)
        rule.update_feature_maps(data, W, H)
        W2, H2 = oracle.c_init_rand(data, L=W0.shape[2], K=W0.shape[0], seed=99)
        W2c, H2c = np.array(W2, order="F", copy=True), np.array(H2, order="F", copy=True)
        rule.update_motifs(data, W2c, H2c)
        got = rule.update_feature_maps(data, W2c, H2c)
        oracle.update_motifs(orule, data, W2, H2)
        want = oracle.update_feature_maps(orule, data, W2, H2)
        assert abs(got - want) <= 1e-4 * want and rel(W2c, W2) < 1e-4 and rel(H2c, H2) < 1e-4
        assert rule.reuploads == 1
    finally:
        rule.close()


@pytest.mark.gpu
def test_w_edited_between_the_two_calls(cmf, oracle):
    """The verdict's case.  strict_inplace: update_motifs writes W back, the edit is made to the current W and is honoured -- the
    oracle's answer.  Default: the caller's W is the one update_motifs started from, so the edit is refused with a clear error."""
    data, W0, H0 = problem(oracle)
    rule, W, H, orule, Wo, Ho = make(cmf, oracle, data, W0, H0, strict_inplace=True)
    try:
        for it in range(2):
            rule.update_motifs(data, W, H)
            oracle.update_motifs(orule, data, Wo, Ho)
            assert rel(W, Wo) < 1e-4  # the caller sees the new motifs between the calls, like the reference's caller
            W[:, :, 0] *= 1.5  # ... and edits them
            Wo[:, :, 0] *= 1.5
            got = rule.update_feature_maps(data, W, H)
            want = oracle.update_feature_maps(orule, data, Wo, Ho)
            assert abs(got - want) <= 1e-4 * want and rel(W, Wo) < 1e-4 and rel(H, Ho) < 1e-4
        assert rule.reuploads == 2
    finally:
        rule.close()
    rule, W, H, orule, Wo, Ho = make(cmf, oracle, data, W0, H0)
    try:
        rule.update_motifs(data, W, H)
        W[:, :, 0] *= 1.5
        with pytest.raises(RuntimeError, match="strict_inplace"):
            rule.update_feature_maps(data, W, H)
        rule.upload(W, H)  # the documented way out
        rule.update_feature_maps(data, W, H)
    finally:
        rule.close()


@pytest.mark.gpu
def test_full_verification_sees_one_element_and_none_sees_nothing(cmf, oracle):
    data, W0, H0 = problem(oracle)
    for mode, seen in (("full", 1), ("none", 0)):
        rule, W, H, orule, Wo, Ho = make(cmf, oracle, data, W0, H0, verify_args=mode)
        try:
            rule.update_motifs(data, W, H)
            rule.update_feature_maps(data, W, H)
            H[3, 77] += 0.25  # not on a sampled line start, one element
            rule.update_motifs(data, W, H)
            assert rule.reuploads == seen
        finally:
            rule.close()


@pytest.mark.gpu
@pytest.mark.parametrize("alg", ["hals", "pgd"])
def test_the_other_rules_read_their_arguments_too(cmf, oracle, alg):
    data, W0, H0 = problem(oracle, K=6)
    Rule = cmf.HALSUpdate if alg == "hals" else cmf.PGDUpdate
    rule = Rule(data, W0, H0)
    ref = Rule(data, W0, H0)
    rule.sync_every_call = True
    W, H = np.array(W0, order="F", copy=True), np.array(H0, order="F", copy=True)
    try:
        rule.update_motifs(data, W, H)
        rule.update_feature_maps(data, W, H)
        ref.update_motifs()
        ref.update_feature_maps()
        H *= 0.5
        Wr, Hr = ref.download()
        ref.upload(Wr, 0.5 * Hr)  # the explicit way of saying the same thing
        rule.update_motifs(data, W, H)
        got = rule.update_feature_maps(data, W, H)
        ref.update_motifs()
        want = ref.update_feature_maps()
        assert got == want and rule.reuploads == 1
        Wr, Hr = ref.download()
        assert np.array_equal(W, Wr) and np.array_equal(H, Hr)
    finally:
        rule.close()
        ref.close()
