"""Generates the committed fixtures in tests/golden/.

The reference is Julia and cannot run here, and its own tests hold no vectors for
the MU path, so these fixtures are produced by the oracle
(oracle/cmf_oracle.py + oracle/cmf_oracle.c), whose two restatements must agree
before anything is written.  They pin the oracle against regressions and give the
GPU parity tests fixed inputs/outputs that travel to the GPU box.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import cmf_oracle as o  # noqa: E402


def fixture(name, N, T, Ktrue, Ltrue, K, L, max_itr, reg, data_seed=1234, init_seed=0):
    data, _, _ = o.c_gen_synthetic(N=N, T=T, K=Ktrue, L=Ltrue, seed=data_seed)
    W0, H0 = o.c_init_rand(data, L=L, K=K, seed=init_seed)
    Wa, Ha, la, _ = o.fit_mult(data, W0, H0, max_itr=max_itr, check_convergence=False, **reg)
    Wb, Hb, lb, _ = o.c_fit_mult(data, W0, H0, max_itr=max_itr, check_convergence=False, **reg)
    np.testing.assert_allclose(la, lb, rtol=1e-10)
    np.testing.assert_allclose(Wa, Wb, rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(Ha, Hb, rtol=1e-8, atol=1e-12)
    r = dict(l1W=0.0, l2W=0.0, l1H=0.0, l2H=0.0)
    r.update(reg)
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"),
        data=data.astype(np.float64), W0=W0, H0=H0, W=Wa, H=Ha, loss_hist=la, max_itr=max_itr,
        conv0=o.tensor_conv(W0, H0), transconv0=o.tensor_transconv(W0, data), **r)
    print(name, data.shape, "loss", la[0], "->", la[-1])


def fixture_rule(name, rule, N, T, Ktrue, Ltrue, K, L, max_itr, data_seed=1234, init_seed=0):
    """HALS (hals.jl:31-42, 90-154) and PGD (pgd.jl:158-255, default loss / penalties / constraints) fixtures: the C and the
    numpy restatement must agree before anything is written."""
    data, _, _ = o.c_gen_synthetic(N=N, T=T, K=Ktrue, L=Ltrue, seed=data_seed)
    W0, H0 = o.c_init_rand(data, L=L, K=K, seed=init_seed)
    if rule == "hals":
        Wa, Ha, la, _ = o.c_fit_hals(data, W0, H0, max_itr=max_itr, check_convergence=False)
        Wb, Hb, lb, _ = o.fit_hals(data, W0, H0, max_itr=max_itr, check_convergence=False)
    else:
        Wa, Ha, la, _ = o.fit_pgd(data, W0, H0, max_itr=max_itr)
        Wb, Hb, lb, _ = o.c_fit_pgd(data, W0, H0, max_itr=max_itr)
    np.testing.assert_allclose(la, lb, rtol=1e-9)
    np.testing.assert_allclose(Wa, Wb, rtol=1e-7, atol=1e-11)
    np.testing.assert_allclose(Ha, Hb, rtol=1e-7, atol=1e-11)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), data=data.astype(np.float64), W0=W0, H0=H0, W=Wa, H=Ha, loss_hist=la,
                        max_itr=max_itr, l1W=0.0, l2W=0.0, l1H=0.0, l2H=0.0, rule=rule)
    print(name, rule, data.shape, "loss", la[0], "->", la[-1])


def main():
    lib = o.c_lib()
    np.savez(
        os.path.join(HERE, "rng_kat.npz"),
        u01_seed1234_stream0=np.array([lib.oracle_rng_u01(1234, 0, i) for i in range(16)]),
        normal_seed1234_stream16=np.array([lib.oracle_rng_normal(1234, 16, i) for i in range(16)]),
    )
    fixture("mu_small", N=48, T=300, Ktrue=3, Ltrue=10, K=4, L=8, max_itr=20, reg={})
    fixture("mu_small_reg", N=48, T=300, Ktrue=3, Ltrue=10, K=4, L=8, max_itr=20,
            reg=dict(l1W=0.1, l2W=0.5, l1H=0.1, l2H=0.2))  # README.md:52 values
    fixture("mu_k5", N=70, T=257, Ktrue=3, Ltrue=20, K=5, L=10, max_itr=10, reg={})  # config-1 K,L at reduced N,T
    fixture_rule("hals_small", "hals", N=48, T=300, Ktrue=3, Ltrue=10, K=4, L=8, max_itr=8)
    fixture_rule("pgd_small", "pgd", N=48, T=300, Ktrue=3, Ltrue=10, K=4, L=8, max_itr=10)


if __name__ == "__main__":
    main()
