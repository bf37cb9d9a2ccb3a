"""Two CU-masked half-chips against one whole chip at the T/8 shard size.

A T/8 shard iteration is four contraction launches that are each ONE synchronised round of the chip: every wave's prologue
coincides at the start and every epilogue at the end (~100 us of 735).  The idea under test: cut the shard's columns in two
halves, run each half on its own stream restricted to half of the CUs (hipExtStreamCreateWithCUMask) with the statically
dealt kernels planned for that half, and let the two streams drift against each other so that one half's ramp / drain falls
into the other's steady state.  This script measures the UPPER BOUND of that design: the two halves as two INDEPENDENT
problems of T/16 columns (no halo, no all-reduce, no meeting point between them -- the real thing would have two per
iteration), each driven by its own host thread through cmf_iterate, against one handle with all T/8 columns on the whole chip.

    CMF_TEST_HOOKS=1 python tools/cu_mask_experiment.py [T_shard=6250] [steps=400]
"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CMF_TEST_HOOKS"] = "1"
import cmf_jl_amd as cmf  # noqa: E402

N, K, L = 2000, 32, 20
Ts = int(sys.argv[1]) if len(sys.argv) > 1 else 6250
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400


def make(T, mask=None):
    if mask:
        os.environ["CMF_EXP_CU_MASK"] = mask
    else:
        os.environ.pop("CMF_EXP_CU_MASK", None)
    d = cmf.gen_synthetic(N=N, T=T, seed=1234)
    W, H = cmf.init_rand(d, L=L, K=K, seed=0)
    r = cmf.MultUpdate(d, W, H)
    os.environ.pop("CMF_EXP_CU_MASK", None)
    return r


def run(rules, stagger_s=0.0):
    for r in rules:
        r.iterate(5)
        r.synchronize()
    th = []
    t0 = time.perf_counter()
    for i, r in enumerate(rules):
        def work(r=r, i=i):
            if i and stagger_s:
                time.sleep(stagger_s)
            r.iterate(steps)
            r.synchronize()
        th.append(threading.Thread(target=work))
        th[-1].start()
    for t in th:
        t.join()
    return 1e3 * (time.perf_counter() - t0 - (stagger_s if len(rules) > 1 else 0.0)) / steps


whole = make(Ts)
print(f"one handle, T = {Ts}, whole chip: {run([whole]):.4f} ms per iteration", flush=True)
whole.close()
half_plain = [make(Ts // 2), make(Ts // 2)]
print(f"two handles of T = {Ts // 2} on two UNMASKED streams (both planned for 256 CUs): {run(half_plain):.4f} ms per iteration pair", flush=True)
for r in half_plain:
    r.close()
for mode in ("x", "c"):
    halves = [make(Ts // 2, f"0/2/{mode}"), make(Ts // 2, f"1/2/{mode}")]
    one = run(halves[:1])
    both = run(halves)
    print(f"two handles of T = {Ts // 2} on CU-masked streams (128 CUs each, cut {'by whole XCDs' if mode == 'x' else 'inside every XCD'}): "
          f"one alone {one:.4f} ms, both at once {both:.4f} ms per iteration pair", flush=True)
    for r in halves:
        r.close()
