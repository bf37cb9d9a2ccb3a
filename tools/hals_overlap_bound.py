"""Bound experiment for overlapping the HALS H row pipeline with the residual conv behind it.

The persistent row pipeline (hals_h_persist_kernel: K + (K-1)P = 156 workgroups at config 5, VALU only) leaves 100 of the
256 CUs and every MFMA pipe idle for 1.44 ms; the residual / loss conv behind it needs, per tile row, only that the LAST
row's sweeper has passed.  Before building the flag protocol: how much sooner does the pair finish when the first pct % of
the conv's tile rows run on a CU-masked second stream beside the pipeline with NO dependency at all (an upper bound: a real
chaser also waits for the last row, which starts 0.35 ms into the pipeline)?  cmf_time_kernel("hals_overlap:<pct>:<mode>").

    python tools/hals_overlap_bound.py [reps=20]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmf_jl_amd as cmf  # noqa: E402

N, T, K, L = 2000, 50000, 32, 20
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
data = cmf.gen_synthetic(N=N, T=T, seed=1234)
W0, H0 = cmf.init_rand(data, L=L, K=K, seed=0)


def fresh():
    rule = cmf.HALSUpdate(data, W0, H0)
    for _ in range(2):  # the H phase's state (P, taps, flags) exists
        rule.update_motifs()
        rule.update_feature_maps()
    return rule


rule = fresh()
rule.set_option("profile", 1)
for _ in range(4):
    rule.update_motifs()
    rule.update_feature_maps()
pipe, _ = rule.kernel_times("hals_h_pipeline")
conv, _ = rule.kernel_times("conv_resid")
rule.set_option("profile", 0)
print(f"in the iteration: pipeline {pipe:.4f} ms, residual conv {conv:.4f} ms, back to back {pipe + conv:.4f} ms", flush=True)
seq, _ = rule.time_kernel("hals_overlap:0:0", reps)
print(f"back to back, timed as a pair: {seq:.4f} ms", flush=True)
rule.close()
for mode in (0, 1):
    for pct in (20, 30, 40, 50, 60):
        rule = fresh()
        ms, _ = rule.time_kernel(f"hals_overlap:{pct}:{mode}", reps)
        reruns = rule.get_counter("hals_pipeline_reruns") if hasattr(rule, "get_counter") else -1
        print(f"mode {mode} ({'both streams masked 160 / 96 CUs' if mode == 0 else 'only the conv part masked (96 CUs)'}), "
              f"first {pct} % of the conv beside the pipeline: {ms:.4f} ms per pair ({seq - ms:+.4f} ms against back to back)", flush=True)
        rule.close()
