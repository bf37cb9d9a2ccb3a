#!/usr/bin/env python3
"""bench.py -- MU iterations/sec of the MI355X hot path on BASELINE.json's config 2
(N=2000, T=50000, K=32, L=20, fp32, alg=:mult).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N ...                      # launched plainly: ONE process drives the N GPUs (cmf_create_multi)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W      # one process per GPU (cmf_create_shard + RCCL)

A "step" is one full MU iteration (update_motifs! + update_feature_maps!, alternating.jl:51-54, including the
per-iteration loss scalar read-back; for N > 1 also the RCCL all-reduce of [numW|denomW|loss tail] and the H halo
all-gather, both issued by libcmf_hip.so itself).  The K timed steps are ONE cmf_iterate call: the same kernels and
collectives as K x (cmf_update_motifs; cmf_update_feature_maps), with each loss read one iteration late from pinned
memory so that the host never stalls the device between iterations (every loss is still read by the host inside
the timed region).  Inputs (data, W, H) are resident in HBM when the timed region starts.  For N > 1 the T axis of
the SAME problem is sharded over the GPUs ("scaling": "strong").  Rank 0 prints one JSON line.

Which multi-GPU form runs is decided by `route()`: WORLD_SIZE == --gpus > 1 (a launcher started one process per GPU)
takes the per-process form; --gpus N > 1 with no launcher takes the one-process form -- no relaunch, no exec.  `comm` in
the JSON line says which transport carried the collectives, the RCCL version and how many ranks it saw.
Config 5 (HALS) does not shard over T (its H sweep is sequential along T): with --gpus N it runs N independent
replicas ("scaling": "weak").

First contact with a multi-GPU node (no such node was available to the build): for --gpus N > 1 the process that is
started does NOT touch the GPU.  It supervises: the measurement runs in a child process (a fresh process per attempt, so
a wedged collective or a crash inside RCCL costs an attempt, not the run), bounded by --attempt-timeout, down a ladder of
forms -- `attempts` in the JSON line says which ones were tried and why they ended:
    plain launch      one process, an enqueue thread per GPU  ->  one process, calling thread enqueues (grouped RCCL calls)
                      ->  one process per GPU started by bench.py itself (the launcher form below)
    launcher form     library's own RCCL communicator  ->  collectives through torch.distributed's RCCL backend on staged
                      buffers  ->  through gloo on host buffers
Under a launcher the supervising ranks agree through the launcher's TCP store (port, outcome of every rank, early abort).
When every form has failed, rank 0 still prints ONE JSON line -- "value": null, the failing phase, cmf_last_error, `comm` and
whatever was measured before the failure -- and the exit code is non-zero.  CMF_BENCH_SUPERVISE=0 runs the measurement in
the started process as before.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PMC_PROFILE = "r06b_pmc_summary.json"  # rocprofv3 --pmc passes of this round's kernels (profiles/README.md)
PMC_FALLBACK = "r05_pmc_summary.json"
PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 matrix peak
CONFIGS = {
    1: dict(N=500, T=2000, K=5, L=10),
    2: dict(N=2000, T=50000, K=32, L=20),
    3: dict(N=2000, T=400000, K=32, L=20),
    4: dict(N=2000, T=50000, K=32, L=20, l1_H=0.1, l2_H=0.2, l1_W=0.1, l2_W=0.5),
    5: dict(N=2000, T=50000, K=32, L=20, alg="hals"),
}


T_PROCESS_START = time.time()

def route(gpus, world_size):
    """Which form of the run (gpus, WORLD_SIZE) asks for:
    "single" -- one GPU, one process; "multi" -- ONE process drives `gpus` GPUs (cmf_create_multi, RCCL ncclCommInitAll):
    what a plain `python bench.py --gpus N` gets; "ranks" -- a launcher (torch.distributed.run) started one process per
    GPU (cmf_create_shard + cmf_comm_init_rccl).  Anything else is a launcher / flag mismatch."""
    gpus, world_size = int(gpus), int(world_size or 1)
    if gpus < 1:
        raise ValueError(f"--gpus must be >= 1 (got {gpus})")
    if world_size == 1:
        return "single" if gpus == 1 else "multi"
    if world_size == gpus:
        return "ranks"
    raise ValueError(f"--gpus {gpus} but WORLD_SIZE={world_size}: start one process per GPU (torch.distributed.run "
                     f"--nproc-per-node {gpus}) or none at all (plain `python bench.py --gpus {gpus}`)")


def flops_per_iter(N, T, K, L):
    S = L * T - L * (L - 1) / 2
    return 14.0 * K * N * S  # 7 contractions x 2*K*N*S


def parse_comm(info, mode, fallback=None):
    """cmf_comm_info's text as a record: did the collectives go through RCCL, which one, and how many ranks did it see."""
    rec = {"mode": mode, "info": info}
    for tok in info.split():
        if "=" in tok:
            k, v = tok.split("=", 1)
            rec[k] = int(v) if v.lstrip("-").isdigit() else v
    if "ranks" in rec and isinstance(rec["ranks"], str):
        rec["devices"] = [int(x.split("@dev")[1]) for x in rec["ranks"].split(",") if "@dev" in x]
    if fallback:
        rec["fallback_from_rccl"] = fallback
    return rec


def cpu_baseline_hals(data, W0, H0, budget_s):
    """HALS (src/algs/hals.jl) has no BLAS structure; the oracle's C restatement is timed on the first
    columns of the workload (one full iteration costs minutes on a CPU) and scaled to the full T."""
    import numpy as np
    from oracle import cmf_oracle as oracle

    T = data.shape[1]
    Ts = min(T, 1000)
    d = np.asfortranarray(data[:, :Ts])
    H = np.asfortranarray(H0[:, :Ts])
    t1 = time.perf_counter()
    oracle.c_fit_hals(d, W0, H, max_itr=1, check_convergence=False)
    el = time.perf_counter() - t1
    return dict(value=(Ts / T) / el, unit="iter/s", cores=int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1)), kind="port",
                sample=f"1 HALS iteration (hals.jl update_motifs!+update_feature_maps!, oracle C restatement, fp64) on the "
                       f"first {Ts} of {T} columns: {el:.1f} s, scaled by {Ts}/{T} (both HALS sweeps are linear in T)")


def cpu_baseline(data, W0, H0, budget_s):
    """The oracle (numpy/OpenBLAS fp64, the reference's per-lag GEMM structure) timed on this
    host's cores on a bounded sample: whole iterations at the bench workload until ~budget_s."""
    import numpy as np
    from oracle import cmf_oracle as oracle

    cores = os.cpu_count() or 1
    try:
        from threadpoolctl import threadpool_info

        blas = [p for p in threadpool_info() if p.get("user_api") == "blas"]
        if blas:
            cores = int(blas[0].get("num_threads", cores))
    except Exception:
        pass
    W = np.array(W0, dtype=np.float64, order="F", copy=True)
    H = np.array(H0, dtype=np.float64, order="F", copy=True)
    t0 = time.perf_counter()
    rule = oracle.MultUpdate(data, W, H)
    n = 0
    t1 = time.perf_counter()
    while True:
        oracle.update_motifs(rule, data, W, H)
        oracle.update_feature_maps(rule, data, W, H)
        n += 1
        el = time.perf_counter() - t1
        if el + el / n > budget_s or n >= 3:
            break
    return dict(value=n / el, unit="iter/s", cores=cores, kind="port",
                sample=f"{n} full MU iteration(s) (update_motifs!+update_feature_maps!) on the bench workload, "
                       f"fp64 numpy/OpenBLAS per-lag dgemm as in the reference; {el:.1f} s "
                       f"(+{t1 - t0:.1f} s rule setup)")


def hals_roofline(T, K, L, spans, ms_per_step):
    """The dominant part of a HALS iteration is the K*T strictly ordered H entry updates (hals.jl:121-154), run as a
    software pipeline over the rows (one persistent launch, hals_h_persist_kernel: a sweeper wave per row, puller
    workgroups applying the cross-row terms, flags in memory between them; option "hals_persist" = 0 selects the older
    one-launch-per-stage form).  Its bound is dependency latency, not MFMA or HBM: entry (k, t) needs (k, t-1) and the
    push of (k-1, t+L-1), so the critical path is T + (K-1)(L-1) dependent steps.  A step of the sweep is 9 single-wave
    instructions whose chain is one FMA and one MAX; in isolation it issues in 43 cycles (tools/valu_latency.hip,
    profiles/).  achieved = critical-path steps per second over the measured pipeline span; peak = one step per 43
    cycles at 2.4 GHz."""
    pipe_ms, n_pipe = spans.get("hals_h_pipeline", (0.0, 0))
    wsw_ms, _ = spans.get("hals_w_sweep", (0.0, 0))
    STEP_CYCLES, CLK = 43.0, 2.4e9
    crit_steps = T + (K - 1) * (L - 1)
    peak_steps = CLK / STEP_CYCLES
    ach_steps = crit_steps / (pipe_ms * 1e-3) if pipe_ms else 0.0
    return {"bound": "dependency-latency",
            "kernel": "hals_h_persist_kernel row pipeline (K*T ordered entry updates of H, hals.jl:121-154)",
            "achieved": ach_steps, "peak": peak_steps, "unit": "critical-path steps/s", "frac": ach_steps / peak_steps,
            "traffic": None, "critical_path_steps": crit_steps, "step_cycles_model": STEP_CYCLES,
            "pipeline_span_ms": pipe_ms, "pipeline_spans_timed": n_pipe, "pipeline_floor_ms": 1e3 * crit_steps / peak_steps,
            "share_of_step": pipe_ms / ms_per_step if ms_per_step else None,
            "w_sweep_ms": wsw_ms,
            "timing": "HIP event pair around the whole pipeline (option profile), over a bracketed pass of the steps behind the timed ones"}


def hals_steps(rule, n, reg_kw):
    out_ = []
    for _ in range(n):  # HALS: the rule's two calls per step
        rule.update_motifs(l1W=reg_kw["l1W"], l2W=reg_kw["l2W"])
        out_.append(rule.update_feature_maps(l1H=reg_kw["l1H"], l2H=reg_kw["l2H"]))
    return out_


def other_configs(cmf, rule, data, W0, H0, N, T, K, L, with_config3, device):
    """BASELINE.json's other configurations in the same run, same inputs and seeds (short: they are reported next to the
    headline, never part of `value`): config 4 (regularised MU) on the resident rule, config 5 (HALS) on a second rule
    over the same data with its latency roofline block, config 3's whole problem (T=400000) on this one GPU."""
    zero = dict(l1W=0.0, l2W=0.0, l1H=0.0, l2H=0.0)
    res = {}

    def mu_time(r, nsteps, kw):
        r.iterate(1, **kw)
        t0 = time.perf_counter()
        ls = r.iterate(nsteps, **kw)
        return (time.perf_counter() - t0) / nsteps, ls

    try:  # config 4: config 2 + l1_H=0.1 l2_H=0.2 l1_W=0.1 l2_W=0.5 (README.md:52)
        rule.upload(W0, H0)
        c4 = CONFIGS[4]
        kw = dict(l1W=c4["l1_W"], l2W=c4["l2_W"], l1H=c4["l1_H"], l2H=c4["l2_H"])
        dt, ls = mu_time(rule, 5, kw)
        res["configs[3]"] = {"workload": "N=2000 T=50000 K=32 L=20 alg=:mult l1_H=0.1 l2_H=0.2 l1_W=0.1 l2_W=0.5", "steps": 5, "warmup": 1,
                             "ms_per_step": 1e3 * dt, "iters_per_s": 1.0 / dt, "loss_last": float(ls[-1])}
    except Exception as e:  # noqa: BLE001 - side measurements never cost the headline
        res["configs[3]"] = {"error": repr(e)}
    try:  # config 5: alg=:hals
        hr = cmf.HALSUpdate(data, W0, H0, device=device)
        try:
            hals_steps(hr, 2, zero)
            # the timed steps carry no event pairs (one around each of a HALS iteration's ~20 launches idles the device ~0.2 ms per
            # iteration; the MU headline does the same: only its dominant kernel is bracketed inside the timed steps) ...
            t0 = time.perf_counter()
            ls = hals_steps(hr, 10, zero)
            dt = (time.perf_counter() - t0) / 10
            # ... the per-kernel table comes from a second pass of 5 bracketed iterations
            hr.set_option("profile", 1)
            t0 = time.perf_counter()
            hals_steps(hr, 5, zero)
            dt_prof = (time.perf_counter() - t0) / 5
            spans = {nm: hr.kernel_times(nm) for nm in ("hals_h_pipeline", "hals_w_sweep")}
            # the MFMA contraction launches of a HALS iteration (two thirds of it), in-loop HIP event times like the MU table:
            # each is ONE contraction of 2*K*N*S flop (residual conv = tensor_conv + (est - data) + loss, hals.jl:41;
            # hxt on the residual = G of the W sweep, hals.jl:104-110; transconv(W, data) = half of P, hals.jl:139-152)
            f1 = 2.0 * K * N * (L * T - L * (L - 1) / 2)
            hk = {}
            for nm, fl in (("conv_resid", f1), ("hxt_resid", f1), ("transconv_1src", f1), ("hxt_hh", 2.0 * K * 32 * ((K + 31) // 32) * (L * T - L * (L - 1) / 2)),
                           ("gram_denom_h", None), ("gram_tables", None), ("hals_h_pipeline", None), ("hals_w_sweep", None)):
                kms, n_ = hr.kernel_times(nm)
                if n_:
                    hk[nm] = {"avg_ms": kms, "launches": n_, "share_of_step": kms * (n_ / 5.0) / (1e3 * dt_prof)}
                    if fl:
                        hk[nm].update(flops_per_launch=fl, tflops=fl / kms / 1e9, frac=fl / kms / 1e9 / PEAK_FP32_MFMA_TFLOPS)
            hr.set_option("profile", 0)
            res["configs[4]"] = {"workload": "N=2000 T=50000 K=32 L=20 alg=:hals", "steps": 10, "warmup": 2, "ms_per_step": 1e3 * dt,
                                 "ms_per_step_bracketed": 1e3 * dt_prof,
                                 "chase": "the first ~64 % of the residual conv's tile rows chase the H row pipeline on the CUs it leaves free "
                                          "(option hals_chase; conv_resid below averages the chasing launch and the rest)",
                                 "iters_per_s": 1.0 / dt, "loss_last": float(ls[-1]), "metric": "HALS iters/sec",
                                 "pipeline_reruns": hr.counter("hals_pipeline_reruns"),
                                 "kernels": hk,
                                 "kernels_note": "in-loop HIP event pairs (option profile) over 5 iterations after the timed ones; frac = flops_per_launch / "
                                                 "avg_ms / 157.3 TFLOP/s for the contraction launches; the two sweeps are dependency-latency bound (roofline block)",
                                 "roofline": hals_roofline(T, K, L, spans, 1e3 * dt)}
        finally:
            hr.close()
    except Exception as e:  # noqa: BLE001
        res["configs[4]"] = {"error": repr(e)}
    # The shapes the reference itself publishes on have FEW components (README.md: K = 5; figures/fast_bcd/synthetic_comparison.jl:58-64:
    # N = 250, K = 5, L = 20, T up to 50000), and configs[0] is K = 5 too: they run on the few-component kernels (csrc/cmf_small_k.h,
    # the MFMA axes carry the flattened (lag, component) index).  Reported with the roofline on USEFUL flops (2*K*N*S per
    # contraction, 6 executed per iteration), next to the general kernels (option small_k = 0: K padded to a 32-wide MFMA axis).
    def few_components(key, N_, T_, K_, L_, what):
        try:
            d_ = cmf.gen_synthetic(N=N_, T=T_, seed=1234, device=device)
            W_, H_ = cmf.init_rand(d_, L=L_, K=K_, seed=0, device=device)
            f1_ = 2.0 * K_ * N_ * (L_ * T_ - L_ * (L_ - 1) / 2)
            rec = {"workload": f"N={N_} T={T_} K={K_} L={L_} alg=:mult ({what})", "steps": 200, "warmup": 20, "useful_flops_per_contraction": f1_}
            for small in (1, 0):
                r_ = cmf.MultUpdate(d_, W_, H_, device=device)
                try:
                    r_.set_option("small_k", small)
                    r_.iterate(20, **zero)  # (iterations of 0.06-0.3 ms: the first ten milliseconds of a loop run 4 % slower than its steady state)
                    r_.synchronize()
                    t0 = time.perf_counter()
                    ls = r_.iterate(200, **zero)
                    r_.synchronize()
                    dt_ = (time.perf_counter() - t0) / 200
                    if small:
                        ks = {}
                        for nm in ("conv_t", "conv_loss_store", "hxt", "transconv"):
                            ms_, _ = r_.time_kernel(nm, reps=10)
                            fl_ = f1_ * (2.0 if nm in ("hxt", "transconv") else 1.0)
                            ks[nm] = {"avg_ms": ms_, "useful_tflops": fl_ / ms_ / 1e9, "frac": fl_ / ms_ / 1e9 / PEAK_FP32_MFMA_TFLOPS}
                        dom = max(ks, key=lambda k_: ks[k_]["avg_ms"])
                        fused_ = r_.counter("small_k_fused_h_updates") > 0  # (the element-wise H update inside the C3 launch: long launches only)
                        rec.update(launches_per_step=5 if fused_ else 6,
                                   launches="C2 | slab sum + W update (+ the carried loss reduction) | conv_t | C3" + (" + H update" if fused_ else " | H update") + " | loss conv")
                        rec.update(ms_per_step=1e3 * dt_, iters_per_s=1.0 / dt_, loss_last=float(ls[-1]), kernels_standalone=ks,
                                   whole_iteration_useful_mfma_frac=6.0 * f1_ / dt_ / (PEAK_FP32_MFMA_TFLOPS * 1e12),
                                   roofline={"bound": "mfma", "kernel": dom + " (few-component kernels; hxt / transconv: both sources, incl. their slab sum / fold)",
                                             "achieved": ks[dom]["useful_tflops"], "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ks[dom]["frac"],
                                             "traffic": None, "flops": "useful: 2*K*N*S per contraction and source",
                                             "timing": "cmf_time_kernel: HIP events around 10 stand-alone launches"})
                    else:
                        rec.update(ms_per_step_general_kernels=1e3 * dt_, speedup_over_general_kernels=dt_ / (1e-3 * rec["ms_per_step"]))
                finally:
                    r_.close()
            res[key] = rec
        except Exception as e:  # noqa: BLE001
            res[key] = {"error": repr(e)}

    # What one rank of the T-sharded run computes per iteration, measured on this one GPU: the same problem with T / 2, T / 4,
    # T / 8 columns (a shard's kernels; its halo columns and the collectives are not in it), >= 200 back-to-back iterations
    # each, in the reference formulation and in the Gram form (option gram = 1).  `speedup_before_communication`
    # = the full problem's ms_per_step / the shard's: the ceiling the 8-GPU strong-scaling figure starts from.
    def shard(div):
        Ts = T // div
        rec = {"T_shard": Ts, "steps": 200, "warmup": 5}
        try:
            d_ = cmf.gen_synthetic(N=N, T=Ts, seed=1234, device=device)
            W_, H_ = cmf.init_rand(d_, L=L, K=K, seed=0, device=device)
            r_ = cmf.MultUpdate(d_, W_, H_, device=device)
            try:
                f1_ = 2.0 * K * N * (L * Ts - L * (L - 1) / 2)
                for gram in (0, 1):
                    r_.upload(W_, H_)
                    r_.set_option("gram", gram)
                    r_.iterate(5, **zero)
                    r_.synchronize()
                    if not gram:
                        r_.set_option("profile", 4)
                    t0 = time.perf_counter()
                    r_.iterate(200, **zero)
                    r_.synchronize()
                    dt_ = (time.perf_counter() - t0) / 200
                    if gram:
                        rec["ms_per_step_gram"] = 1e3 * dt_
                    else:
                        ks = {}
                        for nm in ("conv_t", "conv_loss_store", "hxt", "transconv"):
                            kms, n_ = r_.kernel_times(nm)
                            if n_:
                                fl_ = f1_ * (2.0 if nm in ("hxt", "transconv") else 1.0)
                                ks[nm] = {"avg_ms": kms, "launches": n_, "frac": fl_ / kms / 1e9 / PEAK_FP32_MFMA_TFLOPS}
                        r_.set_option("profile", 0)
                        rec.update(ms_per_step=1e3 * dt_, kernels=ks, whole_iteration_mfma_frac=6.0 * f1_ / dt_ / (PEAK_FP32_MFMA_TFLOPS * 1e12))
                r_.set_option("gram", 0)
            finally:
                r_.close()
        except Exception as e:  # noqa: BLE001
            rec["error"] = repr(e)
        return rec

    res["shards"] = {"what": "per-rank compute of the T-sharded run on ONE GPU (T/2, T/4, T/8 columns of config 2; no halo, no collective)",
                     "T/2": shard(2), "T/4": shard(4), "T/8": shard(8)}
    few_components("reference_protocol_shape", 250, 50000, 5, 20, "figures/fast_bcd/synthetic_comparison.jl:58-64")
    few_components("configs[0]", CONFIGS[1]["N"], CONFIGS[1]["T"], CONFIGS[1]["K"], CONFIGS[1]["L"],
                   "BASELINE.json configs[0]: the reference's CPU-runnable case; a problem this small is launch-latency bound on a GPU")
    if with_config3:
        try:  # config 3's problem on ONE GPU (its 8-GPU form gives every GPU exactly the config-2 shard)
            c3 = CONFIGS[3]
            t0 = time.perf_counter()
            d3 = cmf.gen_synthetic(N=c3["N"], T=c3["T"], seed=1234, device=device)
            W3, H3 = cmf.init_rand(d3, L=c3["L"], K=c3["K"], seed=0, device=device)
            r3 = cmf.MultUpdate(d3, W3, H3, device=device)
            setup = time.perf_counter() - t0
            try:
                dt, ls = mu_time(r3, 2, zero)
                F3 = flops_per_iter(c3["N"], c3["T"], c3["K"], c3["L"]) * 6.0 / 7.0
                res["configs[2]"] = {"workload": "N=2000 T=400000 K=32 L=20 alg=:mult, the whole problem on 1xMI355X (unsharded)", "steps": 2,
                                     "warmup": 1, "ms_per_step": 1e3 * dt, "iters_per_s": 1.0 / dt, "loss_last": float(ls[-1]),
                                     "whole_iteration_mfma_frac": F3 / dt / (PEAK_FP32_MFMA_TFLOPS * 1e12), "setup_s": setup}
            finally:
                r3.close()
        except Exception as e:  # noqa: BLE001
            res["configs[2]"] = {"error": repr(e)}
    return res


def measure_call_by_call(rule, W0, H0, reg_kw, nsteps, sync):
    """ms per iteration of the reference's own loop shape (alternating.jl:51-59: update_motifs!; loss = update_feature_maps!,
    the loss read by the host every iteration) in three forms: no write-back (W, H stay on the device: `sync_every_call=false`
    + one download after the loop), write-back through cmf_arm_writeback (CMFHip.jl's default since round 5), and a
    synchronous cmf_get_factors after every call (its default until round 4).  The arrays are checked against each other."""
    import numpy as np
    from cmf_jl_amd._lib import check, ptr

    K, N, L = W0.shape
    T = H0.shape[1]
    W = np.zeros((K, N, L), order="F")
    H = np.zeros((K, T), order="F")
    rec = {"steps": nsteps, "what": "Python loop of cmf_update_motifs + cmf_update_feature_maps (loss synchronous), alternating.jl:51-59"}
    kw_w = dict(l1W=reg_kw["l1W"], l2W=reg_kw["l2W"])
    kw_h = dict(l1H=reg_kw["l1H"], l2H=reg_kw["l2H"])

    def loop(n, after=None):
        for _ in range(n):
            rule.update_motifs(None, W, H, **kw_w)
            rule.update_feature_maps(None, W, H, **kw_h)
            if after:
                after()

    def timed_loop(after=None):
        loop(2, after)
        sync()
        t0 = time.perf_counter()
        loop(nsteps, after)
        sync()
        return 1e3 * (time.perf_counter() - t0) / nsteps

    rule.upload(W0, H0)
    rule.sync_every_call = False
    rec["ms_per_step_call_by_call"] = timed_loop()
    rule.upload(W0, H0)
    W[...] = W0  # (under sync_every_call the rule reads the arrays it is handed: they start as the initial factors, like `fit`'s copies)
    H[...] = H0
    rule.sync_every_call = True
    try:
        rec["ms_per_step_call_by_call_writeback"] = timed_loop()
    finally:
        rule.sync_every_call = False
    Wd, Hd = rule.download()
    rec["writeback_equals_get_factors"] = bool(np.array_equal(W, Wd) and np.array_equal(H, Hd))
    rec["writeback_overlapped_calls"] = rule.counter("writeback_overlapped")
    rule.upload(W0, H0)
    rec["ms_per_step_call_by_call_get_factors"] = timed_loop(lambda: check(rule._lib.cmf_get_factors(rule._h, ptr(W), ptr(H))))
    rec["writeback_bytes_per_step_fp64"] = 8 * (W.size + H.size)
    return rec


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--T", type=int, default=0, help="override T (debugging only; invalidates the metric)")
    ap.add_argument("--sustain", type=float, default=8.0, help="seconds of back-to-back iterations after the timed steps (0 = skip)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the no-reuse / Gram-form / other-config side measurements (profiling runs: keeps one launch shape per kernel)")
    ap.add_argument("--no-config3", action="store_true", help="skip config 3 (T=400000 on one GPU) in other_configs")
    ap.add_argument("--attempt-timeout", type=float, default=float(os.environ.get("CMF_BENCH_ATTEMPT_TIMEOUT", "300")),
                    help="--gpus N > 1: seconds one attempt (one form of the ladder, in its own child process) may take")
    ap.add_argument("--child", default="", help=argparse.SUPPRESS)  # set by the supervisor: this process IS the measurement
    return ap.parse_args(argv)


# ---- supervisor (multi-GPU runs): the measurement in a child process per attempt --------------------------------------
LADDER = {
    # plain `python bench.py --gpus N`: (label, form of the child, extra environment)
    "multi": [("one process, an enqueue thread per GPU, RCCL", "multi", {}),
              ("one process, the calling thread enqueues every GPU, grouped RCCL calls", "multi", {"CMF_ENQUEUE_THREADS": "0"}),
              ("one process per GPU (started by bench.py), the library's RCCL communicator", "ranks", {}),
              ("one process per GPU (started by bench.py), collectives through torch.distributed's RCCL backend", "ranks", {"CMF_TRANSPORT": "host"})],
    # under a launcher (WORLD_SIZE == --gpus)
    "ranks": [("the library's RCCL communicator", "ranks", {}),
              ("collectives through torch.distributed's RCCL backend on staged buffers", "ranks", {"CMF_TRANSPORT": "host"}),
              ("collectives through gloo on host buffers", "ranks", {"CMF_TRANSPORT": "host", "CMF_DIST_BACKEND": "gloo"})],
}


def free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def child_command(args, tag):
    # (CMF_BENCH_FAKE_CHILD: the CPU tests of the supervisor put a stub in the measurement's place -- honoured only together with
    # CMF_TEST_HOOKS=1, like the library's own test knobs, and the command that ran is recorded in `attempts`)
    fake = os.environ.get("CMF_BENCH_FAKE_CHILD") if os.environ.get("CMF_TEST_HOOKS") == "1" else None
    cmd = [sys.executable, fake or os.path.abspath(__file__), "--child", tag, "--gpus", str(args.gpus), "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--config", str(args.config), "--cpu-seconds", str(args.cpu_seconds), "--sustain", str(args.sustain)]
    if args.T:
        cmd += ["--T", str(args.T)]
    if args.no_extras:
        cmd.append("--no-extras")
    if args.no_config3:
        cmd.append("--no-config3")
    return cmd


def child_env(extra, rank=None, world=None, port=None):
    env = dict(os.environ)
    for k in list(env):  # the child makes its own rendezvous: nothing of the launcher's elastic agent may leak into it
        if k.startswith("TORCHELASTIC_") or k in ("GROUP_RANK", "ROLE_RANK", "ROLE_NAME", "LOCAL_WORLD_SIZE", "GROUP_WORLD_SIZE", "ROLE_WORLD_SIZE"):
            env.pop(k)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("CMF_WAIT_TIMEOUT_S", "90")  # a collective that never completes ends the attempt with a message, well inside its time limit
    env["CMF_BENCH_SUPERVISE"] = "0"
    env.update(extra)
    if rank is None:  # one process drives all GPUs
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
    else:
        env.update(RANK=str(rank), LOCAL_RANK=str(rank if "LOCAL_RANK" not in os.environ or world is None else os.environ.get("LOCAL_RANK", rank)),
                   WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    return env


LIVE_CHILDREN = []  # children of the attempt in flight: ended too when the supervisor itself is told to stop


def stop_children(procs):
    """End exactly the processes started here (each its own session: its process group goes with it)."""
    import signal

    for p_ in procs:
        if p_.poll() is None:
            try:
                os.killpg(p_.pid, signal.SIGTERM)
            except OSError:
                pass
    t_end = time.time() + 10
    for p_ in procs:
        while p_.poll() is None and time.time() < t_end:
            time.sleep(0.1)
        if p_.poll() is None:
            try:
                os.killpg(p_.pid, signal.SIGKILL)
            except OSError:
                pass
            p_.wait()


def last_json_line(text):
    for line in reversed(text.splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                return json.loads(line)
            except ValueError:
                continue
    return None


def run_attempt(cmds_envs, timeout, should_abort=None):
    """Start the children of one attempt (each in its own session), wait for all of them -- at most `timeout` seconds, and
    no longer than `should_abort()` allows -- and return (ok, reason, stdout of child 0, tail of the stderrs)."""
    import subprocess
    import tempfile

    procs, outs, errs = [], [], []
    for cmd, env in cmds_envs:
        fo, fe = tempfile.TemporaryFile("w+"), tempfile.TemporaryFile("w+")
        outs.append(fo)
        errs.append(fe)
        procs.append(subprocess.Popen(cmd, env=env, stdout=fo, stderr=fe, start_new_session=True, cwd=ROOT))
    LIVE_CHILDREN[:] = procs
    t0, reason = time.time(), None
    while True:
        codes = [p_.poll() for p_ in procs]
        if any(c not in (None, 0) for c in codes):
            bad = [(i, c) for i, c in enumerate(codes) if c not in (None, 0)]
            reason = "child process(es) failed: " + ", ".join(f"#{i} exit code {c}" for i, c in bad)
            break
        if all(c == 0 for c in codes):
            break
        if time.time() - t0 > timeout:
            reason = f"no result within {timeout:.0f} s (--attempt-timeout)"
            break
        if should_abort is not None and should_abort():
            reason = "another rank's attempt failed"
            break
        time.sleep(0.2)
    stop_children(procs)
    LIVE_CHILDREN[:] = []

    def text(f):
        f.seek(0)
        return f.read()

    out0 = text(outs[0])
    err_tail = []
    for i, fe in enumerate(errs):
        lines = [ln for ln in text(fe).splitlines() if ln.strip()]
        if lines and (reason or i == 0):
            err_tail.append({"child": i, "stderr_tail": lines[-6:]})
    for f in outs + errs:
        f.close()
    fatal = any(c == 2 for c in codes)  # exit code 2: this box cannot run the job at all (e.g. fewer GPUs than --gpus): no other form can
    if fatal:
        for e_ in err_tail:
            for ln in e_["stderr_tail"]:
                print(ln, file=sys.stderr, flush=True)
    return (reason is None, "FATAL: " + (reason or "") if fatal else reason, out0, err_tail)


def supervise(args, form):
    """See the module docstring ("First contact").  This process never initialises the GPU."""
    import signal

    def on_term(signum, _frame):  # a launcher that gives up on this rank must not leave its measurement child on the GPU
        stop_children(list(LIVE_CHILDREN))
        sys.exit(128 + signum)

    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sig, on_term)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1") or 1)
    store = None
    if form == "ranks":
        # the supervising ranks agree through a TCP store: the launcher's own (torch.distributed.run hosts one on MASTER_PORT
        # and sets TORCHELASTIC_USE_AGENT_STORE), else one hosted by rank 0
        from datetime import timedelta

        from torch.distributed import PrefixStore, TCPStore

        agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "").lower() == "true"
        try:
            base = TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ["MASTER_PORT"]), world_size=None if agent else world,
                            is_master=(rank == 0 and not agent), timeout=timedelta(seconds=args.attempt_timeout + 60), wait_for_workers=False)
            store = PrefixStore(f"cmfbench/{os.environ.get('TORCHELASTIC_RUN_ID', 'run')}/{os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')}/", base)
            store.set(f"hello/{rank}", "1")
            store.wait([f"hello/{r}" for r in range(world)])  # every supervising rank is there: from here on they walk the ladder together
        except Exception as e:  # noqa: BLE001 - no store, no agreement: every rank measures in the started process, as before round 4
            print(f"bench.py rank {rank}: the supervising ranks could not meet through the launcher's store ({e!r}); measuring in this process",
                  file=sys.stderr, flush=True)
            return False
    attempts, line = [], None
    t_ladder = time.time()

    def say(msg):  # one flushed line per rung on stderr: a driver that kills the run at ITS limit still has the diagnosis in stderr_tail
        print(f"bench.py supervisor rank {rank} [{time.time() - t_ladder:6.1f} s] {msg}", file=sys.stderr, flush=True)

    for a, (label, child_form, extra) in enumerate(LADDER[form]):
        rec = {"form": label, "env": extra, "child": os.path.basename(child_command(args, child_form)[1])}
        # a total budget over the ladder: the forms that are left share what remains of (rungs x attempt timeout), never more than
        # --attempt-timeout each -- the worst case stays bounded however the earlier rungs ended
        extra = dict(extra, CMF_BENCH_ATTEMPT=str(a))
        t0 = time.time()
        say(f"attempt {a + 1}/{len(LADDER[form])}: {label} (limit {args.attempt_timeout:.0f} s)")
        if form == "multi" and child_form == "multi":
            ok, reason, out0, errs = run_attempt([(child_command(args, "multi"), child_env(extra))], args.attempt_timeout)
        elif form == "multi":  # bench.py is its own launcher: one child per GPU
            port = free_port()
            ok, reason, out0, errs = run_attempt([(child_command(args, "ranks"), child_env(extra, r, args.gpus, port)) for r in range(args.gpus)],
                                                 args.attempt_timeout)
        else:
            if rank == 0:
                store.set(f"port{a}", str(free_port()))
            port = int(store.get(f"port{a}").decode())
            env = child_env(extra, rank, world, port)
            env["LOCAL_RANK"] = os.environ.get("LOCAL_RANK", str(rank))

            def should_abort():
                try:
                    return bool(store.check([f"failed{a}"]))
                except Exception:  # noqa: BLE001 - a store hiccup must not end a healthy attempt
                    return False

            ok, reason, out0, errs = run_attempt([(child_command(args, "ranks"), env)], args.attempt_timeout, should_abort)
            if ok and rank == 0:  # rank 0's outcome includes "its child printed a line with a value": all ranks must agree on THAT
                got0 = last_json_line(out0)
                if got0 is None or got0.get("value") is None:
                    ok, reason = False, "no JSON line with a value on the child's stdout"
            if not ok:
                store.set(f"failed{a}", "1")
            if not ok and (reason or "").startswith("FATAL"):
                store.set(f"fatal{a}", "1")
            store.set(f"done{a}/{rank}", "ok" if ok else (reason or "failed"))
            try:
                store.wait([f"done{a}/{r}" for r in range(world)])
                states = [store.get(f"done{a}/{r}").decode() for r in range(world)]
            except Exception as e:  # noqa: BLE001 - a rank that never reports (its child hung past every limit) fails the attempt, not the supervisor
                states = [f"no report through the store within {args.attempt_timeout + 60:.0f} s ({type(e).__name__})" if r != rank else
                          ("ok" if ok else (reason or "failed")) for r in range(world)]
            if ok and any(st_ != "ok" for st_ in states):
                ok, reason = False, "; ".join(f"rank {r}: {st_}" for r, st_ in enumerate(states) if st_ != "ok")
            if any(st_.startswith("FATAL") for st_ in states) and not (reason or "").startswith("FATAL"):
                reason = "FATAL: " + (reason or "")
        rec.update(ok=ok, seconds=round(time.time() - t0, 1))
        say(f"attempt {a + 1}: {'ok' if ok else 'FAILED: ' + str(reason)} after {rec['seconds']} s"
            + ("" if ok or not errs else "; child stderr: " + " | ".join(ln for e_ in errs for ln in e_["stderr_tail"][-2:])))
        got = last_json_line(out0) if rank == 0 else None
        if not ok:
            rec["ended"] = reason
            rec["children"] = errs
            if got is not None:  # the child's own failure record: phase, cmf_last_error, comm, what it had measured
                rec["child_line"] = {k: got.get(k) for k in ("failed_phase", "error", "cmf_last_error", "comm", "ms_per_step", "partial") if k in got}
        attempts.append(rec)
        if not ok and (reason or "").startswith("FATAL"):
            sys.exit(2)  # (the child's message is on stderr; nothing on stdout, as before the supervisor existed)
        if ok and (rank != 0 or (got is not None and got.get("value") is not None)):
            line = got
            break
        if ok:  # every child exited 0 but rank 0's printed no usable line
            attempts[-1]["ok"], attempts[-1]["ended"] = False, "no JSON line with a value on the child's stdout"
    if store is not None:  # the rank that hosts the store leaves last
        try:
            store.set(f"bye/{rank}", "1")
            if rank == 0:
                store.wait([f"bye/{r}" for r in range(world)])
        except Exception:  # noqa: BLE001 - the run is over either way
            pass
    if rank != 0:
        sys.exit(0 if line is not None or attempts[-1]["ok"] else 3)
    if line is not None:
        line["attempts"] = attempts
        print(json.dumps(line), flush=True)
        sys.exit(0)
    cfg = CONFIGS[args.config]
    print(json.dumps({"metric": "MU iters/sec (convolutive NMF multiplicative update; achieved HBM GB/s in `hbm`)", "value": None, "unit": "iter/s",
                      "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
                      "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                      "config": {"workload": f"configs[{args.config - 1}]: N={cfg['N']} T={args.T or cfg['T']} K={cfg['K']} L={cfg['L']} fp32", "launch": form},
                      "failed_phase": "every form of the ladder failed", "attempts": attempts}), flush=True)
    sys.exit(3)


def main():
    args = parse_args()
    try:
        form = route(args.gpus, os.environ.get("WORLD_SIZE", "1"))
    except ValueError as e:
        raise SystemExit(str(e))
    if args.child:
        form = args.child
    elif args.gpus > 1 and os.environ.get("CMF_BENCH_SUPERVISE", "1") != "0":
        if supervise(args, form) is not False:  # (it exits with the line printed; False: it could not set itself up)
            return
    progress = {"phase": "start", "rank": int(os.environ.get("RANK", "0"))}
    try:
        measure(args, form, progress)
    except SystemExit:
        raise
    except BaseException as e:  # noqa: BLE001 - first contact: whatever went wrong, say where, and what was known by then
        import traceback

        traceback.print_exc()
        rec = {"metric": "MU iters/sec (convolutive NMF multiplicative update; achieved HBM GB/s in `hbm`)", "value": None, "unit": "iter/s",
               "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"configs[{args.config - 1}]", "launch": form},
               "failed_phase": progress.get("phase"), "error": repr(e), "rank": progress.get("rank")}
        try:
            rec["cmf_last_error"] = progress["lib"].cmf_last_error().decode() if progress.get("lib") else None
        except Exception:  # noqa: BLE001
            rec["cmf_last_error"] = None
        for k in ("comm", "partial"):
            if k in progress:
                rec[k] = progress[k]
        if "rule" in progress and "comm" not in rec:
            try:
                rec["comm"] = parse_comm(progress["rule"].comm_info(), form)
            except Exception:  # noqa: BLE001
                pass
        # every rank says what it saw (stderr); the JSON line is rank 0's -- or the failing rank's own when rank 0 is not it
        print(f"bench.py rank {progress.get('rank')}: failed in phase '{progress.get('phase')}': {e!r}", file=sys.stderr, flush=True)
        print(json.dumps(rec), flush=True)
        sys.exit(3)


def measure(args, form, progress):
    # the oracle's OpenMP loops (HALS baseline) are short: one thread per visible core only adds spinning
    os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))
    # RCCL shares device buffers between processes through dmabuf IPC; the host driver of these boxes supports only that
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = args.gpus if form == "ranks" else 1   # processes
    ngpu = args.gpus                              # GPUs of the job

    import numpy as np

    import __graft_entry__

    cfg = dict(CONFIGS[args.config])
    if args.T:
        cfg["T"] = args.T
    N, T, K, L = cfg["N"], cfg["T"], cfg["K"], cfg["L"]
    alg = cfg.pop("alg", "mult")
    reg = {k: v for k, v in cfg.items() if k.startswith("l")}
    reg_kw = dict(l1W=reg.get("l1_W", 0.0), l2W=reg.get("l2_W", 0.0), l1H=reg.get("l1_H", 0.0), l2H=reg.get("l2_H", 0.0))

    dist = None
    backend = None
    if form == "ranks":
        import torch
        import torch.distributed as dist

        # The process group is only the rendezvous (ncclUniqueId hand-over, barriers, the max over ranks of the
        # elapsed time): the data path's collectives are the library's own RCCL calls.  CMF_DIST_BACKEND=gloo with
        # CMF_TRANSPORT=host rehearses the multi-rank path on a box with fewer GPUs than ranks.
        backend = os.environ.get("CMF_DIST_BACKEND", "nccl")
        local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if rank == 0:
        __graft_entry__.build(quiet=True)  # no-op when the in-tree .so files are current
    if form == "ranks":
        dist.barrier()
    import cmf_jl_amd as cmf

    lib = cmf.load_library()
    progress.update(lib=lib, phase="inputs (gen_synthetic, init_rand)")
    # CMF_BENCH_DEVICES="0,0,0,0": rehearse the one-process form on a box with fewer GPUs than shards (the same device may
    # be listed several times: loopback transport) -- everything of the multi-GPU code path but RCCL itself
    multi_devices = list(range(ngpu))
    if form == "multi" and os.environ.get("CMF_BENCH_DEVICES"):
        multi_devices = [int(x) for x in os.environ["CMF_BENCH_DEVICES"].split(",")]
        if len(multi_devices) != ngpu:
            raise SystemExit(f"CMF_BENCH_DEVICES lists {len(multi_devices)} devices, --gpus is {ngpu}")
    if form == "multi" and lib.cmf_device_count() <= max(multi_devices):
        print(f"bench.py --gpus {ngpu}: only {lib.cmf_device_count()} HIP device(s) visible "
              f"(cmf_last_error: {lib.cmf_last_error().decode() or 'none'})", file=sys.stderr, flush=True)
        sys.exit(2)

    # ---- synthetic inputs: gen_synthetic(seed 1234) + init_rand(seed 0), SURVEY.md section 8d ----
    # Every rank generates the same arrays (counter-based RNG) and keeps its own T block.
    data = cmf.gen_synthetic(N=N, T=T, seed=1234, device=local_rank)
    W0, H0 = cmf.init_rand(data, L=L, K=K, seed=0, device=local_rank)

    replicas = []
    progress["phase"] = "rule / group construction"
    # the overlap form (its bulk all-reduce on a second stream with a communicator of its own) is probed against the plain
    # form from 4 GPUs on, where the all-reduce is a visible share of the step: both times are recorded, the faster is kept
    overlap_env = os.environ.get("CMF_ALLREDUCE_OVERLAP", "probe" if ngpu >= 4 else "0")
    if alg == "hals":
        # replicas only: the H sweep is one dependent chain along T, nothing to exchange
        devs = multi_devices if form == "multi" else [local_rank]
        replicas = [cmf.HALSUpdate(data, W0, H0, device=d) for d in devs]
        rule = replicas[0]
    elif form == "single":
        rule = cmf.MultUpdate(data, W0, H0, device=local_rank)
    elif form == "multi":
        # ONE process, ngpu devices: cmf_create_multi -> RCCL communicators from ncclCommInitAll, a stream per device
        # CMF_BENCH_TRANSPORT: rccl (default for distinct devices) | peer (direct xGMI reads / writes between event fences,
        # opt-in) | loopback | loopback-streams (rehearsals with CMF_BENCH_DEVICES listing one device several times)
        tr_name = os.environ.get("CMF_BENCH_TRANSPORT", "auto")
        tr_codes = {"auto": 0, "rccl": 1, "loopback": 2, "loopback-streams": 3, "peer": 4}
        if tr_name not in tr_codes:
            raise SystemExit(f"CMF_BENCH_TRANSPORT must be one of {sorted(tr_codes)}")
        try:
            rule = cmf.MultUpdate(data, W0, H0, devices=multi_devices, transport=tr_codes[tr_name])
        except cmf.CMFError as e:
            print(f"bench.py --gpus {ngpu}: the {ngpu}-device group could not be formed: {e}", file=sys.stderr, flush=True)
            sys.exit(2)
    else:
        from cmf_jl_amd.sharded import ShardedMultUpdate

        # in-library RCCL; if its communicator cannot be formed on some rank, all ranks agree to take the host-collective
        # transport (torch.distributed on staged buffers) so that the run still measures something -- reported in `comm`
        rule = ShardedMultUpdate(data, W0, H0, device=local_rank, fallback_to_host=True,
                                 transport=os.environ.get("CMF_TRANSPORT", "rccl" if backend == "nccl" else "host"))

    def sync():
        for r in (replicas or [rule]):
            r.synchronize()  # every stream of every local shard
        if form == "ranks":
            import torch

            dist.barrier()
            torch.cuda.synchronize()

    def run_steps(n):
        """n steps = n x (update_motifs!; update_feature_maps!): the losses, all read by the host before this returns."""
        if n <= 0:
            return []
        if alg == "mult":
            return list(rule.iterate(n, **reg_kw))  # cmf_iterate
        if len(replicas) > 1:  # independent replicas, one host thread each
            import threading

            outs = [None] * len(replicas)

            def work(i):
                outs[i] = hals_steps(replicas[i], n, reg_kw)

            th = [threading.Thread(target=work, args=(i,)) for i in range(len(replicas))]
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
            return outs[0]
        return hals_steps(rule, n, reg_kw)

    def timed(nwarm, nsteps):
        run_steps(nwarm)
        sync()
        t0 = time.perf_counter()
        ls = run_steps(nsteps)
        sync()
        el = time.perf_counter() - t0
        if form == "ranks":
            import torch

            tmax = torch.tensor([el], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        return el, ls

    progress.update(rule=rule, phase="first loss (the group's first collectives)")
    loss0 = rule.compute_loss()
    probe = None
    sharded = alg == "mult" and ngpu > 1
    if sharded:
        # Two forms of the W phase exist (cmf_groups.hip): one all-reduce of [numW | denomW] after both contractions, or
        # numW contracted and all-reduced right after the H update, underneath the loss conv and the denominator
        # contraction.  "0" (default): everything on one stream, the plain single all-reduce; "1": the overlap form;
        # "probe": time a few steps of each (max over ranks, so every rank takes the same decision), keep the faster.
        if overlap_env in ("0", "1"):
            rule.set_overlap(overlap_env == "1")
        else:
            progress["phase"] = "overlap probe"
            probe = {}
            for f_ in (False, True):
                try:
                    rule.set_overlap(f_)
                    if f_ and not rule.overlap:  # refused (the second communicator could not be formed on some rank): stay single
                        probe["overlap_refused"] = str(getattr(rule, "overlap_refused", None))
                        break
                    timed(2, 0)
                    probe["overlap" if f_ else "single"] = timed(0, 5)[0] / 5
                except cmf.CMFError as e:  # the probe must never cost the headline
                    probe["overlap_error"] = repr(e)
                    break
            rule.set_overlap("overlap" in probe and probe["overlap"] < probe["single"])
            progress["partial"] = {"allreduce_overlap_probe_ms": {k: (1e3 * v if isinstance(v, float) else v) for k, v in probe.items()}}
    # The timed region carries HIP event pairs (option "profile": events on the launch stream) around every fourth launch of the
    # DOMINANT kernel only -- the roofline block's duration is measured live over these very steps -- because an event pair idles the
    # device a few microseconds: pairs around all four contraction kernels cost 0.5 % of the step.  The other kernels' durations (the
    # `kernels` table) are taken the same way over a second pass of the same K steps right behind the timed one.
    prof = rule
    one_class = alg == "mult" and form == "single"
    progress["phase"] = "warm-up steps"
    if sharded and os.environ.get("CMF_BENCH_HALO_IN_ALLREDUCE") == "0":  # (the all-gather form of the halo exchange: to time both on a node)
        rule.set_option("halo_in_allreduce", 0)
    # which kernel is the dominant one is MEASURED, not assumed: the warm-up steps run with every class bracketed and the class with the
    # largest mean duration is the one the timed steps bracket (bit i of "profile_mask" = the i-th class name of cmf_kernel_times)
    PROF_CLASSES = ("conv", "conv_t", "conv_loss", "conv_loss_store", "hxt", "transconv", "hxt_num", "hxt_den")
    dom_class = "transconv"
    if one_class and args.warmup > 0:
        prof.set_option("profile", 1)
        timed(args.warmup, 0)
        warm = {nm: prof.kernel_times(nm) for nm in PROF_CLASSES}
        warm = {nm: v[0] for nm, v in warm.items() if v[1]}
        prof.set_option("profile", 0)
        if warm:
            dom_class = max(warm, key=warm.get)
    else:
        timed(args.warmup, 0)
    progress["phase"] = "timed steps"
    prof.set_option("profile_mask", (1 << PROF_CLASSES.index(dom_class)) if one_class else 0)
    # (HALS: the timed steps carry no event pairs at all -- one around each of an iteration's ~20 launches idles the device ~0.2 ms per
    # iteration; the two sweeps' spans of the roofline block come from a bracketed pass of the same steps right behind the timed one)
    prof.set_option("profile", 4 if alg == "mult" else 0)  # every 4th launch of the bracketed class
    coll0 = None
    if sharded:
        try:  # the collectives this rank's handle issues over the timed steps (cmf_get_counter): the north star asks for ONE per iteration
            coll0 = (rule.counter("allreduce_calls"), rule.counter("allgather_calls"))
        except Exception:  # noqa: BLE001
            coll0 = None
    dt, losses = timed(0, args.steps)
    coll_per_step = None
    if coll0 is not None:
        try:
            coll_per_step = {"allreduce": (rule.counter("allreduce_calls") - coll0[0]) / max(1, args.steps),
                             "allgather": (rule.counter("allgather_calls") - coll0[1]) / max(1, args.steps),
                             "halo_in_allreduce": rule.counter("halo_in_allreduce"),
                             "note": "per iteration over the timed steps; the all-gathers are the flush of the batch's last loss (one per "
                                     "cmf_iterate call) and, in the all-gather form of the halo exchange (halo_in_allreduce = 0), one per iteration"}
        except Exception as e:  # noqa: BLE001
            coll_per_step = {"error": repr(e)}
    progress.setdefault("partial", {}).update(ms_per_step=1e3 * dt / args.steps, iters_per_s=args.steps / dt, loss_last=float(losses[-1]) if len(losses) else None)
    progress["phase"] = "side measurements after the timed steps"
    names = ("conv", "conv_t", "conv_loss", "conv_loss_store", "hxt", "hxt_num", "hxt_den", "transconv", "hals_h_pipeline", "hals_w_sweep")
    inloop = {}
    for name in names:
        kms, n = prof.kernel_times(name)
        if n:
            inloop[name] = (kms, n)
    prof.set_option("profile", 0)
    prof.set_option("profile_mask", 0)
    hals_spans = {}
    if alg != "mult":
        prof.set_option("profile", 1)
        timed(0, min(args.steps, 5))
        for name in names:
            kms, n = prof.kernel_times(name)
            if n:
                inloop[name] = (kms, n)
        prof.set_option("profile", 0)
        hals_spans, inloop = inloop, {}
    dt_allpairs = None
    if one_class:  # second pass: every contraction kernel bracketed (the dominant kernel keeps its figure from the timed steps)
        prof.set_option("profile", 4)
        dt_allpairs, _ = timed(0, args.steps)
        for name in names:
            kms, n = prof.kernel_times(name)
            if n and name not in inloop:
                inloop[name] = (kms, n)
        prof.set_option("profile", 0)
    # Steady state: the timed region above is ~0.1 s; run back-to-back iterations for several seconds more (same call) so that
    # the figure also holds at the clock the card settles to (and the driver's GPU-busy sampling has something to see).
    sustained = None
    if alg == "mult" and args.sustain > 0:
        per = max(dt / max(args.steps, 1), 1e-4)
        n_sus = int(min(max(args.sustain / per, args.steps), 200000))
        dt_sus, _ = timed(0, n_sus)
        sustained = {"seconds": dt_sus, "steps": n_sus, "ms_per_step": 1e3 * dt_sus / n_sus, "iters_per_s": n_sus / dt_sus}
    # Same loop with the reference's redundant est recomputation left in (7 executed contractions
    # instead of 6): reported beside the headline so both numbers come from one run.
    dt_noreuse = None
    extras = form == "single" and alg == "mult" and not args.no_extras
    if extras:
        rule.set_option("reuse_est", 0)
        dt_noreuse, _ = timed(1, max(3, args.steps // 2))
        dt_noreuse /= max(3, args.steps // 2)
        rule.set_option("reuse_est", 1)
    # The drop-in path as CMF.jl's own `fit` drives it (alternating.jl:51-59): two rule calls per iteration, the loss synchronous --
    # and, as CMFHip.jl does by default (sync_every_call), W and H written back into the caller's arrays by every
    # update_feature_maps! call (cmf_arm_writeback: the download under the call's own kernels), next to the plain
    # cmf_get_factors after every call that the binding used until round 4.
    call_by_call = None
    if extras:
        progress["phase"] = "call-by-call side measurements"
        try:
            call_by_call = measure_call_by_call(rule, W0, H0, reg_kw, max(10, args.steps), sync)
        except Exception as e:  # noqa: BLE001 - side measurements never cost the headline
            call_by_call = {"error": repr(e)}
        # ... and the same loop on the 8-shard partition `--gpus 8` runs, rehearsed on THIS GPU (loopback transport with a stream and
        # an enqueue worker per shard; the shards share the device, so the iteration is ~8 shard iterations long): the ABSOLUTE extra
        # time per iteration of each way to deliver W and H says what the reference's `fit` would pay per iteration on a node
        if args.config == 2 and not args.T and isinstance(call_by_call, dict) and "error" not in call_by_call:
            g8 = None
            try:
                g8 = cmf.MultUpdate(data, W0, H0, devices=[local_rank] * 8, transport=3)
                rec8 = measure_call_by_call(g8, W0, H0, reg_kw, 10, g8.synchronize)
                rec8["what"] = "8 loopback shards on one GPU (cmf_create_multi, a stream and an enqueue worker per shard): " + rec8["what"]
                call_by_call["rehearsal_8_loopback_shards"] = rec8
            except Exception as e:  # noqa: BLE001
                call_by_call["rehearsal_8_loopback_shards"] = {"error": repr(e)}
            finally:
                if g8 is not None:
                    g8.close()
    # Optional Gram form of the denominators (executes 2.3 + 1 contractions): reported as
    # extra fields only, the headline value is the reference formulation above.
    dt_gram = dt_gram2 = None
    if extras:
        nrep = max(3, args.steps // 2)
        rule.upload(W0, H0)
        rule.set_option("gram", 1)
        dt_gram, lg = timed(5, nrep)  # (a few warm-up iterations: the side measurements before this one leave the device idle for a moment)
        dt_gram /= nrep
        rule.set_option("gram", 2)
        dt_gram2, _ = timed(3, nrep)
        dt_gram2 /= nrep
        rule.set_option("gram", 0)

    # Sharded runs: the Gram form on the group -- the all-reduce carries [numW | HH | tail] (6.9 MB instead of 10.5 MB at
    # config 2) -- as extra fields (never part of `value`).  OPT-IN (CMF_BENCH_GROUP_EXTRAS=gram, or =1 for its overlap
    # variant too): RCCL with more than one rank has never run these group forms (no multi-GPU node was available to the
    # build), a collective that wedges here is not caught by the try/except below but only by the library's bounded wait,
    # and a side measurement must never be able to cost the headline of a multi-GPU run.  Switch it on once
    # tests/test_multi_gpu.py has passed on a node.
    group_extras = None
    ge_mode = os.environ.get("CMF_BENCH_GROUP_EXTRAS", "0")
    if sharded and ge_mode != "0" and not args.no_extras:
        group_extras = {}
        keep_overlap = bool(rule.overlap)
        nrep = max(5, args.steps // 2)
        try:
            for name, ov in ((("gram", False), ("gram_overlap", True)) if ge_mode == "1" else (("gram", False),)):
                rule.upload(W0, H0)
                rule.set_option("gram", 1)
                rule.set_overlap(ov)
                t_, ls_ = timed(2, nrep)
                group_extras[f"ms_per_step_{name}"] = 1e3 * t_ / nrep
                group_extras[f"loss_last_{name}"] = float(ls_[-1])
        except Exception as e:  # noqa: BLE001
            group_extras["error"] = repr(e)
        rule.set_option("gram", 0)
        rule.set_overlap(keep_overlap)

    # the group's bulk exchange alone (every rank takes part: it is a collective), so that the record says how much of a
    # step is communication
    allreduce = None
    collectives = None
    if sharded:
        progress["phase"] = "collectives timed alone"

        def coll(name, payload, ring=True):
            try:
                ms_, by_ = rule.time_kernel(name, reps=10)
                rec_ = {"avg_ms": ms_, "bytes": by_, "algbw_GBps": by_ / ms_ / 1e6 if ms_ > 0 else None, "payload": payload}
                if ring:
                    rec_["busbw_GBps"] = (by_ / ms_ / 1e6) * 2.0 * (ngpu - 1) / ngpu if ms_ > 0 else None
                return rec_
            except Exception as e:  # noqa: BLE001
                return {"error": repr(e)}

        allreduce = coll("allreduce", "[numW | denomW | loss tail]: what one iteration all-reduces")
        collectives = {"allreduce_gram_payload": coll("allreduce_gram", "[numW | HH | loss tail]: the Gram form's all-reduce (option gram = 1)"),
                       "allgather_halo": coll("allgather_halo", "every shard's [first | last] L-1 columns of H", ring=False)}
        if rule.overlap:
            collectives["allreduce_lane1"] = coll("allreduce_lane1", "numW on the communication stream and its own communicator (overlap form)")
        progress.setdefault("partial", {}).update(allreduce=allreduce, collectives=collectives)
        progress["phase"] = "report"

    out = None
    nrep_hals = len(replicas) if form == "multi" else (world if alg == "hals" else 1)
    if rank == 0:
        ms = 1e3 * dt / args.steps
        iters_per_s = args.steps / dt * (nrep_hals if alg == "hals" else 1)
        F_iter = flops_per_iter(N, T, K, L)
        if alg == "hals":
            # executed MFMA work of one HALS iteration: hxt (2 sources) + Gram of H_unfold (L*K32 columns instead
            # of N) + conv_t + transconv (2 sources) + loss conv; the sweeps themselves are latency-bound VALU work
            F_iter = (6.0 + (L * 32.0 * ((K + 31) // 32)) / N) * 2.0 * K * N * (L * T - L * (L - 1) / 2)
        if alg == "hals":
            comm = {"mode": "replicas", "nranks": nrep_hals, "info": "independent replicas, no data-path collective"} if ngpu > 1 else None
        elif ngpu > 1:
            comm = parse_comm(rule.comm_info(), "one-process" if form == "multi" else "one-process-per-gpu",
                              getattr(rule, "transport_fallback", None))
            try:
                comm["rccl"] = cmf.rccl_version()
            except Exception as e:  # noqa: BLE001
                comm["rccl"] = {"error": repr(e)}
            comm["HSA_ENABLE_IPC_MODE_LEGACY"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
            comm["allreduce"] = allreduce
            comm["collectives_per_step"] = coll_per_step
            comm["collectives"] = collectives
        else:
            comm = None
        out = {
            "metric": ("MU iters/sec (convolutive NMF multiplicative update; achieved HBM GB/s in `hbm`)" if alg == "mult"
                       else "HALS iters/sec (convolutive NMF, src/algs/hals.jl)"),
            "value": iters_per_s, "unit": "iter/s", "n_gpus": ngpu, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "strong" if alg == "mult" else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[{args.config - 1}]: N={N} T={T} K={K} L={L} fp32 alg=:{alg}"
                                   + (" regularised " + json.dumps(reg) if reg else "")
                                   + (f", T sharded over {ngpu} GPUs, RCCL all-reduce on W" if sharded else
                                      (f", {nrep_hals} independent replicas" if ngpu > 1 else " on 1xMI355X")),
                       "N": N, "T": T, "K": K, "L": L, "parallelism": f"t-shard{ngpu}" if alg == "mult" else f"replicas{ngpu}",
                       "launch": form, "gen_synthetic_seed": 1234, "init_rand_seed": 0,
                       "loss_first": loss0, "loss_last": losses[-1] if losses else loss0},
            "library": lib.cmf_version().decode(),
            "flops_per_iter": F_iter,
            "executed_flops_per_iter": F_iter * 6.0 / 7.0 if alg == "mult" else F_iter,
            "est_reuse": "the est of mult.jl:55 is kept for the next mult.jl:28 (same W, H): 6 of the 7 contractions "
                         "are executed, results bitwise identical; ms_per_step_no_reuse runs all 7",
            "allreduce_overlap": (bool(rule.overlap) if sharded else None),
            "allreduce_overlap_probe_ms": ({k: (1e3 * v if isinstance(v, float) else v) for k, v in probe.items()} if probe else None),
            "ms_per_step_second_pass_all_event_pairs": (1e3 * dt_allpairs / args.steps) if dt_allpairs else None,
            "ms_per_step_no_reuse": (1e3 * dt_noreuse) if dt_noreuse else None,
            # the headline both ways: `value` executes 6 of mult.jl's 7 contractions (est reuse, bitwise identical results);
            # the reference formulation recomputes est at mult.jl:28 (option reuse_est = 0: all 7 executed)
            "value_reference_formulation": (1.0 / dt_noreuse) if dt_noreuse else None,
            "ms_per_step_call_by_call": call_by_call.get("ms_per_step_call_by_call") if call_by_call else None,
            "ms_per_step_call_by_call_writeback": call_by_call.get("ms_per_step_call_by_call_writeback") if call_by_call else None,
            "call_by_call": call_by_call,
            "ms_per_step_gram": (1e3 * dt_gram) if dt_gram else None,
            "ms_per_step_gram_loss": (1e3 * dt_gram2) if dt_gram2 else None,
            "gram_note": "option gram=1: denomW/denomH through Gram matrices (exact rewriting, rounding-level differences), "
                         "loss still by conv; gram=2: loss from Gram sums too.  Not part of `value`.",
            # whole-iteration MFMA fraction on EXECUTED flops (6 contractions with est reuse).  (SURVEY.md section 8d's
            # 7-contraction figure F_iter is `flops_per_iter`; dividing it by the time counts the reused est as work done,
            # so no fraction is formed from it.)
            "whole_iteration_tflops_executed": (F_iter * (6.0 / 7.0 if alg == "mult" else 1.0)) * iters_per_s / 1e12,
            "whole_iteration_mfma_frac": (F_iter * (6.0 / 7.0 if alg == "mult" else 1.0)) * iters_per_s / (ngpu * PEAK_FP32_MFMA_TFLOPS * 1e12),
            "sustained": sustained,
            "comm": comm,
            "group_extras": group_extras,
        }

    # ---- roofline of the dominant kernel = the class with the largest share of the timed region; durations from
    # the HIP event pairs recorded inside the timed loop above (rank 0's shard) ----
    PMC_NAMES = {"transconv": "void transconv_kernel<20>", "hxt": "void hxt_kernel<5>", "hxt_num": "void hxt_kernel<5>", "hxt_den": "void hxt_kernel<5>", "conv_t": "void conv3_kernel<1>",
                 "conv_loss_store": "void conv3_kernel<3>", "conv": "void conv3_kernel<0>", "conv_loss": "void conv3_kernel<2>"}
    DESCR = {"hxt_num": "hxt_kernel<LP> (H_shift x data', mult.jl:32)", "hxt_den": "hxt_kernel<LP> (H_shift x est', mult.jl:33)",
             "transconv": "transconv_kernel<LT> (W' x data and W' x est, mult.jl:47-48)", "hxt": "hxt_kernel<LP> (H_shift x data' and H_shift x est', mult.jl:31-34)",
             "conv_t": "conv3_kernel<1> (tensor_conv, est'[n][t], mult.jl:44)", "conv_loss_store": "conv3_kernel<3> (tensor_conv + loss, mult.jl:55-57)",
             "conv": "conv3_kernel<0> (tensor_conv, mult.jl:28)", "conv_loss": "conv3_kernel<2> (tensor_conv + loss, mult.jl:55-57)"}

    def load_pmc():
        for nm in (PMC_PROFILE, PMC_FALLBACK):
            pth = os.path.join(ROOT, "profiles", nm)
            if os.path.exists(pth):
                return json.load(open(pth)), nm
        raise FileNotFoundError("no PMC summary under profiles/")

    progress["phase"] = "stand-alone kernel timings"
    if rank == 0:
        kern = {}
        timer = rule
        Tl = T // ngpu if alg == "mult" else T
        f1 = 2.0 * K * N * (L * Tl - L * (L - 1) / 2)  # one contraction on this rank's columns
        for name in ("conv", "conv_t", "conv_loss", "conv_loss_store", "hxt", "transconv"):
            kms, kfl = timer.time_kernel(name, reps=5)
            kern[name] = {"avg_ms": kms, "tflops": kfl / kms / 1e9, "frac": kfl / kms / 1e9 / PEAK_FP32_MFMA_TFLOPS}
        out["kernels_standalone"] = kern
        if inloop:
            tab = {}
            for name, (kms, n) in inloop.items():
                kfl = f1 * (2.0 if name in ("hxt", "transconv") else 1.0)
                tab[name] = {"avg_ms": kms, "launches": n, "tflops": kfl / kms / 1e9, "frac": kfl / kms / 1e9 / PEAK_FP32_MFMA_TFLOPS,
                             "share_of_step": kms * args.steps / (1e3 * dt)}
            out["kernels"] = tab
            # every class runs once per step; hxt and transconv are within 1 % of each other: the roofline block stays on the one that is
            # bracketed inside the timed steps
            dom = dom_class if (one_class and dom_class in tab) else max(tab, key=lambda k: tab[k]["avg_ms"])
            dom_second = max(tab, key=lambda k: tab[k]["avg_ms"])  # (hxt and transconv are within 1 % of each other: either may lead a pass)
            ach, avg_ms, kfl = tab[dom]["tflops"], tab[dom]["avg_ms"], f1 * (2.0 if dom in ("hxt", "transconv") else 1.0)
            src = ("HIP event pairs around every 4th launch of this kernel inside the timed region (the other rows of `kernels`: the same over a second pass of the K steps)"
                   if one_class else "HIP event pairs around each launch inside the timed region")
            for name in tab:
                tab[name]["measured_in"] = "timed steps" if (not one_class or name == dom) else "second pass of the same steps"
        else:  # HALS: see the latency-bound block below; the MFMA kernels are timed stand-alone
            dom_second = None
            dom, ach, avg_ms, kfl = "conv", kern["conv"]["tflops"], kern["conv"]["avg_ms"], f1
            src = "cmf_time_kernel: HIP events around 5 stand-alone launches"
        traffic, traffic_src = None, None
        try:  # HBM bytes per launch from the committed rocprofv3 PMC passes (same workload, same kernel)
            if args.config in (2, 4) and ngpu == 1 and not args.T:
                pm, pm_name = load_pmc()
                traffic = pm[PMC_NAMES[dom]]["hbm_bytes_corrected"]
                traffic_src = f"profiled earlier, not in this run: profiles/{pm_name} -- (2*FETCH_SIZE + WRITE_SIZE)*1024, separate rocprofv3 --pmc passes of this workload"
        except Exception:
            pass
        out["roofline"] = {"bound": "mfma", "kernel": DESCR[dom] + ("" if ngpu == 1 or alg != "mult" else f" on rank 0's shard of {Tl} columns"),
                           "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": ach / PEAK_FP32_MFMA_TFLOPS,
                           # executed flops of the WHOLE iteration (6 contractions with est reuse) / time / peak
                           "frac_whole_iteration": out.get("whole_iteration_mfma_frac"),
                           "frac_whole_iteration_reference_formulation": ((F_iter / dt_noreuse) / (ngpu * PEAK_FP32_MFMA_TFLOPS * 1e12)
                                                                          if (alg == "mult" and dt_noreuse) else None),
                           "traffic": traffic, "traffic_source": traffic_src,
                           "algorithmic_flops_per_launch": kfl, "avg_launch_ms": avg_ms, "timing": src,
                           # how the bracketed class was chosen, and whether the all-classes pass agrees
                           "dominant_by_warmup": (dom_class if (inloop and one_class) else None),
                           "dominant_by_second_pass": (dom_second if inloop else None)}

    if rank == 0 and alg == "hals":
        out["roofline_mfma_kernel"] = out["roofline"]
        out["roofline"] = hals_roofline(T, K, L, hals_spans, 1e3 * dt / args.steps if dt else None)
        out["hals_pipeline_reruns"] = rule.counter("hals_pipeline_reruns")

    if rank == 0 and alg == "mult":
        # BASELINE.json's metric also asks for the achieved HBM rate.  Algorithmic bytes per iteration
        # (est never round-tripped): B_iter = 12*N*T + 48*K*N*L + 40*K*T; measured bytes
        # per iteration = sum of the PMC-corrected traffic of the four big launches (profiles/*_pmc_summary.json).
        B_iter = 12.0 * N * T + 48.0 * K * N * L + 40.0 * K * T
        hbm = {"algorithmic_bytes_per_iter": B_iter, "achieved_algorithmic_GBps": B_iter * out["value"] / 1e9,
               "peak_GBps": 8000.0, "note": "the path is fp32-MFMA-bound (intensity ~700 flop/B), not HBM-bound"}
        try:
            if args.config in (2, 4) and ngpu == 1 and not args.T:
                pm, pm_name = load_pmc()
                meas = sum(pm[k]["hbm_bytes_corrected"] for k in ("void hxt_kernel<5>", "void conv3_kernel<1>",
                                                                 "void transconv_kernel<20>", "void conv3_kernel<3>"))
                hbm["profiled_bytes_per_iter"] = meas
                hbm["profiled_GBps"] = meas * out["value"] / 1e9
                hbm["profiled_source"] = f"profiles/{pm_name} (PMC passes run separately, not in this run)"
        except Exception:
            pass
        out["hbm"] = hbm
    if rank == 0 and extras and args.config == 2 and not args.T:
        out["other_configs"] = other_configs(cmf, rule, data, W0, H0, N, T, K, L, not args.no_config3, local_rank)
        for k_, rec_ in out["other_configs"].get("shards", {}).items():
            if isinstance(rec_, dict) and rec_.get("ms_per_step"):
                rec_["speedup_before_communication"] = out["ms_per_step"] / rec_["ms_per_step"]
                if rec_.get("ms_per_step_gram") and dt_gram:
                    rec_["speedup_before_communication_gram"] = 1e3 * dt_gram / rec_["ms_per_step_gram"]
    if rank == 0:
        if args.cpu_seconds > 0 and ngpu == 1:
            try:
                out["cpu_baseline"] = (cpu_baseline_hals if alg == "hals" else cpu_baseline)(data, W0, H0, args.cpu_seconds)
            except Exception as e:  # the baseline is a reported extra; never lose the GPU line for it
                out["cpu_baseline"] = {"value": None, "unit": "iter/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {e!r}"}
        out["bench_wall_s"] = round(time.time() - T_PROCESS_START, 1)  # the whole run of this process: extras and the CPU baseline included
        print(json.dumps(out), flush=True)
    for r in (replicas or [rule]):
        r.close()
    if form == "ranks":
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
