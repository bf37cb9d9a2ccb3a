"""Per-CU timeline from the raw stamps of tools/conv_stamps.hip (s_memtime is per XCD, so everything is per XCD/CU)."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
hw = a[:, 4].astype(np.int64) & 0xFFFFFFFF
xcc = (a[:, 4].astype(np.int64) >> 32) & 0xF
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7
sh = (hw >> 12) & 1
key = xcc * 1000 + se * 100 + sh * 50 + cu
print("workgroups", len(a), "distinct CUs", len(np.unique(key)), "xcc ids", np.unique(xcc))
spans, ntiles, busy3 = [], [], []
for x in np.unique(xcc):
    m = xcc == x
    t0 = a[m, 0].min()
    t1 = a[m, 3].max()
    print(f"xcc {x}: {m.sum()} workgroups, span {(t1 - t0) / 1e3:.1f} kcyc; CUs {len(np.unique(key[m]))}")
    for k in np.unique(key[m]):
        mk = key == k
        s = a[mk, 0].astype(np.int64) - int(t0)
        e = a[mk, 3].astype(np.int64) - int(t0)
        spans.append((e.max() - s.min()))
        ntiles.append(mk.sum())
        # time with fewer than 3 resident workgroups
        ev = sorted([(t, 1) for t in s] + [(t, -1) for t in e])
        cur, last, less = 0, s.min(), 0
        for t, d in ev:
            if cur < 3:
                less += t - last
            last = t
            cur += d
        busy3.append(less)
    # last start / first end on this XCD
    print(f"   first start {0}, last start {(a[m, 0].max() - t0) / 1e3:.1f}, first CU done / last CU done: "
          f"{min((a[(key == k), 3].max() - t0) for k in np.unique(key[m])) / 1e3:.1f} / {(t1 - t0) / 1e3:.1f} kcyc")
ntiles = np.array(ntiles); spans = np.array(spans); busy3 = np.array(busy3)
print("tiles per CU: min %d max %d mean %.2f" % (ntiles.min(), ntiles.max(), ntiles.mean()))
print("per-CU span kcyc: min %.1f max %.1f mean %.1f" % (spans.min() / 1e3, spans.max() / 1e3, spans.mean() / 1e3))
print("per-CU time with <3 resident workgroups, kcyc: mean %.1f max %.1f" % (busy3.mean() / 1e3, busy3.max() / 1e3))
d = a[:, 3].astype(np.int64) - a[:, 0].astype(np.int64)
print("workgroup lifetime kcyc: mean %.1f p5 %.1f p95 %.1f" % (d.mean() / 1e3, np.percentile(d, 5) / 1e3, np.percentile(d, 95) / 1e3))

# concurrency of main-loop phases per CU: time with c workgroups between stamp 1 (main loop starts) and stamp 2 (ends)
Tc = np.zeros(5)
for k in np.unique(key):
    mk = key == k
    s = a[mk, 1].astype(np.int64)
    e = a[mk, 2].astype(np.int64)
    lo = a[mk, 0].astype(np.int64).min()
    hi = a[mk, 3].astype(np.int64).max()
    ev = sorted([(t, 1) for t in s] + [(t, -1) for t in e])
    cur, last = 0, lo
    for t, d in ev:
        Tc[min(cur, 4)] += t - last
        last = t
        cur += d
    Tc[0] += hi - last
ncu = len(np.unique(key))
print("per-CU mean time (kcyc) with c workgroups inside their main loop: " + "  ".join(f"c={c}: {Tc[c] / ncu / 1e3:.1f}" for c in range(5)))
print("MFMA cycles needed per SIMD per CU: %.1f kcyc (tiles/CU x 81.92k)" % (len(a) / ncu * 81.92))

# per-workgroup regression: 81.92k MFMA cycles = sum_c time_in_state_c * (u_c / c), c = workgroups concurrently in main loop
rows = []
for k in np.unique(key):
    idx = np.where(key == k)[0]
    s = a[idx, 1].astype(np.int64)
    e = a[idx, 2].astype(np.int64)
    ev = sorted([(t, 1, j) for j, t in enumerate(s)] + [(t, -1, j) for j, t in enumerate(e)])
    active = set()
    acc = np.zeros((len(idx), 4))
    last = ev[0][0]
    for t, d, j in ev:
        c = len(active)
        if c:
            for w in active:
                acc[w, min(c, 3)] += t - last
        last = t
        if d > 0:
            active.add(j)
        else:
            active.discard(j)
    rows.append(acc[:, 1:4])
A = np.vstack(rows)
x, res, rk, sv = np.linalg.lstsq(A, np.full(len(A), 81920.0), rcond=None)
print("fitted MFMA pipe utilisation with c workgroups in their main loop: c=1 %.3f  c=2 %.3f  c=3 %.3f" % (x[0], 2 * x[1], 3 * x[2]))
pred = A @ x
print("fit residual rms %.1f kcyc of 81.92" % (np.sqrt(np.mean((pred - 81920.0) ** 2)) / 1e3))
