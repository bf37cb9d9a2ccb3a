"""Timeline of the persistent HALS H pipeline from its s_memtime stamps (CMF_HALS_STAMPS, 100 MHz ticks):
python tools/hals_stamps.py   (config-5 sizes; runs two iterations and analyses the second H phase)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cmf_jl_amd as cmf

N, T, K, L = 2000, 50000, 32, 20
path = "/tmp/hals_stamps.bin"
data = cmf.gen_synthetic(N=N, T=T, seed=1234)
W0, H0 = cmf.init_rand(data, L=L, K=K, seed=0)
rule = cmf.HALSUpdate(data, W0, H0)
rule.update_motifs(); rule.update_feature_maps()
rule.update_motifs()
os.environ["CMF_HALS_STAMPS"] = path
rule.update_feature_maps()
os.environ.pop("CMF_HALS_STAMPS")
raw = np.fromfile(path, dtype=np.uint64)
Kf, nblk = int(raw[0]), int(raw[1])
st = raw[2:].astype(np.float64) * 0.01  # us
sw = st[: Kf * nblk].reshape(Kf, nblk)
pu = st[Kf * nblk:].reshape(Kf, nblk, 4)
nf = int((sw[0] > 0).sum()) - 2  # fast-path blocks (the last two are the generic tail)
tail = sw[:, nf:nf + 2] - sw[:, nf - 1:nf]
print("tail: end of the two generic blocks after the last fast block (us), rows 0, 1, 16, 31:", np.round(tail[[0, 1, 16, 31]], 1).tolist())
print("row end relative to the row above (us):", np.array2string(np.diff(sw[:, nf + 1]), precision=1, max_line_width=220))
sw = sw[:, :nf]
t0 = sw[sw > 0].min()
print(f"rows {Kf}, blocks {nblk} (stamped {nf}); span of the sweeps {sw.max() - t0:.1f} us")
per = np.diff(sw, axis=1)
print("block period per row (median us):", np.array2string(np.median(per, axis=1), precision=2, max_line_width=220))
lag = sw[1:, :] - sw[:-1, :]
print("lag behind the row above at block 100 / 400 / 700 (us):")
for c in (100, 400, 700):
    print(f"  block {c}:", np.array2string(lag[:, c], precision=1, max_line_width=220))
print("row start (first block end) relative to row 0 (us):", np.array2string(sw[:, 0] - sw[0, 0], precision=1, max_line_width=220))
# pullers: block b of row k waits for row k-1 having published block b+1
k = 16
b = np.arange(50, nf - 8)
wait_done = pu[k, b, 0]
src_end = sw[k - 1, b + 1]
print(f"row {k} pullers: wait satisfied - end of block b+1 of row {k-1}: median {np.median(wait_done - src_end):.2f} us (publish at mid-block + flag + poll)")
print(f"  staging {np.median(pu[k, b, 1] - pu[k, b, 0]):.2f}  compute {np.median(pu[k, b, 2] - pu[k, b, 1]):.2f}  reduce+store+flag {np.median(pu[k, b, 3] - pu[k, b, 2]):.2f} us")
print(f"  flag of block b raised -> end of block b-2 of row {k} (its mid-block check): median {np.median(sw[k, b - 2] - pu[k, b, 3]):.2f} us of slack")
for kk in (1, 8, 31):
    print(f"  row {kk}: end-to-end per block {np.median(pu[kk, b, 3] - sw[kk - 1, b + 1]):.2f} us, compute {np.median(pu[kk, b, 2] - pu[kk, b, 1]):.2f}")
tot = sw[:, -1] - sw[:, 0]
print("row total first->last stamped block (us):", np.array2string(tot, precision=0, max_line_width=220))
big = (per > 2.5)
print("periods > 2.5 us per row:", big.sum(axis=1).tolist())
print("time in those periods per row (us):", np.array2string((per * big).sum(axis=1), precision=0, max_line_width=220))
r = 20
idx = np.nonzero(big[r])[0][:12]
print(f"row {r}: long periods at blocks", idx.tolist(), "lengths", np.round(per[r, idx], 1).tolist())
print(f"row {r-1}: periods at the same blocks", np.round(per[r - 1, idx], 1).tolist())
for r in (2, 3, 4, 12):
    idx = np.nonzero(big[r])[0]
    print(f"row {r}: long periods at blocks", idx.tolist(), "lengths", np.round(per[r, idx], 1).tolist())
r = 2
for c in np.nonzero(big[r])[0][:3]:
    c = int(c)
    print(f"row {r} block {c+1} ended {per[r, c]:.1f} us after block {c}; it needed the pull of block {c+3} (checked at its middle):")
    for bb in (c + 1, c + 2, c + 3, c + 4):
        if bb + 1 >= sw.shape[1]:
            break
        print(f"   pull of block {bb}: src row {r-1} block {bb+1} ended {sw[r-1, bb+1] - sw[r, c]:+.1f}, wait done {pu[r, bb, 0] - sw[r, c]:+.1f}, staged {pu[r, bb, 1] - sw[r, c]:+.1f}, computed {pu[r, bb, 2] - sw[r, c]:+.1f}, flag {pu[r, bb, 3] - sw[r, c]:+.1f}  (us, relative to the end of block {c} of row {r})")
# the end of a row in detail: relative to the end of the row above (its last stamped block, the edge tail)
raw_sw = (raw[2:].astype(np.float64) * 0.01)[: Kf * nblk].reshape(Kf, nblk)
for r in (10, 20):
    F = raw_sw[r - 1, nblk - 1]
    print(f"row {r}: ends of its blocks {nblk-6}..{nblk-1} relative to the end of row {r-1} (us):", np.round(raw_sw[r, nblk - 6:] - F, 1).tolist())
    for bb in range(nblk - 4, nblk):
        print(f"   pull of block {bb}: wait done {pu[r, bb, 0] - F:+.1f}, staged {pu[r, bb, 1] - F:+.1f}, computed {pu[r, bb, 2] - F:+.1f}, flag {pu[r, bb, 3] - F:+.1f}")
