"""Per-iteration kernel timeline from a rocprofv3 --kernel-trace CSV (start/end timestamps -> busy time and gaps).

usage: python tools/trace_gaps.py <dir with *_kernel_trace.csv> [marker-kernel-substring]
Prints, for the last few iterations (delimited by the marker kernel, default `h_update_kernel`), each launch's
duration and the idle gap before it.
"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "w_update_kernel"
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    idx = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(idx) < 4:
        print("marker not found often enough", len(idx))
        return
    # one period = marker to marker, taken in the middle of the run
    mid = len(idx) // 2
    a, b = idx[mid], idx[mid + 1]
    prev_end = rows[a - 1][1]
    t0 = rows[a][0]
    busy = 0
    for s, e, n in rows[a:b]:
        print(f"{(s - t0) / 1e3:9.1f} us  gap {max(0, s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:8.1f}  {n[:70]}")
        busy += e - s
        prev_end = max(prev_end, e)
    per = rows[b][0] - rows[a][0]
    print(f"period {per / 1e3:.1f} us, busy {busy / 1e3:.1f} us, idle {(per - busy) / 1e3:.1f} us")
    # averages over all full periods in the timed part
    pers = [rows[idx[i + 1]][0] - rows[idx[i]][0] for i in range(len(idx) - 1)]
    pers.sort()
    print(f"median period {pers[len(pers) // 2] / 1e3:.1f} us over {len(pers)} periods")


if __name__ == "__main__":
    main()
