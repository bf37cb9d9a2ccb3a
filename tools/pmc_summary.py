"""Per-kernel means of rocprofv3 --pmc passes (one directory per pass) -> JSON.

usage: python tools/pmc_summary.py out.json <dir_fetch> <dir_write> <dir_sq>
Each directory holds the *counter_collection.csv of one `rocprofv3 --kernel-trace --pmc ...` run of bench.py.
FETCH_SIZE / WRITE_SIZE are in KiB; hbm_bytes_corrected = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE reads
half of a wide coalesced stream, MI355X_MICROARCH.md HBM section).  effective_clock_GHz = GRBM_GUI_ACTIVE/8/duration,
mfma_pipe_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE/8).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def read(d):
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        with open(f) as fh:
            for r in csv.DictReader(fh):
                name = r["Kernel_Name"].split("(")[0]
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
                key = r.get("Dispatch_Id")
                if key not in seen and "Start_Timestamp" in r and r["Start_Timestamp"]:
                    seen.add(key)
                    dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return acc, dur


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    res = defaultdict(dict)
    for d in dirs:
        acc, dur = read(d)
        for k, cs in acc.items():
            for c, v in cs.items():
                res[k][c + ("_KiB_mean" if c in ("FETCH_SIZE", "WRITE_SIZE") else "")] = sum(v) / len(v)
            if "GRBM_GUI_ACTIVE" in cs and dur.get(k):
                res[k]["avg_duration_us_under_pmc"] = sum(dur[k]) / len(dur[k])
    for k, v in res.items():
        if "FETCH_SIZE_KiB_mean" in v and "WRITE_SIZE_KiB_mean" in v:
            v["hbm_bytes_corrected"] = (2 * v["FETCH_SIZE_KiB_mean"] + v["WRITE_SIZE_KiB_mean"]) * 1024
        if "GRBM_GUI_ACTIVE" in v and "avg_duration_us_under_pmc" in v:
            v["effective_clock_GHz"] = v["GRBM_GUI_ACTIVE"] / 8 / v["avg_duration_us_under_pmc"] / 1e3
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
            v["mfma_pipe_busy_frac"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (v["GRBM_GUI_ACTIVE"] / 8)
    keep = {k: v for k, v in res.items() if any(s in k for s in ("conv", "hxt", "transconv", "g_gemm", "fold_small", "slab_sum"))}
    json.dump(keep, open(out, "w"), indent=1, sort_keys=True)
    for k, v in sorted(keep.items()):
        print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in
                  ("hbm_bytes_corrected", "mfma_pipe_busy_frac", "effective_clock_GHz", "avg_duration_us_under_pmc")})


if __name__ == "__main__":
    main()
