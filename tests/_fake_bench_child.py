"""Stands in for bench.py's measurement child in the CPU tests of the supervisor (CMF_BENCH_FAKE_CHILD).

FAKE_PLAN = comma-separated outcomes per attempt index (CMF_BENCH_ATTEMPT): ok | fail | hang | noline | diag;
a trailing "@r" restricts an outcome to rank r ("fail@1": only rank 1 fails, the others behave as "ok").
FAKE_LOG (optional): a file every child appends "attempt rank form env-extras" to."""
import json
import os
import sys
import time

a = int(os.environ.get("CMF_BENCH_ATTEMPT", "0"))
rank = int(os.environ.get("RANK", "0"))
plan = os.environ.get("FAKE_PLAN", "ok").split(",")
what = plan[min(a, len(plan) - 1)]
if "@" in what:
    what, only = what.split("@")
    if int(only) != rank:
        what = "ok"
form = sys.argv[sys.argv.index("--child") + 1]
if os.environ.get("FAKE_LOG"):
    with open(os.environ["FAKE_LOG"], "a") as f:
        f.write(f"{a} {rank} {form} threads={os.environ.get('CMF_ENQUEUE_THREADS', '-')} transport={os.environ.get('CMF_TRANSPORT', '-')} "
                f"backend={os.environ.get('CMF_DIST_BACKEND', '-')} world={os.environ.get('WORLD_SIZE', '-')} port={os.environ.get('MASTER_PORT', '-')}\n")
if what == "hang":
    time.sleep(600)
if what == "fail":
    print("fake child: boom", file=sys.stderr)
    sys.exit(7)
if what == "diag":  # the measurement's own failure record, then a non-zero exit
    print(json.dumps({"value": None, "failed_phase": "warm-up steps", "error": "CMFError(5, 'x')", "cmf_last_error": "ncclAllReduce failed", "comm": {"transport": "rccl"}}))
    sys.exit(3)
if what == "ok" and rank != 0 and form == "ranks":
    time.sleep(0.3)  # rank 0 prints, the others just finish
    sys.exit(0)
if what == "noline":
    sys.exit(0)
print("some other output")
print(json.dumps({"metric": "MU iters/sec", "value": 123.0 + a, "unit": "iter/s", "n_gpus": int(sys.argv[sys.argv.index("--gpus") + 1]), "comm": {"mode": form}}))
