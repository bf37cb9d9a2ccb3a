"""bench.py's first-contact supervisor for --gpus N > 1 (CPU; the measurement child is replaced by tests/_fake_bench_child.py):
the ladder of forms, one fresh child process per attempt, bounded attempts, the launcher form's agreement between the
supervising ranks through a TCP store, and the JSON line that is printed when everything failed."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "_fake_bench_child.py")


def run_bench(args, env_extra, timeout=120):
    env = dict(os.environ, CMF_BENCH_FAKE_CHILD=FAKE, CMF_TEST_HOOKS="1", **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_USE_AGENT_STORE"):
        if k not in env_extra:
            env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def the_line(proc):
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (proc.stdout, proc.stderr)  # ONE JSON line, whatever happened
    return json.loads(lines[0])


def test_plain_launch_first_form_succeeds(tmp_path):
    log = tmp_path / "log"
    p = run_bench(["--gpus", "8"], {"FAKE_PLAN": "ok", "FAKE_LOG": str(log)})
    rec = the_line(p)
    assert p.returncode == 0 and rec["value"] == 123.0 and rec["n_gpus"] == 8
    assert [a["ok"] for a in rec["attempts"]] == [True] and "enqueue thread per GPU" in rec["attempts"][0]["form"]
    assert log.read_text().split()[:4] == ["0", "0", "multi", "threads=-"]  # one child, the one-process form, workers on


def test_plain_launch_walks_down_the_ladder(tmp_path):
    """attempt 0 crashes, attempt 1 (calling thread enqueues) prints only its failure record, attempt 2 (bench.py as its
    own launcher: 4 ranks) has one rank failing, attempt 3 (collectives through torch.distributed) succeeds."""
    log = tmp_path / "log"
    p = run_bench(["--gpus", "4"], {"FAKE_PLAN": "fail,diag,fail@2,ok", "FAKE_LOG": str(log)})
    rec = the_line(p)
    assert p.returncode == 0 and rec["value"] == 126.0
    at = rec["attempts"]
    assert [a["ok"] for a in at] == [False, False, False, True]
    assert "exit code 7" in at[0]["ended"] and at[0]["children"][0]["stderr_tail"] == ["fake child: boom"]
    assert at[1]["env"] == {"CMF_ENQUEUE_THREADS": "0"} and at[1]["child_line"]["failed_phase"] == "warm-up steps"
    assert at[1]["child_line"]["cmf_last_error"] == "ncclAllReduce failed"
    assert "#2 exit code 7" in at[2]["ended"]
    assert at[3]["env"] == {"CMF_TRANSPORT": "host"}
    runs = [ln.split() for ln in log.read_text().splitlines()]
    assert sum(1 for r in runs if r[0] == "2") == 4 and {r[1] for r in runs if r[0] == "3"} == {"0", "1", "2", "3"}
    assert all(r[2] == "ranks" and r[6] == "world=4" for r in runs if r[0] in ("2", "3"))
    assert len({r[7] for r in runs if r[0] == "2"}) == 1  # one rendezvous port for the ranks of an attempt


def test_a_hung_attempt_is_ended_at_its_limit():
    p = run_bench(["--gpus", "2", "--attempt-timeout", "2"], {"FAKE_PLAN": "hang,ok"})
    rec = the_line(p)
    assert p.returncode == 0 and rec["value"] == 124.0
    assert "no result within 2 s" in rec["attempts"][0]["ended"] and rec["attempts"][1]["ok"]


def test_everything_failed_still_prints_one_line():
    p = run_bench(["--gpus", "2", "--attempt-timeout", "5"], {"FAKE_PLAN": "fail,noline,diag,fail"})
    rec = the_line(p)
    assert p.returncode == 3 and rec["value"] is None and rec["n_gpus"] == 2 and rec["failed_phase"]
    assert len(rec["attempts"]) == 4 and not any(a["ok"] for a in rec["attempts"])
    assert "no JSON line" in rec["attempts"][1]["ended"]
    assert rec["attempts"][2]["child_line"]["comm"] == {"transport": "rccl"}


@pytest.mark.parametrize("plan,want_attempts", [("ok", 1), ("fail@1,ok", 2), ("hang@0,fail@1,ok", 3), ("noline@0,ok", 2)])
def test_launcher_form_ranks_agree_through_the_store(tmp_path, plan, want_attempts):
    """WORLD_SIZE == --gpus: every started process supervises its own child; a failure on ANY rank sends ALL of them to
    the next form (the healthy ranks' children are ended early through the store), and only rank 0 prints."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    log = tmp_path / "log"
    procs = []
    for r in range(2):
        env = dict(os.environ, CMF_BENCH_FAKE_CHILD=FAKE, CMF_TEST_HOOKS="1", FAKE_PLAN=plan, FAKE_LOG=str(log), RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--attempt-timeout", "4"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    assert not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]  # rank 1 prints nothing
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["value"] == 123.0 + want_attempts - 1 and len(rec["attempts"]) == want_attempts
    assert [a["ok"] for a in rec["attempts"]] == [False] * (want_attempts - 1) + [True]
    runs = [ln.split() for ln in log.read_text().splitlines()]
    for a in range(want_attempts):  # both ranks ran every attempt, with the attempt's own rendezvous port (not the launcher's)
        mine = [r for r in runs if r[0] == str(a)]
        assert {r[1] for r in mine} == {"0", "1"} and len({r[7] for r in mine}) == 1 and mine[0][7] != f"port={port}"
    if want_attempts >= 2:
        assert any(r[0] == "1" and r[4] == "transport=host" for r in runs)
    if want_attempts == 3:
        assert any(r[0] == "2" and r[5] == "backend=gloo" for r in runs)


def test_supervisor_told_to_stop_takes_its_child_along(tmp_path):
    """SIGTERM to the supervising process (a launcher giving up on the rank): the measurement child, which runs in a session of
    its own, is ended too instead of being left behind on the GPU."""
    import signal
    import time

    log = tmp_path / "log"
    env = dict(os.environ, CMF_BENCH_FAKE_CHILD=FAKE, CMF_TEST_HOOKS="1", FAKE_PLAN="hang", FAKE_LOG=str(log))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_USE_AGENT_STORE"):
        env.pop(k, None)
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    t_end = time.time() + 60
    while not log.exists() and time.time() < t_end:  # the child has started (it logs, then hangs)
        time.sleep(0.1)
    assert log.exists()
    kids = subprocess.run(["pgrep", "-P", str(p.pid)], capture_output=True, text=True).stdout.split()
    assert kids
    p.send_signal(signal.SIGTERM)
    p.wait(timeout=30)
    assert p.returncode == 128 + signal.SIGTERM
    time.sleep(0.5)
    for k in kids:
        assert not os.path.exists(f"/proc/{k}") or open(f"/proc/{k}/stat").read().split()[2] == "Z", k


def test_ladder_progress_goes_to_stderr_as_it_happens(tmp_path):
    """One flushed stderr line when a rung starts and one when it ends: a driver that ends the run at
    its own time limit still finds which form was running, and why the earlier ones ended, in the tail of stderr."""
    p = run_bench(["--gpus", "2", "--attempt-timeout", "5"], {"FAKE_PLAN": "fail,ok"})
    rec = the_line(p)
    assert p.returncode == 0 and [a["ok"] for a in rec["attempts"]] == [False, True]
    sup = [ln for ln in p.stderr.splitlines() if ln.startswith("bench.py supervisor rank 0")]
    assert len(sup) == 4, p.stderr
    assert "attempt 1/4: one process, an enqueue thread per GPU, RCCL (limit 5 s)" in sup[0]
    assert "attempt 1: FAILED: child process(es) failed: #0 exit code 7" in sup[1] and "fake child: boom" in sup[1]
    assert "attempt 2/4: one process, the calling thread enqueues every GPU" in sup[2] and "attempt 2: ok" in sup[3]
    assert all(a["child"] == "_fake_bench_child.py" for a in rec["attempts"])  # the record says a stand-in ran


def test_the_stand_in_child_needs_test_hooks(monkeypatch):
    """CMF_BENCH_FAKE_CHILD alone must not replace the measurement (the library's own test knobs are gated the
    same way)."""
    sys.path.insert(0, ROOT)
    import bench

    args = bench.parse_args(["--gpus", "2"])
    monkeypatch.setenv("CMF_BENCH_FAKE_CHILD", FAKE)
    monkeypatch.delenv("CMF_TEST_HOOKS", raising=False)
    assert bench.child_command(args, "multi")[1] == os.path.join(ROOT, "bench.py")
    monkeypatch.setenv("CMF_TEST_HOOKS", "1")
    assert bench.child_command(args, "multi")[1] == FAKE
