"""bench.py's multi-GPU supervisor (top of bench.py) on the CPU: N supervisor processes, one per rank, each running a stand-in worker
(tests/fake_bench_worker.py) through the marker protocol -- the undisturbed run, a rank that hangs in its validation, an agreed validation
failure, a worker that dies, a hang in the timed region, an explicit --exchange dense.  The real thing (two ranks of the real benchmark
on one GPU over gloo, with injected hangs / failures) is tests/test_hip_data_parallel.py::test_bench_two_ranks_*."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(mode, tmp_path, world=2, argv=(), validate_s="2", run_s="3"):
    env = dict(os.environ, WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="29999", RF_BENCH_RUN_DIR=str(tmp_path), FAKE_MODE=mode,
               RF_BENCH_WORKER_CMD=f"{sys.executable} {os.path.join(ROOT, 'tests', 'fake_bench_worker.py')}", RF_BENCH_VALIDATE_TIMEOUT_S=validate_s,
               RF_BENCH_RUN_TIMEOUT_S=run_s, RF_BENCH_READY_TIMEOUT_S="20")
    env.pop("RF_BENCH_WORKER", None)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), *argv], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=120) for p in procs]
    lines = [json.loads(ln) for o, _ in outs for ln in o.splitlines() if ln.startswith("{")]
    return [p.returncode for p in procs], lines, "\n".join(e for _, e in outs)


def test_undisturbed_run_is_attempt_zero(tmp_path):
    rcs, lines, err = _run("ok", tmp_path)
    assert rcs == [0, 0] and len(lines) == 1, (rcs, lines, err)
    assert lines[0]["attempt"] == 0 and lines[0]["reason"] is None and lines[0]["halves"] is None


@pytest.mark.parametrize("mode,needle", [("hang", "a hang"), ("fail", "replicas diverged (fake)"), ("die", "exited with code 41"), ("late", "timed region not finished")])
def test_a_broken_first_attempt_ends_in_a_labelled_line(tmp_path, mode, needle):
    rcs, lines, err = _run(mode, tmp_path, world=3)
    assert rcs == [0, 0, 0] and len(lines) == 1, (rcs, lines, err)
    line = lines[0]
    # the conservative owner-computes configuration, in fresh processes, with the reason of the abandoned attempt
    assert line["attempt"] == 1 and line["halves"] == "1" and line["fast"] == "0" and "--exchange" not in line["argv"]
    assert "attempt 0" in line["reason"] and needle in line["reason"], line["reason"]


def test_explicit_dense_exchange_is_a_single_attempt(tmp_path):
    rcs, lines, err = _run("ok", tmp_path, argv=("--exchange", "dense"))
    assert rcs == [0, 0] and len(lines) == 1 and lines[0]["attempt"] == 2, (rcs, lines, err)
    assert lines[0]["argv"].count("--exchange") == 2  # (the user's and the attempt's own: argparse keeps the last, both say dense)


def test_a_stopped_supervisor_takes_its_worker_with_it(tmp_path):
    """SIGTERM to the supervisors (a launcher's timeout): the workers are killed, nothing is left running."""
    import signal
    import time

    env = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29998", RF_BENCH_RUN_DIR=str(tmp_path), FAKE_MODE="hang",
               RF_BENCH_WORKER_CMD=f"{sys.executable} {os.path.join(ROOT, 'tests', 'fake_bench_worker.py')}", RF_BENCH_VALIDATE_TIMEOUT_S="60",
               RF_BENCH_RUN_TIMEOUT_S="60", RF_BENCH_READY_TIMEOUT_S="60", FAKE_PID_DIR=str(tmp_path))
    env.pop("RF_BENCH_WORKER", None)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(2)]
    deadline = time.time() + 30
    while time.time() < deadline and len(list(tmp_path.glob("a0.r*.ready"))) < 2:
        time.sleep(0.1)
    pids = [int(f.read_text()) for f in tmp_path.glob("pid.*")]
    assert len(pids) == 2
    for p in procs:
        p.send_signal(signal.SIGTERM)
    for p in procs:
        assert p.wait(timeout=20) == 128 + signal.SIGTERM
    time.sleep(0.5)
    for pid in pids:  # gone, or a zombie waiting for the container's init to reap it -- not running
        try:
            state = open(f"/proc/{pid}/stat").read().rsplit(")", 1)[1].split()[0]
        except FileNotFoundError:
            state = "gone"
        assert state in ("gone", "Z"), (pid, state)
