"""Stand-in for a bench.py worker rank: follows the marker protocol of bench.py's supervisor (ready -> valid -> done, fail, exit codes)
without a GPU.  FAKE_MODE says how attempt 0 goes wrong: "ok" (it does not), "hang" (rank 1 never finishes its validation), "fail"
(agreed validation failure: every rank leaves with EXIT_FALLBACK), "die" (rank 1 exits with code 41), "late" (rank 1 hangs in the timed
region).  Every later attempt succeeds; rank 0 prints one JSON line that says which attempt it is and why the earlier ones ended."""
import json
import os
import sys
import time

d, k, r = os.environ["RF_BENCH_RUN_DIR"], int(os.environ["RF_BENCH_ATTEMPT"]), int(os.environ["RANK"])
mode = os.environ.get("FAKE_MODE", "ok") if k == 0 else "ok"


def mark(name, text=""):
    with open(os.path.join(d, f"a{k}.r{r}.{name}"), "w") as fh:
        fh.write(text)


def barrier(kind):  # (like a collective: returns when every rank got there, never if one does not)
    import glob

    while len(glob.glob(os.path.join(d, f"a{k}.r*.{kind}"))) < int(os.environ["WORLD_SIZE"]):
        time.sleep(0.05)


if os.environ.get("FAKE_PID_DIR"):
    with open(os.path.join(os.environ["FAKE_PID_DIR"], f"pid.{k}.{r}"), "w") as fh:
        fh.write(str(os.getpid()))
assert os.environ["RF_BENCH_WORKER"] == "1" and os.environ["TORCHELASTIC_USE_AGENT_STORE"] == "False" and int(os.environ["MASTER_PORT"]) > 0
time.sleep(0.2 * r)
mark("ready")
if mode == "hang" and r == 1:
    time.sleep(3600)
if mode == "die" and r == 1:
    os._exit(41)
if mode == "fail":
    if r == 1:
        mark("fail", "rank 1: RuntimeError: replicas diverged (fake)")
    time.sleep(0.3)
    os._exit(75)
time.sleep(0.3)
mark("valid")
barrier("valid")
if mode == "late" and r == 1:
    time.sleep(3600)
time.sleep(0.3)
mark("done")
barrier("done")
if r == 0:
    print(json.dumps({"attempt": k, "argv": sys.argv[1:], "reason": os.environ.get("RF_BENCH_FALLBACK_REASON"), "halves": os.environ.get("RF_OWNER_HALVES"),
                      "fast": os.environ.get("RF_DIST_FAST")}), flush=True)
