#!/usr/bin/env python
"""Headline benchmark of the MI355X ReLU-Fields render path.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json: "ray-samples/sec (fwd+bwd) ... 800x800 @ 128^3 grid", configs[2]/[3]):
one STEP = one training iteration of the posed-image trainer on a 128^3 SH-degree-2 ReLU field -- a random 16384-ray batch
out of 8 synthetic 800x800 images, specular render fwd, diffuse render fwd, L1 + L1, backward of both, [gradient exchange
over RCCL when N > 1], Adam -- with 256 stratified (jittered) samples per ray.  ``value`` = nominal ray-samples/s over the
whole job = N * 2 renders * 16384 rays * 256 samples * K / time, inputs resident in HBM, nothing skipped.
Weak scaling: every rank draws its own 16384-ray batch.

Also in the same JSON line:
  roofline      the dominant kernel of the step: HIP events recorded by the library INSIDE the timed region (rf_train_step's
                timing_events, on the launch stream); ``frac`` = HBM bytes from the PMC counters (profiles/pmc_traffic.json,
                rocprofv3 --pmc passes of this command) / this run's kernel time / 8 TB/s; ``frac_processed`` = the algorithmic
                bytes of SURVEY 8d on the units the launch REALLY processes (records emitted, counted on the device) -- never the
                nominal in-AABB count a kernel legitimately skips most of;
  fwd_render    configs[1], full 800x800 frame (one launch), on the init field and on a semi-transparent field where every
                ray traverses the whole volume;
  highres_render configs[4]; strict_dropin = the torch.autograd.Function path a reference user gets;
  cpu_baseline  the oracle (checker only) timed on the host cores.
"""
import argparse
import json
import os
import sys
import time


def host_cpu_quota() -> int:
    """CPU cores this process may really use: the cgroup's CFS quota (cpu.max = "quota period") when there is one, else
    os.cpu_count().  The GPU boxes show 256 logical cores under a 16-core quota: OpenMP pools sized by os.cpu_count() (torch's
    default: 128 threads) burn the quota spinning and the whole cgroup is THROTTLED for tens of milliseconds at a time
    (cpu.stat: nr_throttled) -- a 40-60 ms stall in the middle of a timed region, and the reason 256-thread CPU baselines collapse."""
    n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:  # noqa: BLE001  (cgroup v1 / no cgroup: keep os.cpu_count())
        pass
    return n


# =====================================================================================================================================
# Multi-GPU runs are supervised.  Launched as one rank of N > 1 (torch.distributed.run's environment), this process does not join the
# process group itself: it becomes the SUPERVISOR of its rank and runs the benchmark in a child process, so that a configuration that
# hangs (a collective one rank never enters, an ordering bug of in-flight all-gathers) or dies still ends in a labelled JSON line:
#
#   attempt 0   what the command asks for (default: the owner-computes exchange, pipelined in interleaved halves, parameter all-gathers
#               in flight across the iteration boundary, several workgroups per owned brick, ProcessGroup entry points without wrappers)
#   attempt 1   owner-computes, conservative: one contiguous ownership range per rank, every all-gather waited for at the end of its
#               iteration, torch.distributed's public collectives only, one workgroup per brick
#   attempt 2   --exchange dense: reduce-scatter -> sharded Adam -> all-gather of the flat bucket (public collectives only)
#
# Every attempt VALIDATES itself before anything is timed (the first iterations compare the replicas' parameter checksums) under a
# watchdog: once all ranks are ready, the validation has RF_BENCH_VALIDATE_TIMEOUT_S (30) seconds; the timed region and the final
# replica check RF_BENCH_RUN_TIMEOUT_S (300).  A rank whose child dies, fails its validation or runs out of time says so in a marker
# file of the run's directory (shared by the N supervisors of the node); every supervisor then kills its child and all move on to the
# next attempt together, on a rendezvous port of its own.  The line printed by the attempt that completes carries
# ``distributed.exchange_fallback_reason`` = why the earlier ones were abandoned.  (The reference has no multi-GPU code to mirror:
# the step being wrapped is one device's loss.backward(); optimizer.step(), modules/trainers.py:338-341.)
# =====================================================================================================================================
EXIT_FALLBACK = 75  # a worker's "this configuration failed its validation on some rank; all ranks agreed; try the next one"

ATTEMPTS = [
    ("as asked: owner-computes, pipelined", {"RF_OWNER_PIPELINED": "1"}, []),
    ("owner-computes, conservative (contiguous ownership, no all-gather left in flight, public collectives, one workgroup per brick)",
     {"RF_OWNER_PIPELINED": "0", "RF_OWNER_HALVES": "1", "RF_OWNER_OVERLAP_PARAMETERS": "0", "RF_DIST_FAST": "0", "RF_OWNER_BRICK_PARTS": "1"}, []),
    ("dense exchange (reduce-scatter -> sharded Adam -> all-gather)", {"RF_DIST_FAST": "0"}, ["--exchange", "dense"]),
]


def _mark(name: str, text: str = "") -> None:
    """worker side: leave a marker of this rank's progress in the run directory (no-op without a supervisor)"""
    d = os.environ.get("RF_BENCH_RUN_DIR")
    if d and os.environ.get("RF_BENCH_WORKER"):
        path = os.path.join(d, f"a{os.environ.get('RF_BENCH_ATTEMPT', '0')}.r{os.environ.get('RANK', '0')}.{name}")
        with open(path + ".tmp", "w") as fh:
            fh.write(text)
        os.replace(path + ".tmp", path)


def supervise(argv) -> int:
    import glob
    import signal
    import socket
    import subprocess

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ["WORLD_SIZE"])
    # one directory per RUN, shared by the node's supervisors: named after the launcher they all descend from -- its pid AND its start
    # time (a recycled pid + port cannot resurrect another run's markers)
    try:
        launcher_start = open(f"/proc/{os.getppid()}/stat").read().rsplit(")", 1)[1].split()[19]
    except Exception:  # noqa: BLE001
        launcher_start = "0"
    run_dir = os.environ.get("RF_BENCH_RUN_DIR") or f"/tmp/rf_bench_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}_{launcher_start}"
    os.makedirs(run_dir, exist_ok=True)
    validate_s = float(os.environ.get("RF_BENCH_VALIDATE_TIMEOUT_S", "30"))
    ready_s = float(os.environ.get("RF_BENCH_READY_TIMEOUT_S", "600"))
    run_s = float(os.environ.get("RF_BENCH_RUN_TIMEOUT_S", "300"))
    report_s = float(os.environ.get("RF_BENCH_REPORT_TIMEOUT_S", "1200"))
    attempts = list(range(len(ATTEMPTS)))
    if "--exchange" in argv and argv[argv.index("--exchange") + 1] == "dense":
        attempts = [2]
    reasons = []

    def say(msg):
        print(f"[bench supervisor, rank {rank}] {msg}", file=sys.stderr, flush=True)

    def files(k, kind):
        return sorted(glob.glob(os.path.join(run_dir, f"a{k}.r*.{kind}")))

    def write(k, kind, text):
        path = os.path.join(run_dir, f"a{k}.r{rank}.{kind}")
        with open(path + ".tmp", "w") as fh:
            fh.write(text)
        os.replace(path + ".tmp", path)

    # a supervisor that is told to stop (the launcher's timeout, ^C) takes its worker with it: no orphan keeps a GPU busy
    current = {"child": None}

    def stop(signum, _frame):
        child_ = current["child"]
        if child_ is not None and child_.poll() is None:
            child_.send_signal(signal.SIGKILL)
        os._exit(128 + signum)

    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sig, stop)

    for pos, k in enumerate(attempts):
        label, env_add, arg_add = ATTEMPTS[k]
        last = pos == len(attempts) - 1
        # a rendezvous of its own per attempt: rank 0's supervisor picks a free port, the others read it
        port_file = os.path.join(run_dir, f"a{k}.port")
        if rank == 0:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            with open(port_file + ".tmp", "w") as fh:
                fh.write(str(port))
            os.replace(port_file + ".tmp", port_file)
        t_wait = time.time()
        while not os.path.exists(port_file):
            if time.time() - t_wait > 120.0:
                say(f"attempt {k}: no rendezvous port from rank 0's supervisor")
                return 1
            time.sleep(0.05)
        port = int(open(port_file).read())
        env = dict(os.environ)
        env.update(env_add)
        env.update({"RF_BENCH_WORKER": "1", "RF_BENCH_ATTEMPT": str(k), "RF_BENCH_RUN_DIR": run_dir, "RF_BENCH_LAST_ATTEMPT": "1" if last else "0",
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "TORCHELASTIC_USE_AGENT_STORE": "False"})
        if reasons:
            env["RF_BENCH_FALLBACK_REASON"] = " | ".join(reasons)
        if pos > 0 and not os.environ.get("RF_INJECT_IN_ALL_ATTEMPTS"):  # (the test hooks break the first attempt only)
            for name in ("RF_OWNER_INJECT_HANG", "RF_OWNER_INJECT_FAILURE"):
                env.pop(name, None)
        # (RF_BENCH_WORKER_CMD: tests/test_bench_supervisor.py drives this protocol on the CPU with a stand-in worker)
        worker = os.environ["RF_BENCH_WORKER_CMD"].split() if os.environ.get("RF_BENCH_WORKER_CMD") else [sys.executable, os.path.abspath(__file__)]
        child = subprocess.Popen(worker + list(argv) + arg_add, env=env)
        current["child"] = child
        t_spawn = time.time()
        t_ready = t_valid = None
        verdict = None  # None: running; "next": abandon this attempt; int: return code to leave with
        committed, t_commit = False, None
        while verdict is None:
            time.sleep(0.1)
            rc = child.poll()
            now = time.time()
            committed = len(files(k, "done")) == world  # timed region + final replica check passed on every rank: no way back
            if rc is not None and (committed or rc != 0):
                if committed:
                    verdict = rc
                elif rc == EXIT_FALLBACK and not last:
                    verdict = "next"
                else:
                    write(k, "dead", f"rank {rank}: worker exited with code {rc}")
                    verdict = rc if last else "next"
                break
            # (a worker that left with 0 before every rank was done -- it cannot, its last act is a barrier -- is treated like a running
            # one: the attempt counts only when ALL ranks are done)
            if committed:
                t_commit = t_commit or now
                if now - t_commit > report_s:  # (the reporting legs behind the timed region are local work of minutes at most)
                    say(f"attempt {k}: worker still running {report_s:.0f} s after every rank was done -- ended")
                    verdict = 1
                    break
                continue
            bad = files(k, "dead") + files(k, "timeout") + files(k, "fail")
            if bad:
                verdict = "next" if not last else 1
                break
            if t_ready is None:
                if len(files(k, "ready")) == world:
                    t_ready = now
                elif now - t_spawn > ready_s:
                    write(k, "timeout", f"rank {rank}: not all ranks ready {ready_s:.0f} s after the workers were started")
            elif t_valid is None:
                if len(files(k, "valid")) == world:
                    t_valid = now
                elif now - t_ready > validate_s:
                    write(k, "timeout", f"rank {rank}: validation steps not finished {validate_s:.0f} s after all ranks were ready (a hang)")
            elif now - t_valid > run_s:
                write(k, "timeout", f"rank {rank}: timed region not finished {run_s:.0f} s after the validation")
        if child.poll() is None:
            child.send_signal(signal.SIGKILL)
            child.wait()
        if committed and isinstance(verdict, int):
            # every rank finished the timed region and the replica check: this attempt IS the run, whatever the exit code of a rank's
            # reporting legs (a crash there must not send one supervisor into another attempt that nobody else joins)
            return int(verdict)
        if verdict == "next" or (isinstance(verdict, int) and verdict != 0 and not last):
            # all supervisors leave the attempt before anyone starts the next one (their children must be gone: GPU memory, ports)
            write(k, "left", "")
            t_wait = time.time()
            while len(files(k, "left")) < world and time.time() - t_wait < 60.0:
                time.sleep(0.05)
            # (read behind that barrier: every rank's supervisor has written what it saw -- the rank that died, the peers whose
            # collectives broke on it)
            why = "; ".join(open(f).read().strip() for f in files(k, "dead") + files(k, "timeout") + files(k, "fail")) or "abandoned"
            reasons.append(f"attempt {k} [{label}]: {why}")
            say(f"attempt {k} [{label}] abandoned: {why}")
            continue
        return int(verdict)
    return 1


if __name__ == "__main__" and int(os.environ.get("WORLD_SIZE", "1")) > 1 and not os.environ.get("RF_BENCH_WORKER") and not os.environ.get("RF_BENCH_NO_SUPERVISOR"):
    sys.exit(supervise(sys.argv[1:]))

os.environ.setdefault("OMP_NUM_THREADS", str(host_cpu_quota()))  # (before torch creates its thread pools)

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import thr3ed_atom_amd as rf  # noqa: E402
from thr3ed_atom_amd import distributed as rfdist  # noqa: E402
from thr3ed_atom_amd import ops  # noqa: E402
from thr3ed_atom_amd.trainers import PosedImagesInMemory, TrainStepper  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GBS = 6300.0  # what a streaming kernel measures on it (same guide): the second denominator of the roofline line
L2_PEAK_TBS = 34.5  # aggregate L2 bandwidth, same guide
LDS_READ_PEAK_TBS = 150.0  # aggregate ds_read_b128 rate, same guide
NEAR = float(np.float32(2.0) * 0.9)  # hotdog-like bounds, SURVEY.md 8d
FAR = float(np.float32(6.0) * 1.1)
RADIUS = 4.0311
WORLD = 3.0


def make_grid(dev, G, sh_degree, seed, sparse=False, storage="reference", rho=None):
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    F = 3 * (sh_degree + 1) ** 2
    dens = torch.empty((G, G, G, 1), device=dev).uniform_(-1.0, 1.0, generator=gen)
    feat = torch.empty((G, G, G, F), device=dev).uniform_(-1.0, 1.0, generator=gen)
    if sparse:  # a blob: positive raw density inside radius ~0.75 (SURVEY.md 8d cfg5 recipe)
        ax = ((torch.arange(G, device=dev, dtype=torch.float32) + 0.5) / G * WORLD - WORLD / 2) / (WORLD / 2)
        r = torch.sqrt(ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2)
        dens = (0.5 - r + 0.05 * dens[..., 0])[..., None].contiguous()
    return rf.VoxelGrid(
        dens,
        feat,
        rf.VoxelSize(WORLD / G, WORLD / G, WORLD / G),
        density_preactivation=torch.nn.Identity(),
        density_postactivation=torch.nn.ReLU(),
        expected_density_scale=rf.compute_expected_density_scale_for_relu_field_grid((WORLD,) * 3) if rho is None else rho,
        tunable=True,
        storage=storage,
    )


def count_inside(origins, directions, num_samples, aabb):
    """samples strictly inside the AABB at the un-jittered sample positions (harness-side, torch ops)"""
    t = torch.linspace(0.0, 1.0, num_samples, device=origins.device)
    z = NEAR * (1.0 - t) + FAR * t
    total = 0
    for s in range(0, origins.shape[0], 65536):
        o, d = origins[s : s + 65536], directions[s : s + 65536]
        p = o[:, None, :] + d[:, None, :] * z[None, :, None]
        m = torch.ones(p.shape[:2], dtype=torch.bool, device=p.device)
        for a, (lo, hi) in enumerate(aabb):
            m &= (p[..., a] > lo) & (p[..., a] < hi)
        total += int(m.sum().item())
    return total


MODEL_ERRORS = []


def frac(bytes_, ms, what=""):
    """fraction of the HBM peak.  A value above 1 cannot be a bandwidth: it means the byte model charges the kernel for work it
    does not do.  Such a figure is never printed: it is reported as None, shouted on stderr and listed in the JSON line."""
    f = bytes_ / 1e9 / (ms / 1e3) / HBM_PEAK_GBS
    if f > 1.0:
        msg = f"ROOFLINE MODEL ERROR: {what}: {bytes_:.4g} B in {ms:.4f} ms = {f:.3f} of the HBM peak (> 1)"
        print(msg, file=sys.stderr)
        MODEL_ERRORS.append(msg)
        return None
    return f


KERNEL_SOURCE = os.path.join(ROOT, "thr3ed_atom_amd", "csrc", "relu_field_kernels.hip")


def kernel_source_sha256():
    import hashlib

    return hashlib.sha256(open(KERNEL_SOURCE, "rb").read()).hexdigest()


def load_pmc_table():
    """profiles/pmc_traffic.json (tools/make_pmc_traffic.py) is tied to the kernel source it was measured on by a sha256 of
    relu_field_kernels.hip: a table measured on other kernels is STALE -- its byte counts are not used for any fraction (the line
    says ``traffic_stale: true`` and falls back to the algorithmic bytes on processed units)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        table = json.load(open(path))
    except Exception:
        return {}, True
    stale = table.get("_kernel_source_sha256") != kernel_source_sha256()
    return ({} if stale else table), stale


def respawn_under_torchrun(gpus: int):
    """``python bench.py --gpus N`` with N > 1 and no torchrun environment: re-exec this command as N ranks (one per GPU, RCCL)
    under torch.distributed.run on 127.0.0.1.  Never silently runs fewer ranks than asked for."""
    import socket

    if torch.cuda.device_count() < gpus and os.environ.get("RF_SINGLE_DEVICE") != "1":  # (test hook: every rank on device 0, over gloo)
        raise SystemExit(f"--gpus {gpus} but only {torch.cuda.device_count()} HIP device(s) are visible")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def cpu_baseline(grid, rays_cpu, pixels_cpu, num_samples, n_train, n_fwd, threads=0, reps=3, n_full=16384):
    """The oracle's float32 CPU path (the same ATen ops the reference calls) on bounded samples of the same workloads
    (SURVEY 8d protocol): cfg3 = specular+diffuse fwd+bwd of ``n_full`` rays -- ONE full training batch, one timed step after a
    small warm-up (~20 s) -- and, for the spread, the median of ``reps`` repeats on ``n_train`` rays; cfg2 = a forward-only chunk of
    ``n_fwd`` rays.  Threads = the cores this process may really use (the cgroup quota, host_cpu_quota()), at most 64."""
    from oracle import relu_field_oracle as orc  # checker / baseline only

    cores = threads if threads > 0 else min(host_cpu_quota(), 64)
    torch.set_num_threads(cores)
    dens = grid.densities.detach().cpu().clone().requires_grad_(True)
    feat = grid.features.detach().cpu().clone().requires_grad_(True)
    aabb = tuple(tuple(r) for r in grid.aabb)
    rho = grid.expected_density_scale

    def rays_of(n):  # (a leg may hold more rays than the batch handed over: repeat it)
        reps_ = (n + rays_cpu[0].shape[0] - 1) // rays_cpu[0].shape[0]
        return rays_cpu[0].repeat(reps_, 1)[:n], rays_cpu[1].repeat(reps_, 1)[:n], pixels_cpu.repeat(reps_, 1)[:n]

    def train(n):
        o, d, px = rays_of(n)
        dens.grad = feat.grad = None
        total = 0.0
        for diffuse in (False, True):
            t_rand = torch.rand(n, num_samples)
            out = orc.render(dens, feat, o, d, aabb, NEAR, FAR, num_samples, rho, "relu", white_bkgd=True, render_diffuse=diffuse, t_rand=t_rand, interp="aten")
            total = total + torch.nn.functional.l1_loss(out["colour"], px)
        total.backward()

    def forward(n):
        o, d, _ = rays_of(n)
        with torch.no_grad():
            orc.render(dens, feat, o, d, aabb, NEAR, FAR, num_samples, rho, "relu", white_bkgd=True, t_rand=torch.rand(n, num_samples), interp="aten")

    def timed(fn, n, reps_):
        fn(min(256, n))  # warm-up (thread pool, page-in)
        ts = []
        for _ in range(reps_):
            t0 = time.perf_counter()
            fn(n)
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts))

    dt_train = timed(train, n_train, reps)
    dt_full = timed(train, n_full, 1) if n_full > 0 else None
    dt_fwd = timed(forward, n_fwd, reps) if n_fwd > 0 else None
    value = 2 * n_full * num_samples / dt_full if dt_full else 2 * n_train * num_samples / dt_train
    return {
        "value": value,
        "unit": "ray-samples/s",
        "cores": cores,
        "host_cores_visible": os.cpu_count(),
        "host_cpu_quota_cores": host_cpu_quota(),
        "kind": "port",
        "sample": (f"configs[2] in full: ONE training-step core (specular+diffuse fwd+bwd, no optimiser) of {n_full} rays x {num_samples} samples after a 256-ray warm-up, "
                   f"{dt_full:.2f} s" if dt_full else f"median of {reps} repeats of the training-step core on {n_train} rays")
        + f"; same 128^3 SH-2 grid, oracle with interp='aten' (F.grid_sample), torch {torch.__version__} CPU fp32, {cores} threads = the cores this "
        f"process may use (cgroup quota; os.cpu_count() shows {os.cpu_count()})",
        "small_batch": {"value": 2 * n_train * num_samples / dt_train, "unit": "ray-samples/s",
                        "sample": f"median of {reps} repeats after 1 warm-up on {n_train} rays; {dt_train:.2f} s/step"},
        "forward_only": None if dt_fwd is None else {
            "value": n_fwd * num_samples / dt_fwd,
            "unit": "ray-samples/s",
            "sample": f"median of {reps} repeats: forward-only chunk of {n_fwd} rays x {num_samples} samples (configs[1] renders 800x800 = 640000 rays in 32768-ray chunks); {dt_fwd:.2f} s/chunk",
        },
    }


def cfg1_leg(dev, cores, reps=3):
    """configs[0] in full on both sides (the reference's own CPU-runnable case): 64^3 SH-degree-0 U(-1,1) field, one
    64x64 frame (f = 88.9), 32 samples per ray, pose_spherical(30, -30, 4.0311), jitter off."""
    from oracle import relu_field_oracle as orc  # checker / baseline only

    G, S1, H = 64, 32, 64
    g1 = make_grid(dev, G, 0, seed=1)
    cfg1 = rf.SHVoxGridRenderConfig(S1, rf.CameraBounds(NEAR, FAR), perturb_sampled_points=False, white_bkgd=True)
    model = rf.VolumetricModel(g1, rf.render_sh_voxel_grid, cfg1, device=dev)
    intr, pose = rf.CameraIntrinsics(H, H, 88.9), rf.pose_spherical(30.0, -30.0, RADIUS)
    gpu_s = time_frames(lambda: model.render(pose, intr), 5)
    gpu = model.render(pose, intr).colour.cpu()
    torch.set_num_threads(cores)
    dens, feat = g1.densities.detach().cpu(), g1.features.detach().cpu()
    o, d = orc.cast_rays(H, H, 88.9, pose.rotation.cpu(), pose.translation.cpu())
    aabb = tuple(tuple(r) for r in g1.aabb)

    def cpu():
        with torch.no_grad():
            return orc.render(dens, feat, o.reshape(-1, 3), d.reshape(-1, 3), aabb, NEAR, FAR, S1, g1.expected_density_scale, "relu", white_bkgd=True, interp="aten")["colour"]

    ref = cpu()  # warm-up + the check
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        cpu()
        ts.append(time.perf_counter() - t0)
    cpu_s = float(np.median(ts))
    n = H * H * S1
    return {"workload": "configs[0]: 64^3 SH-0 field, 64x64 frame, 32 samples/ray, jitter off", "ray_samples": n,
            "cpu_ms": cpu_s * 1e3, "cpu_ray_samples_per_s": n / cpu_s, "gpu_ms_incl_launch_and_sync": gpu_s * 1e3,
            "gpu_ray_samples_per_s": n / gpu_s, "max_abs_colour_difference_gpu_vs_cpu": float((gpu.reshape(-1, 3) - ref).abs().max())}


def frame_kernel_of(grid, intr, use_mask, pose=None):
    """which kernel rf_render_forward picks for a frame of this camera: asked of the library itself (rf_frame_render_kernel, the
    host-side restatement-free dispatch rule: ray packets where an 8 x 8 pixel tile's footprint at the volume's centre stays within
    2 voxels -- 3 with the occupancy mask --, $RF_FRAME_TILES overrides)"""
    import ctypes as C

    from thr3ed_atom_amd import _lib

    pose = pose if pose is not None else rf.pose_spherical(30.0, -30.0, RADIUS)
    cam = _lib.RFCamera()
    cam.height, cam.width, cam.focal = int(intr.height), int(intr.width), float(np.float32(intr.focal))
    rot, trans = pose.rotation.detach().cpu().reshape(3, 3), pose.translation.detach().cpu().reshape(3)
    for a in range(3):
        for b in range(3):
            cam.pose[4 * a + b] = float(rot[a, b])
        cam.pose[4 * a + 3] = float(trans[a])
    if use_mask and not grid.occupancy_current():
        grid.build_occupancy()
    rf_grid = grid.forward_rf_grid(use_occupancy=bool(use_mask))
    flags = ops.render_flags(True, False, False, bool(use_mask))
    rc = _lib.load().rf_frame_render_kernel(C.byref(rf_grid), C.byref(cam), int(flags))
    _lib.check(min(rc, 0), "rf_frame_render_kernel")
    return "render_frame_tile_kernel (ray packets: one wave per 8x8 pixel tile, window staged through LDS)" if rc == 1 else "render_forward_kernel (one wave per ray)"


def time_frames(fn, frames, kernel=None):
    """median of per-frame wall times (sync before and after every frame).  With ``kernel`` (the name of a launch): also the median
    of that launch's HIP-event time over THE SAME frames (ops.KernelTimer: events on the launch stream, inside the timed call) and
    its launches per frame -- returned as (wall seconds, kernel ms, launches).  A kernel cannot outlast the synchronised call that
    contains it: the two medians come from the same frames so that the line keeps that order."""
    fn()  # warm-up
    ts, ks, launches = [], [], 0
    for _ in range(frames):
        timer = ops.KernelTimer() if kernel else None
        ops.KERNEL_TIMER = timer
        try:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        finally:
            ops.KERNEL_TIMER = None
        if kernel:
            rec = timer.summary()[kernel]
            ks.append(rec["total_ms"])
            launches = rec["launches"]
    if kernel:
        return float(np.median(ts)), float(np.median(ks)), launches
    return float(np.median(ts))


def multi_gpu_forward_legs(args, dev, rank, world, intr, bounds):
    """``north_star``: forward throughput "reported at 1/2/4/8 GPUs".  Every rank takes part (the reference's counterpart is one device's
    loop over a frame's chunks, modules/volumetric_model.py:143-172; rays are independent, the grid is replicated):

      frame_parallel   rank r renders ITS OWN pose (r-th view of an orbit), whole frames, nothing exchanged: the throughput form
                       (a test-set evaluation, a turntable video) -- value = N frames per max-over-ranks frame time;
      sharded_frame    ONE pose, ``VolumetricModel.render(data_parallel=True)``: every rank renders its shard_range of the pixels and
                       the [n, 6] results are all-gathered -- the latency form; also timed without the gather (each rank keeps its rows).

    configs[1] (the grid of the fwd_render leg: same seed, same field) and configs[4] (256^3 sparse field, 512 samples, occupancy mask).
    Timed like the step: barrier + synchronize on both sides, max over ranks.  Returns {"fwd_render": {...}, "highres_render": {...}}."""
    import torch.distributed as dist

    H, W = int(intr.height), int(intr.width)

    def timed(fn, frames):
        fn()  # warm-up (and the first use of a collective shape)
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(frames):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        t = torch.tensor([(time.perf_counter() - t0) / frames], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def legs(model, S_, frames, **over):
        own_pose = rf.pose_spherical(30.0 + 360.0 / world * rank, -30.0, RADIUS)
        pose = rf.pose_spherical(30.0, -30.0, RADIUS)
        lo, hi = rfdist.shard_range(H * W)
        dt_fp = timed(lambda: model.render(own_pose, intr, **over), frames)
        dt_sh = timed(lambda: model.render(pose, intr, data_parallel=True, **over), frames)
        from thr3ed_atom_amd.renderers import render_sh_voxel_grid_frame

        cfg_ = model._update_render_config(model.render_config, over)
        dt_rows = timed(lambda: render_sh_voxel_grid_frame(model.thre3d_repr, intr, pose, cfg_, first_ray=lo, num_rays=hi - lo), frames)
        # the same sharded frame served by the per-ray kernel ($RF_FRAME_TILES is read per call): a packet tile's march is ~0.5 ms long
        # whatever the launch size, so a rank's share of a frame stops shrinking with N once its tiles no longer fill the machine, while the
        # per-ray kernel's time follows the shard (one MI355X, shard of 1/8 of this frame: 0.62 against 0.47 ms, configs[4]: 1.00 against
        # 0.45 -- profiles/r06_sharded_frame_shard_times.json).  The default keeps ONE kernel per frame whatever N is (the sharded frame is
        # then the single-GPU frame bit for bit); this figure says what the other choice would buy at this N.
        keep_env = os.environ.get("RF_FRAME_TILES")
        os.environ["RF_FRAME_TILES"] = "0"
        try:
            dt_sh_per_ray = timed(lambda: model.render(pose, intr, data_parallel=True, **over), frames)
        finally:
            if keep_env is None:
                os.environ.pop("RF_FRAME_TILES", None)
            else:
                os.environ["RF_FRAME_TILES"] = keep_env
        # the sharded frame IS the single-GPU frame (jitter off for the comparison; every rank checks the frame it ended up with)
        a = model.render(pose, intr, data_parallel=True, perturb_sampled_points=False, **over)
        b = model.render(pose, intr, perturb_sampled_points=False, **over)
        same = torch.tensor([int(torch.equal(a.colour, b.colour) and torch.equal(a.depth, b.depth))], device=dev, dtype=torch.int32)
        dist.all_reduce(same, op=dist.ReduceOp.MIN)
        return {
            "frame_parallel": {"what": f"rank r renders its own pose (view r of a {world}-view orbit), whole {H}x{W} frames, no exchange; max over ranks", "ms_per_frame_per_gpu": dt_fp * 1e3,
                               "frames_per_s": world / dt_fp, "rays_per_s": world * H * W / dt_fp, "ray_samples_per_s": world * H * W * S_ / dt_fp},
            "sharded_frame": {"what": "ONE pose, VolumetricModel.render(data_parallel=True): every rank renders its contiguous shard of the pixels, one all-gather of the [n, 6] results "
                              "(every rank ends up with the whole frame); max over ranks", "ms_per_frame": dt_sh * 1e3, "rays_per_s": H * W / dt_sh, "ray_samples_per_s": H * W * S_ / dt_sh,
                              "ms_per_frame_without_the_gather": dt_rows * 1e3, "equals_single_gpu_frame_bit_for_bit": bool(int(same.item())),
                              "ms_per_frame_with_the_per_ray_kernel": dt_sh_per_ray * 1e3},
            "n_gpus": world, "frames_timed": frames,
        }

    out = {}
    if args.render_frames > 0:
        g = make_grid(dev, args.grid, args.sh_degree, seed=42, storage=args.storage)
        cfg = rf.SHVoxGridRenderConfig(args.samples, bounds, perturb_sampled_points=True, white_bkgd=True)
        out["fwd_render"] = legs(rf.VolumetricModel(g, rf.render_sh_voxel_grid, cfg, device=dev), args.samples, args.render_frames)
        del g
        torch.cuda.empty_cache()
    if args.highres_frames > 0:
        hg = make_grid(dev, 256, args.sh_degree, seed=11, sparse=True, storage=args.storage)
        hg.build_occupancy()
        hcfg = rf.SHVoxGridRenderConfig(512, bounds, perturb_sampled_points=True, white_bkgd=True)
        out["highres_render"] = legs(rf.VolumetricModel(hg, rf.render_sh_voxel_grid, hcfg, device=dev), 512, args.highres_frames, use_occupancy_mask=True)
        del hg
        torch.cuda.empty_cache()
    return out


def train_leg_at_reference_final_stage(args, dev, dataset, bounds):
    """The training iteration at the operating point the reference's CLI ENDS at: a 256^3 SH-2 grid, 512 samples per ray, 16384 rays
    (train_sh_based_voxel_grid_with_posed_images.py:55,88-90; modules/trainers.py:125-129: 32^3 -> 64^3 -> 128^3 -> 256^3).  Same step,
    same images, U(-1,1) initialisation; 470 M parameters: the optimizer's 24 B per parameter (11.3 GB) dominate.  A reporting leg
    beside the headline (BASELINE.json quotes its metric on 128^3)."""
    G2, S2, R = 256, 512, args.rays
    C = 3 * (args.sh_degree + 1) ** 2 + 1
    grid = make_grid(dev, G2, args.sh_degree, seed=42, storage="split")
    cfg = rf.SHVoxGridRenderConfig(S2, bounds, perturb_sampled_points=True, white_bkgd=True)
    model = rf.VolumetricModel(grid, rf.render_sh_voxel_grid, cfg, device=dev)
    stepper = TrainStepper(model, R, learning_rate=0.03, apply_diffuse_render_regularization=True, ray_selection="keyed", data_parallel=False)
    executor = stepper.fused and stepper.merged_bricks
    torch.manual_seed(4321)
    batches = dataset.image_batches(args.images)
    for _ in range(3):
        stepper.step(dataset, next(batches))
    steps = args.train256_steps
    timed_idx = sorted({0, steps // 2, steps - 1}) if executor else []
    events = [ops.StepEvents() for _ in timed_idx]
    counts = torch.zeros((max(1, len(timed_idx)), 2), dtype=torch.int64, device=dev)
    used_keys = torch.zeros((max(1, len(timed_idx)), 2), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = 0
    for i in range(steps):
        on = k < len(timed_idx) and i == timed_idx[k]
        stepper.step_events = events[k] if on else None
        stepper.step(dataset, next(batches))
        if on:
            ex = stepper._exec["tensors"]
            for j in (0, 1):  # (device-side: no sync)
                off = ex[f"pass{j}"]["offsets"]
                counts[k, j].copy_(off[-1] - off[0], non_blocking=True)
                used_keys[k, j].copy_((off[1:] > off[:-1]).sum(), non_blocking=True)
            k += 1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    stepper.step_events = None
    kernels = {}
    if events:
        per = [e.elapsed_ms() for e in events]
        kernels = {name: float(np.mean([p_[name] for p_ in per])) for name in per[0]}
    rec = counts.cpu().numpy().astype(np.float64).mean(axis=0)
    nkeys = int(stepper._exec["tensors"]["pass0"]["offsets"].numel() - 1) if executor else None
    nparam = G2**3 * C
    bricks_ms = kernels.get("brick_accumulate")
    brick_bytes = rec[0] * 4 * ops.expanded_record_floats(grid) + rec[1] * 4 * ops.expanded_record_floats(grid, True) + nparam * 24
    leg = {
        "workload": f"the reference CLI's final stage: {G2}^3 SH-{args.sh_degree} ReLU field (U(-1,1) init), {R} random distinct pixels of the {args.images} images per step, {S2} jittered samples/ray, "
                    "specular+diffuse fwd+bwd, L1+L1, Adam in the brick flush (one library call per step)",
        "ms_per_step": dt * 1e3, "ray_samples_per_s": 2 * R * S2 / dt, "steps": steps, "warmup": 3,
        "parameters": nparam, "optimizer_bytes_per_step": nparam * 24, "brick_nodes": "4x8x8" if getattr(stepper, "brick_size", 8) == rf.ops.BRICK_4X8X8 else f"{int(getattr(stepper, 'brick_size', 8))}^3", "backward": stepper.backward,
        "kernels_ms": kernels,
        "records_per_step": {"specular": float(rec[0]), "diffuse": float(rec[1])},
        "key_tables": None if nkeys is None else {"keys": nkeys, "non_empty_specular": float(used_keys[:, 0].double().mean().item()), "non_empty_diffuse": float(used_keys[:, 1].double().mean().item())},
        "brick_pass": None if not bricks_ms else {"avg_launch_ms": bricks_ms, "compulsory_bytes": brick_bytes, "frac_of_hbm_peak": frac(brick_bytes, bricks_ms, "train_256 brick pass"),
                                                 "frac_of_achievable": brick_bytes / 1e9 / (bricks_ms / 1e3) / HBM_ACHIEVABLE_GBS, "share_of_step": bricks_ms / (dt * 1e3)},
        "oracle_checked_by": "tests/test_hip_baseline_size.py::test_bench_train_step_against_oracle_at_baseline_size[256-...]",
    }
    stepper.flat.detach()
    del stepper, model, grid
    torch.cuda.empty_cache()
    return leg


def run_guarded(fn, timeout_s, dev):
    """``fn()`` on a helper thread, waited for at most ``timeout_s``: (result, None), or (None, why) when it raised or is still
    running -- a collective another rank never enters must not cost the line of a run whose timed region is already done."""
    import threading

    box = {}

    def body():
        try:
            torch.cuda.set_device(dev)
            box["result"] = fn()
        except BaseException as exc:  # noqa: BLE001
            box["error"] = f"{type(exc).__name__}: {exc}"

    th = threading.Thread(target=body, daemon=True)
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        return None, f"not finished after {timeout_s:.0f} s (abandoned; the process leaves without joining the group's shutdown)"
    if "error" in box:
        return None, box["error"]
    return box.get("result"), None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50, help="untimed steps first")
    ap.add_argument("--grid", type=int, default=128)
    ap.add_argument("--sh-degree", type=int, default=2)
    ap.add_argument("--rays", type=int, default=16384, help="ray batch per GPU (reference CLI default)")
    ap.add_argument("--samples", type=int, default=256)
    ap.add_argument("--image-size", type=int, default=800)
    ap.add_argument("--images", type=int, default=8)
    ap.add_argument("--render-frames", type=int, default=5, help="full-frame forward renders timed for fwd_render (0 = skip)")
    ap.add_argument("--highres-frames", type=int, default=5, help="frames timed for the 256^3 / 512-sample configs[4] render (0 = skip)")
    ap.add_argument("--cpu-rays", type=int, default=2048, help="rays of the cpu_baseline training sample (0 = skip the CPU baseline)")
    ap.add_argument("--cpu-full-rays", type=int, default=16384, help="rays of the cpu_baseline's full training batch (SURVEY 8d cfg3; one timed step, ~20 s; 0 = skip)")
    ap.add_argument("--cpu-fwd-rays", type=int, default=32768, help="rays of the cpu_baseline forward-only chunk (SURVEY 8d: one parallel_rays_chunk_size chunk)")
    ap.add_argument("--train256-steps", type=int, default=10, help="steps timed for the train_256 leg: the iteration at the reference CLI's final stage, 256^3 / 512 samples (0 = skip)")
    ap.add_argument("--dropin-steps", type=int, default=20, help="steps timed for the strict drop-in configuration (0 = skip)")
    ap.add_argument("--storage", choices=["split", "bricked", "reference"], default="split",
                    help="HBM layout of the grid: split = MI355X-native (what the trainer uses), reference = the reference's two tensors")
    ap.add_argument("--ray-selection", choices=["keyed", "randperm"], default="keyed",
                    help="how a step picks its 16384 random pixels: keyed = fused keyed-permutation kernel (trainer default), "
                    "randperm = torch.randperm over all 5.12 M pixels like the reference")
    ap.add_argument("--backward", choices=["auto", "atomic", "binned"], default="auto",
                    help="gradient scatter of the train step: float32 atomics, or records binned by brick + atomic-free LDS accumulation")
    ap.add_argument("--deterministic", action="store_true", help="binned backward: stable radix sort instead of the counting sort")
    ap.add_argument("--no-fuse-optimizer", action="store_true", help="keep the gradient bucket and the separate Adam kernel on one GPU")
    ap.add_argument("--dp-style-step", action="store_true",
                    help="one GPU: run the step exactly the way a data-parallel rank runs it (a 1-rank RCCL group: same code path, collectives of world size 1)")
    ap.add_argument("--exchange", choices=["auto", "owner", "dense"], default="auto",
                    help="data-parallel exchange: owner = owner-computes (record slices all-to-all -> merged brick pass + Adam on the rank's own bricks -> "
                    "parameter all-gather), dense = reduce-scatter of the gradient bucket -> sharded Adam -> all-gather")
    ap.add_argument("--timed-steps", type=int, default=5, help="how many of the --steps record per-kernel HIP events")
    ap.add_argument("--second-point-rays", type=int, default=32768, help="rays per GPU of the second weak-scaling point timed behind the windows (0 = skip)")
    ap.add_argument("--windows", type=int, default=9, help="further windows of --steps steps timed behind the official one (ms_per_step_windows)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="torch CPU threads for cpu_baseline (0 = the cgroup CPU quota, at most 64)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        respawn_under_torchrun(args.gpus)  # does not return
    if args.dp_style_step and args.gpus == 1 and "WORLD_SIZE" not in os.environ:
        import socket

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        torch.distributed.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
        rfdist.FORCE_COLLECTIVES = True
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:  # (checked before any rendezvous is attempted)
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: launch with torch.distributed.run --nproc-per-node {args.gpus} (or without it: bench.py spawns the ranks itself)")
    rank, local_rank, world = rfdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} (or without it: bench.py spawns the ranks itself)")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the render path has no CPU fallback)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    backend = torch.distributed.get_backend() if torch.distributed.is_initialized() else None
    rccl_world = 1
    if torch.distributed.is_initialized():  # counted by a collective, not read from the environment: every rank adds a one
        ones = torch.ones(1, device=dev, dtype=torch.float32)
        torch.distributed.all_reduce(ones)
        rccl_world = int(round(float(ones.item())))
        if rccl_world != world:
            raise SystemExit(f"--gpus {world} but an all-reduce of ones over the process group gives {rccl_world}")

    H = W = args.image_size
    focal = 1111.111 * (W / 800.0)
    intr = rf.CameraIntrinsics(H, W, focal)
    S, R, G = args.samples, args.rays, args.grid
    C = 3 * (args.sh_degree + 1) ** 2 + 1
    bounds = rf.CameraBounds(NEAR, FAR)
    pmc, pmc_stale = load_pmc_table()
    spec = f"sh{args.sh_degree}"

    # ---- synthetic dataset: images of a procedural ground-truth field rendered once (untimed) -------
    gt = make_grid(dev, G, args.sh_degree, seed=7, sparse=True)
    gt_cfg = rf.SHVoxGridRenderConfig(S, bounds, perturb_sampled_points=False, white_bkgd=True)
    gt_model = rf.VolumetricModel(gt, rf.render_sh_voxel_grid, gt_cfg, device=dev)
    poses = [rf.pose_spherical(45.0 * k, -30.0, RADIUS) for k in range(args.images)]
    images = torch.stack([gt_model.render(p, intr).colour.permute(2, 0, 1) for p in poses])
    pose_mat = torch.stack([torch.cat([p.rotation, p.translation], dim=1) for p in poses]).to(dev)
    dataset = PosedImagesInMemory(images, pose_mat, intr, bounds)
    del gt_model, gt

    # ---- model under training: U(-1,1) grid, the reference's initialisation -----------------------
    grid = make_grid(dev, G, args.sh_degree, seed=42, storage=args.storage)
    cfg = rf.SHVoxGridRenderConfig(S, bounds, perturb_sampled_points=True, white_bkgd=True)
    model = rf.VolumetricModel(grid, rf.render_sh_voxel_grid, cfg, device=dev)

    # ---- forward-only full-frame render (configs[1]), timed separately, before training -----------
    fwd_render = None
    if args.render_frames > 0 and rank == 0:
        pose = rf.pose_spherical(30.0, -30.0, RADIUS)
        rays = rf.flatten_rays(rf.cast_rays(intr, pose, dev))
        n_in = count_inside(rays.origins, rays.directions, S, grid.aabb)
        del rays
        kname = f"render_forward[{spec},frame]"
        legs = {}
        # (a) the field the reference initialises (U(-1,1) raw densities x rho = 33): every ray saturates within a few
        #     samples, so the launch touches a small part of the volume -- fast, but no evidence about HBM;
        # (b) the same grid with rho = 0.25: sigma * delta << 1, transmittance stays far from 0, EVERY in-box sample with
        #     positive density fetches its 8 x 28 channels -- rays traverse the whole volume, the HBM-bound regime.
        for leg, rho in (("init_field", None), ("traversal", 0.25)):
            g2 = grid if rho is None else make_grid(dev, G, args.sh_degree, seed=42, storage=args.storage, rho=rho)
            m2 = rf.VolumetricModel(g2, rf.render_sh_voxel_grid, cfg, device=dev)
            dt, kms, launches = time_frames(lambda: m2.render(pose, intr), args.render_frames, kernel=kname)
            # units really processed: samples whose features are gathered = in-box, positive density, T != 0; counted by a
            # save-forward of a ray subset is not possible at frame size, so the record predicate is evaluated on the device
            # by the training-style forward of a 65536-ray sample of the frame and scaled
            sub = rf.flatten_rays(rf.cast_rays(intr, pose, dev))[:: max(1, (H * W) // 65536)][:65536]
            nb = ops.brick_counts(g2, 8)
            hist = torch.zeros(nb[0] * nb[1] * nb[2] * 8, dtype=torch.int32, device=dev)
            flags = ops.render_flags(True, False, False, False)
            ops.render_forward_raw(g2, sub.origins.contiguous(), sub.directions.contiguous(), ops.KeyedJitter(7, 0), S, NEAR, FAR, flags, save=True, key_hist=hist)
            gathered = float(hist.sum().item()) / len(sub) * (H * W)
            inside_sub = count_inside(sub.origins, sub.directions, S, g2.aabb) / len(sub) * (H * W)
            del hist, sub
            alg = gathered * 8 * C * 4 + (inside_sub - gathered) * 8 * 4 * 4 + H * W * 12  # features | density only | outputs (rays are generated in-kernel)
            counter = pmc.get(f"{kname}:{leg}", {}).get("hbm_bytes_per_launch")
            packets = frame_kernel_of(g2, intr, False).startswith("render_frame_tile_kernel")
            legs[leg] = {
                "density_scale": g2.expected_density_scale,
                "frame_kernel": frame_kernel_of(g2, intr, False),
                "ms_per_frame": dt * 1e3,
                "ray_samples_per_s": H * W * S / dt,
                "rays_per_s": H * W / dt,
                "kernel_ms_per_frame": kms,
                "render_launches_per_frame": launches,
                "samples_gathering_features": gathered,
                "algorithmic_GB_processed": alg / 1e9,
                # a coherent frame re-uses every cell across neighbouring rays and the 235 MB grid sits in the 32 MB of L2 + 256 MB
                # of Infinity Cache: the gathers are served on-die, so the algorithmic bytes are priced against the L2 ceiling
                # (34.5 TB/s aggregate, MI355X_MICROARCH.md) -- the HBM side is the counter figure below.  (The ray-packet kernel does
                # not move these bytes through L2 at all: a tile fetches its neighbourhood once per step and the per-sample gathers
                # are LDS reads; the figure is then an EFFECTIVE rate on SURVEY 8d's per-sample byte model, comparable across rounds.)
                "effective_TBps_processed": alg / 1e12 / (kms / 1e3),
                # which ceiling those bytes move against depends on the kernel that served the frame: the ray-packet kernel reads them
                # out of LDS (ds_read_b128: ~150 TB/s aggregate, MI355X_MICROARCH.md) -- and is bound by vector-ALU issue, not by any
                # bandwidth (profiles/r06_frame_packets_counters.md: VALU 78 % busy, LDS pipe 70 %) --, the per-ray kernel through L1 / L2
                "frac_of_lds_read_peak": (alg / 1e12 / (kms / 1e3) / LDS_READ_PEAK_TBS) if packets else None,
                "frac_of_l2_peak_processed": None if packets else alg / 1e12 / (kms / 1e3) / L2_PEAK_TBS,
                "counter_GB_per_launch": None if counter is None else counter / 1e9,
                "frac_hbm": None if counter is None else frac(counter, kms, f"fwd_render.{leg} (counters)"),
            }
            if rho is not None:
                del m2, g2
        fwd_render = {
            "workload": f"configs[1]: {G}^3 SH-{args.sh_degree} ReLU field, {H}x{W}, {S} jittered samples/ray, VolumetricModel.render = ONE launch (rays + jitter generated in-kernel)",
            "inside_fraction": n_in / (H * W * S),
            "nominal_GB_per_frame_survey_8d": (n_in * 8 * C * 4 + H * W * 48) / 1e9,
            **legs,
        }
        torch.cuda.empty_cache()

    # ---- configs[4]: 256^3 grid, 512 samples/ray, sparse scene, exact empty-space skipping (rank 0) ------------
    highres = None
    if args.highres_frames > 0 and rank == 0:
        hg = make_grid(dev, 256, args.sh_degree, seed=11, sparse=True, storage=args.storage)
        hcfg = rf.SHVoxGridRenderConfig(512, bounds, perturb_sampled_points=True, white_bkgd=True)
        hmodel = rf.VolumetricModel(hg, rf.render_sh_voxel_grid, hcfg, device=dev)
        pose = rf.pose_spherical(30.0, -30.0, RADIUS)
        t_build0 = time.perf_counter()
        hg.build_occupancy()
        torch.cuda.synchronize()
        t_build = time.perf_counter() - t_build0
        times = {use: time_frames(lambda: hmodel.render(pose, intr, use_occupancy_mask=use), args.highres_frames) for use in (False, True)}
        # exactness of the skipping, checked per frame kernel (the dispatch may serve the masked render with ray packets and the
        # unmasked one with the per-ray kernel, which differ by summation order): mask on == mask off, bit for bit, in both
        identical = True
        keep_env = os.environ.get("RF_FRAME_TILES")
        for force in ("1", "0"):
            os.environ["RF_FRAME_TILES"] = force
            a = hmodel.render(pose, intr, use_occupancy_mask=False, perturb_sampled_points=False)
            b = hmodel.render(pose, intr, use_occupancy_mask=True, perturb_sampled_points=False)
            identical = identical and bool(torch.equal(a.colour, b.colour) and torch.equal(a.depth, b.depth))
        if keep_env is None:
            del os.environ["RF_FRAME_TILES"]
        else:
            os.environ["RF_FRAME_TILES"] = keep_env
        occ_bits = int(sum(bin(w & 0xFFFFFFFF).count("1") for w in hg.occupancy.cpu().tolist()))
        highres = {
            "workload": f"configs[4]: 256^3 SH-{args.sh_degree} sparse ReLU field, {H}x{W}, 512 jittered samples/ray, VolumetricModel.render (one launch per frame)",
            "frame_kernel_no_mask": frame_kernel_of(hg, intr, False), "frame_kernel_occupancy_mask": frame_kernel_of(hg, intr, True),
            "ms_per_frame_no_mask": times[False] * 1e3,
            "ms_per_frame_occupancy_mask": times[True] * 1e3,
            "ray_samples_per_s_occupancy_mask": H * W * 512 / times[True],
            "occupied_cell_fraction": occ_bits / float(257**3),
            "mask_build_ms": t_build * 1e3,
            "mask_vs_no_mask_bit_identical": identical,
            "oracle_checked_by": "tests/test_hip_baseline_size.py::test_highres_occupancy_render_against_oracle_at_config4_size",
        }
        del hmodel, hg, a, b
        torch.cuda.empty_cache()

    # ---- strict drop-in: reference storage, torch.autograd.Function ops, torch.randperm selection, torch.rand jitter ----
    dropin = None
    if args.dropin_steps > 0 and rank == 0 and world == 1:
        def dropin_leg(selection, jitter="keyed"):
            dgrid = make_grid(dev, G, args.sh_degree, seed=42, storage="reference")
            dcfg = rf.SHVoxGridRenderConfig(S, bounds, perturb_sampled_points=True, white_bkgd=True, jitter=jitter)
            dmodel = rf.VolumetricModel(dgrid, rf.render_sh_voxel_grid, dcfg, device=dev)
            dstep = TrainStepper(dmodel, R, learning_rate=0.03, fused=False, ray_selection=selection, data_parallel=False)
            torch.manual_seed(99)
            dbatches = dataset.image_batches(args.images)
            for _ in range(5):
                dstep.step(dataset, next(dbatches))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.dropin_steps):
                dstep.step(dataset, next(dbatches))
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.dropin_steps
            dstep.flat.detach()
            del dstep, dmodel, dgrid
            torch.cuda.empty_cache()
            return dt

        dt = dropin_leg("keyed")
        dt_rand = dropin_leg("keyed", jitter="torch")
        dt_randperm = dropin_leg("randperm", jitter="torch")
        dt_randperm_blocks = dropin_leg("randperm_blocks", jitter="torch")
        dropin = {
            "workload": "the same iteration the way a reference user gets it: VoxelGrid in the reference's two tensors (its nn.Parameters; forward passes gather from a split-layout "
            "shadow), both renders of the iteration as ONE torch.autograd.Function (VolumetricModel.render_rays_pair: rf_render_forward_pair / offsets + adjoints of both record lists in two "
            "launches; $RF_AUTOGRAD_PAIR=0: one node per render), SHVoxGridRenderConfig defaults (stratified "
            "jitter drawn inside the kernel from a per-call key: the law of torch.rand(N, S) without the tensor), the trainer's L1 + MSE lines of both renders as one autograd.Function launch, "
            "FusedAdam.step = ONE brick pass with Adam in its flush that writes the shadow AND the Parameters' own layout; the batch = distinct uniformly random pixels by the keyed "
            "bijection (the law of torch.randperm(P)[:R] without sorting 5.12 M keys)",
            "ms_per_step": dt * 1e3,
            "ray_samples_per_s": 2 * R * S / dt,
            "ms_per_step_with_torch_rand_jitter": dt_rand * 1e3,
            "ms_per_step_with_torch_rand_jitter_and_torch_randperm_selection": dt_randperm * 1e3,
            # torch.randperm's own draws without one 5.12 M-key permutation PER ITERATION: one per block of floor(P / R) iterations, consumed in slices
            "ms_per_step_with_torch_rand_jitter_and_torch_randperm_per_block_selection": dt_randperm_blocks * 1e3,
            "steps": args.dropin_steps,
            "warmup": 5,
        }

    # ---- training steps: the headline ---------------------------------------------------------------
    def make_stepper(exchange):
        return TrainStepper(model, R, learning_rate=0.03, apply_diffuse_render_regularization=True, ray_selection=args.ray_selection, backward=args.backward,
                            deterministic=args.deterministic, fuse_optimizer=False if (args.no_fuse_optimizer or exchange == "dense") else None,
                            merge_bricks=False if exchange == "dense" and (world > 1 or args.dp_style_step) else None, exchange=exchange)

    stepper = make_stepper(args.exchange)
    multi_fwd, multi_fwd_error, leave_hard = None, None, False
    exchange_fallback = os.environ.get("RF_BENCH_FALLBACK_REASON")  # (the supervisor's: why earlier attempts were abandoned)
    supervised = bool(os.environ.get("RF_BENCH_WORKER")) and os.environ.get("RF_BENCH_LAST_ATTEMPT") != "1"
    torch.manual_seed(1234 + rank)  # every rank draws its own rays
    batches = dataset.image_batches(args.images)
    validation_steps = 0
    if world > 1:
        # VALIDATION, before anything is timed: the first iterations of the exchange with the replicas compared after each (the
        # owner-computes step does it itself on its first RF_OWNER_CHECK_STEPS = 3 iterations: the all-gathers left in flight across
        # the iteration boundary are first consumed by the second).  Should it raise on ANY rank -- a collective RCCL refuses,
        # diverged replicas -- all ranks agree on it through an all-reduce and leave this configuration together: under the
        # supervisor (top of this file) with EXIT_FALLBACK, so that the next, more conservative attempt starts in fresh processes;
        # unsupervised by falling back to the dense exchange on a freshly initialised grid right here.  A rank that HANGS instead is
        # the supervisor's business (its watchdog starts when every rank has written `ready`).  The line says which it was.
        from thr3ed_atom_amd.trainers import OWNER_CHECK_STEPS

        torch.distributed.barrier()
        torch.cuda.synchronize()
        _mark("ready")
        failed, reason = 0, None
        validation_steps = max(1, OWNER_CHECK_STEPS) if stepper.exchange == "owner" else 1
        try:
            for _ in range(validation_steps):
                stepper.step(dataset, next(batches))
            torch.cuda.synchronize()
            if stepper.exchange != "owner":
                rfdist.assert_replicas_identical(stepper.flat.flat_param, "dense exchange, first iteration")
        except Exception as exc:  # noqa: BLE001
            failed, reason = 1, f"{type(exc).__name__}: {exc}"
            if supervised:
                # said at once: the other ranks may already sit in the next iteration's collectives, where the all-reduce below never
                # meets them -- the supervisors see the marker, end every rank's worker and move on with the real reason
                _mark("fail", f"rank {rank}: {reason}")
        flag = torch.tensor([failed], device=dev, dtype=torch.int32)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
        if int(flag.item()):
            why = reason or "the validation steps failed on another rank"
            print(f"[rank {rank}] {stepper.exchange} exchange failed its validation ({why})", file=sys.stderr, flush=True)
            if supervised or stepper.exchange != "owner":
                if not failed:
                    _mark("fail", f"rank {rank}: {why}")
                torch.distributed.barrier()
                os._exit(EXIT_FALLBACK if supervised else 1)
            exchange_fallback = why
            stepper.flat.detach()
            del stepper
            grid = make_grid(dev, G, args.sh_degree, seed=42, storage=args.storage)
            model = rf.VolumetricModel(grid, rf.render_sh_voxel_grid, cfg, device=dev)
            stepper = make_stepper("dense")
            torch.manual_seed(1234 + rank)
            batches = dataset.image_batches(args.images)
            validation_steps = 0
        _mark("valid")
    owner = stepper.exchange == "owner"
    executor = stepper.fused and stepper.merged_bricks and args.ray_selection == "keyed" and not owner
    for _ in range(max(0, args.warmup - validation_steps)):  # (the validation steps above were the first warm-up steps)
        stepper.step(dataset, next(batches))
    # per-kernel HIP events on `--timed-steps` of the timed steps, recorded by the library between its own launches
    timer_stride = max(1, args.steps // max(1, args.timed_steps))
    timed_idx = [i for i in range(0, args.steps, timer_stride)] if args.timed_steps > 0 else []
    events = [ops.StepEvents() for _ in timed_idx] if executor else []
    legacy_timer = ops.KernelTimer(preallocate=14 * len(timed_idx)) if (timed_idx and not executor and not owner) else None
    phase_events = []  # owner-computes step: torch events around its phases on the timed steps
    counts = torch.zeros((max(1, len(timed_idx)), 2), dtype=torch.int64, device=dev)  # records emitted per render on the event steps
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = 0
    for i in range(args.steps):
        on = k < len(timed_idx) and i == timed_idx[k]
        if executor:
            stepper.step_events = events[k] if on else None
        elif owner:
            stepper.phase_events = phase_events if on else None
        else:
            ops.KERNEL_TIMER = legacy_timer if on else None
        stats = stepper.step(dataset, next(batches))
        if on:
            if executor:  # a 16-byte device-side copy; no sync
                ex = stepper._exec["tensors"]
                counts[k, 0].copy_(ex["pass0"]["offsets"][-1], non_blocking=True)
                counts[k, 1].copy_(ex["pass1"]["offsets"][-1], non_blocking=True)
            k += 1
    host_issue = time.perf_counter() - t0  # time the host needed to enqueue all steps (the launch queue back-pressures it)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    ops.KERNEL_TIMER = None
    stepper.step_events = None
    stepper.phase_events = None
    exchange_bytes = 0
    if owner and stepper.exchange_bytes:  # measured: record slices + offset tables + parameter chunks this rank sent, mean of the last steps
        exchange_bytes = int(np.mean(stepper.exchange_bytes))
    # ---- more windows of the same K steps, behind the official one (which is what the driver's command fixes: steps W .. W + K from
    # the initialisation).  The official window is ~13 ms of GPU time: box-to-box and run-to-run spread is the size of a "kept" win,
    # so the line also says what the following windows took.  (The field gets sparser as it trains: later windows are a little
    # lighter by the data -- they are reported beside `ms_per_step`, never instead of it.)
    window_ms = []
    for _ in range(max(0, args.windows)):
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        tw = time.perf_counter()
        for _i in range(args.steps):
            stepper.step(dataset, next(batches))
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        window_ms.append((time.perf_counter() - tw) / args.steps * 1e3)
    # ---- a second weak-scaling point: the same step on --second-point-rays rays per GPU.  The data-parallel exchange moves the MODEL
    # every iteration (205 MB of parameters per rank at N = 8) whatever the batch is, so the scaling ratio depends on the per-GPU
    # batch: DESIGN section 7 projects < 6 x at 16384 rays per GPU and > 6 x at 32768.  Every N prints both, the ratio is the driver's.
    second = None
    if args.second_point_rays > 0:
        # from a FRESHLY initialised grid, the same warm-up as the official window: the figure is comparable with `ms_per_step` (the
        # heaviest part of a run), not with the sparser field the windows above end on
        R2 = args.second_point_rays
        grid2 = make_grid(dev, G, args.sh_degree, seed=42, storage=args.storage)
        model2 = rf.VolumetricModel(grid2, rf.render_sh_voxel_grid, cfg, device=dev)
        keep_model, model = model, model2
        stepper2 = make_stepper(stepper.exchange if (world > 1 or args.dp_style_step) else args.exchange)
        model = keep_model
        stepper2.ray_batch_size = R2
        torch.manual_seed(4321 + rank)
        batches2 = dataset.image_batches(args.images)
        for _ in range(max(3, args.warmup)):
            stepper2.step(dataset, next(batches2))
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            stepper2.step(dataset, next(batches2))
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        second = time.perf_counter() - t2
        if world > 1:
            grid2.wait_for_parameters()
        stepper2.flat.detach()
        del stepper2, model2, grid2
    replicas_ok = None
    if world > 1:
        model.thre3d_repr.wait_for_parameters()
        replicas_ok = bool(rfdist.replicas_identical(stepper.flat.flat_param))  # every rank must hold the same parameters after the timed steps
        if not replicas_ok and supervised:  # (a collective result: every rank sees the same) -> the next attempt
            _mark("fail", f"rank {rank}: the replicas' parameters differ after the timed steps")
            torch.distributed.barrier()
            os._exit(EXIT_FALLBACK)
        te = torch.tensor([elapsed, second or 0.0] + window_ms, device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(te, op=torch.distributed.ReduceOp.MAX)
        elapsed, second, window_ms = float(te[0].item()), (float(te[1].item()) if second else None), [float(x) for x in te[2:].tolist()]
        _mark("done")
        if not owner:  # reduce-scatter + all-gather (or all-reduce) of the flat bucket: 2 (N-1)/N x bucket bytes sent per rank per step
            exchange_bytes = int(2 * (world - 1) / world * stepper.flat.flat_grad.numel() * 4)
        # ---- the forward legs on all N GPUs (fwd_render.multi_gpu / highres_render.multi_gpu), BEHIND the committed training
        # result and under a guard of their own: should a rank not come back from them, the line is printed without them.
        # Then every rank leaves the group together, BEFORE rank 0 starts its single-process reporting legs.
        def forward_legs_and_goodbye():
            if os.environ.get("RF_BENCH_INJECT_FORWARD_LEGS_HANG") == str(rank):  # (test hook: this rank never enters the legs' collectives)
                time.sleep(10**6)
            legs_ = multi_gpu_forward_legs(args, dev, rank, world, intr, bounds) if (args.render_frames > 0 or args.highres_frames > 0) else {}
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
            return legs_

        multi_fwd, multi_fwd_error = run_guarded(forward_legs_and_goodbye, float(os.environ.get("RF_BENCH_FORWARD_LEGS_TIMEOUT_S", "150")), dev)
        if multi_fwd_error is not None:
            print(f"[rank {rank}] multi-GPU forward legs: {multi_fwd_error}", file=sys.stderr, flush=True)
            leave_hard = True  # (a helper thread may still sit in a collective: no interpreter shutdown, no process-group destructor)
            if rank != 0:
                os._exit(0)
            args.cpu_rays = 0  # (the line, quickly: RCCL's own watchdog ends a process whose collective never completes)
    elif torch.distributed.is_initialized():  # --dp-style-step: the 1-rank group
        torch.distributed.destroy_process_group()
        rfdist.FORCE_COLLECTIVES = False

    if rank != 0:
        return
    ms_per_step = elapsed / args.steps * 1e3
    value = world * 2 * R * S * args.steps / elapsed

    # ---- per-kernel times of the timed region -------------------------------------------------------
    kernels = {}
    if executor and events:
        per = [e.elapsed_ms() for e in events]
        for name in per[0]:
            kernels[name] = {"avg_ms": float(np.mean([p[name] for p in per])), "launches": len(per)}
    elif owner and phase_events:
        # the owner step's marks on the compute stream: the span in front of mark i is named after it ("wait: ..." spans = the compute
        # stream blocked on communication = EXPOSED communication; the other spans = compute, incl. whatever a wait left over)
        for i in range(1, len(phase_events[0])):
            name = phase_events[0][i][0]
            kernels[name] = {"avg_ms": float(np.mean([e[i - 1][1].elapsed_time(e[i][1]) for e in phase_events])), "launches": len(phase_events)}
    elif legacy_timer is not None:
        for name, rec in legacy_timer.summary().items():
            kernels[name] = {"avg_ms": rec["avg_ms"], "launches": rec["launches"]}
    rec_counts = counts.cpu().numpy().astype(np.float64)
    rec_spec = float(rec_counts[:, 0].mean()) if executor and timed_idx else None
    rec_diff = float(rec_counts[:, 1].mean()) if executor and timed_idx else None
    # in-AABB samples of a batch (harness-side, un-jittered positions): density is gathered for these
    rays, pixels = stepper.select(dataset, next(batches))
    n_in = count_inside(rays.origins, rays.directions, S, grid.aabb)
    nparam = G**3 * C

    roofline = None
    if kernels and rec_spec is not None:
        fused_opt = stepper.fuse_optimizer
        # algorithmic bytes (SURVEY 8d per-unit figures) on the units each launch really processes:
        #   forward:  8 corners x C x 4 B per sample whose features are gathered (= the records), 8 x 16 B (the base record) for
        #             the other in-AABB samples (density only), 20 B of cache per CACHED sample (= per record) + the chunk masks, 48 B per ray
        #   emit:     20 B cache read + the record written (48 B specular, 32 B diffuse) per record
        #   bricks:   COMPULSORY HBM bytes only: the records read once + the optimizer traffic when it is fused (3 reads + 3 writes
        #             of 4 B per parameter; else the gradient bucket written once).  The scatter payload (8 corners x C channels per
        #             record) is on-chip work and is not credited as bandwidth.
        rec_b, rec_b_d = 4 * ops.expanded_record_floats(grid), 4 * ops.expanded_record_floats(grid, True)
        mask_b = R * ((S + 63) // 64) * 8
        alg = {
            "render_forward[spec,save]": rec_spec * 8 * C * 4 + max(n_in - rec_spec, 0) * 8 * 16 + rec_spec * 20 + mask_b + R * 48,
            "render_forward[diffuse,save]": n_in * 8 * 16 + rec_diff * 20 + mask_b + R * 48,
            "render_backward_emit_direct[spec]": rec_spec * (20 + rec_b) + mask_b,
            "render_backward_emit_direct[diffuse]": rec_diff * (20 + rec_b_d) + mask_b,
            "brick_accumulate": rec_spec * rec_b + rec_diff * rec_b_d + nparam * 4 * (6 if fused_opt else 1),
        }
        # paired launches (both renders / both adjoints of the iteration in one launch): the pair is priced on the sum of its halves'
        # bytes (the second render finds most of its base records on chip: the counter figure is what reaches HBM)
        for a_, b_, m_ in (("render_forward[spec,save]", "render_forward[diffuse,save]", "render_forward[spec+diffuse,save]"),
                           ("render_backward_emit_direct[spec]", "render_backward_emit_direct[diffuse]", "render_backward_emit_direct[spec+diffuse]")):
            if m_ in kernels:
                alg[m_] = alg.pop(a_) + alg.pop(b_)
        alg = {k: alg[k] for k in sorted(alg, key=lambda k: list(kernels).index(k))}
        names = {"brick_accumulate": f"brick_accumulate_adam[{spec}]" if fused_opt else f"brick_accumulate[{spec}]"}
        by_kernel = {}
        for kname, b in alg.items():
            ms = kernels[kname]["avg_ms"]
            counter = pmc.get(names.get(kname, kname), {}).get("hbm_bytes_per_launch")
            by_kernel[kname] = {
                "avg_launch_ms": ms,
                "algorithmic_bytes_processed": b,
                "frac_processed": frac(b, ms, f"{kname} (processed units)"),
                "counter_bytes_per_launch": counter,
                "frac_hbm": None if counter is None else frac(counter, ms, f"{kname} (counters)"),
            }
        dom = max(alg, key=lambda kname: kernels[kname]["avg_ms"])
        d = by_kernel[dom]
        traffic = d["counter_bytes_per_launch"]
        # `frac` = ALGORITHMIC bytes of the launch (the compulsory HBM bytes on the units it really processed, counted on the device in
        # this run) / this run's launch time / 8 TB/s: reproducible from the run alone.  The counter figure (fabric bytes of a separate
        # rocprofv3 --pmc run of this command, profiles/pmc_traffic.json, tied to the kernel source by hash) is `traffic` and
        # `frac_fabric_counters`: traffic well above the algorithmic bytes = wasted re-reads.
        roofline = {
            "kernel": names.get(dom, dom),
            "bound": "hbm",
            "achieved": d["algorithmic_bytes_processed"] / 1e9 / (d["avg_launch_ms"] / 1e3),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": d["frac_processed"],
            "frac_of_achievable": None if d["frac_processed"] is None else d["frac_processed"] * HBM_PEAK_GBS / HBM_ACHIEVABLE_GBS,
            "achievable_GBps": HBM_ACHIEVABLE_GBS,
            "frac_basis": "algorithmic (compulsory) HBM bytes of the launch on the units it really processed -- records read once + 24 B per parameter for the brick pass with Adam "
            "in its flush, counted on the device in this run -- / this run's HIP-event launch time / the 8 TB/s peak",
            "traffic": traffic,
            "traffic_over_algorithmic": None if traffic is None else traffic / d["algorithmic_bytes_processed"],
            "frac_fabric_counters": d["frac_hbm"],
            "frac_fabric_counters_basis": None if traffic is None else "fabric bytes from the PMC counters (2 x FETCH_SIZE + WRITE_SIZE sectors at the data fabric, Infinity-Cache hits included: an upper "
            "bound of the DRAM traffic; profiles/pmc_traffic.json, sha256-tied to the kernel source) / this run's launch time",
            "avg_launch_ms": d["avg_launch_ms"],
            "frac_processed": d["frac_processed"],
            "algorithmic_bytes_processed": d["algorithmic_bytes_processed"],
            "units_processed": {"records_specular": rec_spec, "records_diffuse": rec_diff, "in_aabb_samples": n_in, "nominal_samples": R * S, "parameters": nparam},
            "note": "brick pass: both renders' gradient records summed per brick ("
            + ("4 x 8 x 8 nodes, four 256-thread workgroups per CU" if getattr(stepper, "brick_size", 8) == rf.ops.BRICK_4X8X8 else "8^3 nodes, two 512-thread workgroups per CU")
            + ") in MFMA accumulators (no atomics)" + (", Adam applied in the flush (no gradient tensor in HBM)" if fused_opt else "")
            + "; bound by the latencies of a workgroup's serial phases: see DESIGN section 4",
            "brick_size": int(getattr(stepper, "brick_size", 8)),
            "traffic_source": pmc.get("_source"),
            "traffic_stale": bool(pmc_stale),
            "by_kernel": by_kernel,
        }

    if roofline is None and kernels and owner:
        # owner-computes data-parallel step: the rank's brick pass (all ranks' record lists for its own 1/N of the bricks, Adam in the
        # flush) priced on its compulsory HBM bytes: the records it consumed + 6 accesses x 4 B per OWN parameter
        bname = "brick pass + Adam (all halves)"
        recs = np.mean(np.array(stepper.owner_records, dtype=np.float64), axis=0) if stepper.owner_records else np.zeros(2)
        bytes_ = recs[0] * 4 * ops.expanded_record_floats(grid) + recs[1] * 4 * ops.expanded_record_floats(grid, True) + nparam / world * 24
        ms = sum(v["avg_ms"] for k, v in kernels.items() if k.startswith("brick pass"))
        roofline = {
            "kernel": bname, "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "achieved": bytes_ / 1e9 / (ms / 1e3), "frac": frac(bytes_, ms, bname),
            "traffic": None, "traffic_stale": bool(pmc_stale), "avg_launch_ms": ms,
            "frac_basis": "compulsory HBM bytes of this rank's brick pass (records consumed + 24 B per own parameter) / its event time",
            "units_processed": {"records_specular_consumed": float(recs[0]), "records_diffuse_consumed": float(recs[1]), "own_parameters": nparam / world},
            "by_kernel": {k: {"avg_launch_ms": v["avg_ms"]} for k, v in kernels.items()},
            "note": "owner-computes data-parallel step: phases timed with events on the compute stream (collectives included where they block it)",
        }
    if roofline is None and kernels:
        # the data-parallel-style step (or any non-merged step): per-launch HIP events of ops.KernelTimer; byte figures where the
        # counter table has the kernel (same kernels as the single-GPU step, plus the per-render brick passes and the separate optimizer)
        alias = {f"render_forward[{spec},save]": "render_forward[spec,save]", f"render_backward_emit_direct[{spec}]": "render_backward_emit_direct[spec]",
                 "brick_accumulate[diffuse]": "brick_accumulate[base]"}
        by_kernel = {}
        for kname, rec in kernels.items():
            counter = pmc.get(alias.get(kname, kname), {}).get("hbm_bytes_per_launch")
            if counter is not None and kname == "adam_step" and world > 1 and stepper.shard_optimizer:
                counter = counter / world  # ZeRO stage 1: a rank updates 1/N of the parameters (the table holds the full pass)
            by_kernel[kname] = {"avg_launch_ms": rec["avg_ms"], "counter_bytes_per_launch": counter,
                                "frac_hbm": None if counter is None else frac(counter, rec["avg_ms"], f"{kname} (counters)")}
        dom = max(kernels, key=lambda kname: kernels[kname]["avg_ms"] * kernels[kname]["launches"])
        d = by_kernel[dom]
        roofline = {
            "kernel": dom, "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "achieved": None if d["counter_bytes_per_launch"] is None else d["counter_bytes_per_launch"] / 1e9 / (d["avg_launch_ms"] / 1e3),
            "frac": d["frac_hbm"], "traffic": d["counter_bytes_per_launch"], "avg_launch_ms": d["avg_launch_ms"],
            "frac_basis": "HBM bytes from the PMC counters (profiles/pmc_traffic.json, single-GPU profile of the same kernels) / this run's launch time",
            "traffic_source": pmc.get("_source"), "traffic_stale": bool(pmc_stale), "by_kernel": by_kernel,
            "note": "data-parallel-style step: specular forward + emit + brick pass -> [gradient exchange of `rest` overlapped with] diffuse forward + emit + base-channel brick pass -> exchange of `base` -> (sharded) Adam",
        }

    train_256 = None
    if args.train256_steps > 0 and world == 1 and not args.dp_style_step:
        train_256 = train_leg_at_reference_final_stage(args, dev, dataset, bounds)

    baseline = None
    if args.cpu_rays > 0:
        baseline = cpu_baseline(grid, (rays.origins.cpu(), rays.directions.cpu()), pixels.cpu(), S, args.cpu_rays, args.cpu_fwd_rays, args.cpu_threads, n_full=args.cpu_full_rays)
        baseline["cfg1_full_frame"] = cfg1_leg(dev, baseline["cores"])

    line = {
        "metric": "ray-samples/sec (fwd+bwd) training step, 800x800 images @ 128^3 SH-2 ReLU field",
        "value": value,
        "unit": "ray-samples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "ms_per_step_windows": None if not window_ms else {
            "min": float(np.min(window_ms)), "median": float(np.median(window_ms)), "max": float(np.max(window_ms)), "windows": len(window_ms), "steps_per_window": args.steps,
            "first_step": args.warmup + args.steps,
            "note": "further windows of the same length timed back to back BEHIND the official one (ms_per_step = steps warmup .. warmup + steps from the initialisation, "
                    "what the driver's command fixes); the field gets sparser as it trains, so later windows are a little lighter by the data"},
        "second_weak_scaling_point": None if not second else {
            "rays_per_gpu_per_step": args.second_point_rays, "ms_per_step": second / args.steps * 1e3, "value": world * 2 * args.second_point_rays * S * args.steps / second,
            "unit": "ray-samples/s", "steps": args.steps, "first_step": max(3, args.warmup), "from": "a freshly initialised grid (like the official window)",
            "note": "the same iteration on a larger per-GPU batch, from the initialisation with the official window's warm-up (max over ranks, barriers on both sides): the exchange "
                    "moves the model, not the batch, so the N-GPU ratio grows with the per-GPU batch (DESIGN section 7)"},
        "host_issue_ms_per_step": host_issue / args.steps * 1e3,
        "kernel_timer_steps": len(timed_idx),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"configs[2]: train step on {G}^3 SH-degree-{args.sh_degree} ReLU field (U(-1,1) init), {args.images} synthetic {H}x{W} images, "
            f"{R} random distinct pixels/GPU/step out of all {args.images}x{H}x{W} ({args.ray_selection} selection), {S} jittered samples/ray, specular+diffuse fwd+bwd, L1+L1, "
            + ("Adam fused into the brick flush" if stepper.fuse_optimizer else "fused Adam kernel")
            + ((", owner-computes exchange over RCCL: all-gather of offset tables -> all-to-all of gradient-record slices -> merged brick pass + Adam on the rank's own x-slabs -> parameter all-gather"
                if owner else ", gradient exchange over RCCL: reduce-scatter -> Adam on 1/N of the grid per rank -> all-gather") if (world > 1 or args.dp_style_step) else ""),
            "rays_per_gpu_per_step": R,
            "samples_per_ray": S,
            "renders_per_step": 2,
            "parallelism": f"dp{world}" + (("+owner-computes" if owner else ("+zero1" if stepper.shard_optimizer else "")) if (world > 1 or args.dp_style_step) else ""),
            "exchange": stepper.exchange if (world > 1 or args.dp_style_step) else None,
            "grid_storage": args.storage,
            "ray_selection": args.ray_selection,
            "backward": stepper.backward,
            "merged_brick_pass": bool(stepper.merged_bricks),
            "fused_optimizer": bool(stepper.fuse_optimizer),
            "one_call_per_step": bool(executor),
        },
        "distributed": {"backend": backend, "world_size_seen_by_collectives": rccl_world, "exchange_bytes_sent_per_rank_per_step": exchange_bytes,
                        "exchange": stepper.exchange if (world > 1 or args.dp_style_step) else None,
                        "exchange_fallback_reason": exchange_fallback,  # not None: the owner-computes step failed and the run fell back to the dense exchange
                        "replicas_bit_identical": replicas_ok,  # parameter checksums of all ranks after the timed steps (None on one GPU)
                        "exposed_communication_ms_per_step": ({k: v["avg_ms"] for k, v in kernels.items() if k.startswith("wait:")} if owner else None),
                        "owner_halves": (stepper._owner or {}).get("H") if owner else None,
                        "owner_workgroups_per_brick": (stepper._owner or {}).get("parts") if owner else None},
        "rays_per_s": world * R * args.steps / elapsed,  # rays of the batch per second (each is rendered twice per step: renders_per_s = 2 x this)
        "final_specular_psnr": stats.psnr()["specular_psnr"],
        "inside_fraction": n_in / (R * S),
        "kernels": kernels,
        "roofline": roofline,
        "cpu_baseline": baseline,
        "fwd_render": fwd_render,
        "highres_render": highres,
        "strict_dropin": dropin,
        "train_256": train_256,
        "roofline_model_errors": MODEL_ERRORS,
    }
    if world > 1:
        # fwd_render / highres_render of a multi-GPU run: rank 0's single-GPU figures + what all N GPUs did together
        for key, asked in (("fwd_render", args.render_frames > 0), ("highres_render", args.highres_frames > 0)):
            leg = (multi_fwd or {}).get(key)
            if not asked or (leg is None and multi_fwd_error is None):
                continue
            line[key] = dict(line[key] or {}, multi_gpu=leg if leg is not None else {"error": multi_fwd_error})
    print(json.dumps(line), flush=True)
    if leave_hard:
        os._exit(0)


if __name__ == "__main__":
    main()
