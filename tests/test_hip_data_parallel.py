"""Two REAL processes training data-parallel on one GPU: the full fused HIP step in both exchange schemes -- owner-computes
(offset tables all-gathered, record slices exchanged, merged brick pass + Adam on the rank's own x-slabs, parameters all-gathered)
and dense (reduce-scatter of the gradient bucket, sharded Adam, all-gather) --, strong and weak scaling modes, with the exchange
going through torch.distributed -- gloo here, because RCCL refuses two ranks on one device; the RCCL calls themselves are exercised
by test_hip_training.py::test_data_parallel_step_through_rccl_single_rank and multi-GPU runs are the driver's."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import thr3ed_atom_amd as rf
from thr3ed_atom_amd import distributed as rfdist
from thr3ed_atom_amd.trainers import PosedImagesInMemory, TrainStepper
from tests.helpers import hash_uniform, hotdog_like_camera

pytestmark = pytest.mark.gpu

G, DEG, S, R = 16, 2, 32, 512
F = 3 * (DEG + 1) ** 2
GRID = {"G": G}  # (the 4-rank test trains a 32^3 grid: 4 x-slabs of bricks, one per owner, so that the middle owners' slices are exercised)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _setup(dev, jitter=False):
    cam = hotdog_like_camera()
    images = torch.from_numpy(hash_uniform((4, 3, 24, 24), 77, 0.0, 1.0)).to(dev)
    cams = [rf.pose_spherical(40.0 * k, -30.0, cam["radius"]) for k in range(4)]
    poses = torch.stack([torch.cat([c.rotation, c.translation.reshape(3, 1)], dim=1) for c in cams]).to(dev)
    data = PosedImagesInMemory(images, poses, rf.CameraIntrinsics(24, 24, 33.0), rf.CameraBounds(cam["near"], cam["far"]))
    G = GRID["G"]
    dims = GRID.get("dims") or (G, G, G)  # (a non-cubic grid: partial bricks along y and z)
    F = 3 * (GRID.get("deg", DEG) + 1) ** 2
    grid = rf.VoxelGrid(
        torch.from_numpy(hash_uniform((*dims, 1), 901)).to(dev), torch.from_numpy(hash_uniform((*dims, F), 900 + F)).to(dev),
        rf.VoxelSize(3.0 / dims[0], 3.0 / dims[1], 3.0 / dims[2]), density_preactivation=torch.nn.Identity(), density_postactivation=torch.nn.ReLU(),
        expected_density_scale=100.0 / 3.0, tunable=True, storage="split",
    )
    cfg = rf.SHVoxGridRenderConfig(S, data.camera_bounds, perturb_sampled_points=jitter, white_bkgd=True)
    return data, rf.VolumetricModel(grid, rf.render_sh_voxel_grid, cfg, device=dev)


def _train(stepper, data, steps=3):
    torch.manual_seed(5)  # the CPU generator picks the rays: the same state on every rank
    for _ in range(steps):
        stepper.step(data, torch.arange(4))
    torch.cuda.synchronize()
    return stepper.flat.flat_param.clone()


def _worker(rank, world, port, result_dir, exchange, shard_optimizer, jitter, grid_size=16, halves=1, brick_parts=0, dims=None, deg=DEG):
    GRID["G"] = grid_size
    GRID["dims"], GRID["deg"] = dims, deg
    if brick_parts:
        import thr3ed_atom_amd.trainers as trainers_module

        trainers_module.OWNER_BRICK_PARTS = brick_parts  # several workgroups per owned brick (rf_brick_accumulate_adam_split)
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        # strong scaling: the ranks split ONE global batch -> the run must equal the single-process run
        # (with the keyed jitter on, rank r must draw the jitter of rays lo..hi of the global batch: KeyedJitter.first_ray)
        data, model = _setup(dev, jitter)
        # (backward="binned": what "auto" picks from SH degree 2 on; the degree-0 case asks for it)
        stepper = TrainStepper(model, R, learning_rate=0.03, global_batch=True, shard_optimizer=shard_optimizer, exchange=exchange, backward="binned" if deg == 0 else "auto")
        assert stepper.exchange == exchange
        dp = _train(stepper, data)
        if exchange == "owner":
            assert stepper.exchange_bytes and stepper.exchange_bytes[-1] > 0 and stepper.owner_records[-1][0] > 0
            assert stepper._owner["H"] == halves, (stepper._owner["H"], halves)  # interleaved ownership: the pipelined step
            assert stepper._owner["parts"] == (brick_parts or (2 if world >= 4 else 1))
        gathered = [torch.empty_like(dp) for _ in range(world)]
        dist.all_gather(gathered, dp)
        assert all(torch.equal(gathered[0], t) for t in gathered), "replicas diverged"
        data, model = _setup(dev, jitter)
        single = _train(TrainStepper(model, R, learning_rate=0.03, data_parallel=False, backward="binned" if deg == 0 else "auto"), data)
        err = float((dp - single).abs().max())
        moved = float((single - torch.cat([t.reshape(-1) for t in model.thre3d_repr.kernel_tensors() if t is not None]).detach()).abs().max())
        assert err <= 2e-4, f"data-parallel run differs from the single-process run by {err}"
        # weak scaling: every rank draws its own batch; replicas must still agree
        data, model = _setup(dev, jitter)
        stepper = TrainStepper(model, R // 2, learning_rate=0.03, shard_optimizer=shard_optimizer, exchange=exchange, backward="binned" if deg == 0 else "auto")
        torch.manual_seed(11 + rank)
        for _ in range(2):
            stepper.step(data, torch.arange(4))
        torch.cuda.synchronize()
        mine = stepper.flat.flat_param.clone()
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        assert all(torch.equal(gathered[0], t) for t in gathered), "replicas diverged (weak scaling)"
        open(os.path.join(result_dir, f"ok{rank}"), "w").write(f"{err} {moved}")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange,shard_optimizer,jitter,grid_size,halves", [
    ("owner", True, False, 16, 1), ("owner", True, True, 16, 1),
    ("owner", True, True, 32, 2),  # 4 x-slabs of bricks: rank r owns slabs r and r + 2 -- the interleaved ownership of the pipelined step
    ("owner", True, False, 64, 2),  # 8 x-slabs: two neighbouring slabs per rank and half
    ("owner", True, True, 128, 2),  # the BASELINE grid: 16 x-slabs, four per rank and half, 235 MB of parameters gathered in two halves
    ("owner", True, True, 32, -2),  # ... with TWO workgroups per owned brick (the source ranks' lists dealt out, partial images merged)
    ("dense", True, False, 16, 1), ("dense", False, True, 16, 1)])
def test_two_processes_train_data_parallel_on_one_gpu(tmp_path, exchange, shard_optimizer, jitter, grid_size, halves):
    assert torch.cuda.is_available()
    world = 2
    parts = 2 if halves < 0 else 0
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), exchange, shard_optimizer, jitter, grid_size, abs(halves), parts), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == [f"ok{r}" for r in range(world)]


@pytest.mark.parametrize("deg,dims", [(0, (32, 20, 28)), (2, (32, 12, 44))])
def test_two_processes_owner_computes_on_odd_grids(tmp_path, deg, dims):
    """SH degree 0 (one kind of list, the base-channel kernel) and non-cubic grids whose y / z extents are not whole bricks, interleaved
    halves and two workgroups per owned brick: equal to the single-process run."""
    assert torch.cuda.is_available()
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "owner", True, True, dims[0], 2, 2, dims, deg), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == [f"ok{r}" for r in range(world)]


@pytest.mark.parametrize("grid_size,halves", [(32, 1), (64, 2)])
def test_four_processes_owner_computes_on_one_gpu(tmp_path, grid_size, halves):
    """Four ranks.  32^3 grid: one x-slab of bricks each -- the two middle owners receive slices that start with the x-flagged
    records of the slab below them and end before the slab above (the general case of the slice formula).  64^3 grid: eight
    slabs owned in two interleaved halves (rank r: slabs r and r + 4): per half, record exchange -> brick pass + Adam -> all-gather of
    that half of the parameters, the pipelined step.  Keyed jitter on; both must equal the single-process run."""
    assert torch.cuda.is_available()
    world = 4
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "owner", True, True, grid_size, halves), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == [f"ok{r}" for r in range(world)]


@pytest.mark.parametrize("grid_size,halves", [(64, 1), (128, 2)])
def test_eight_processes_owner_computes_on_one_gpu(tmp_path, grid_size, halves):
    """EIGHT ranks -- the world size of BASELINE configs[3], where the owner step is at its limits: 8 full-width + 8 base-channel lists per
    brick pass (the most a launch takes), two workgroups per owned brick, seven foreign slices per receive buffer.  64^3: one x-slab of
    bricks per rank.  128^3 (the bench grid): sixteen slabs owned in two interleaved halves, the pipelined step the first attempt of
    ``bench.py --gpus 8`` runs.  Both must equal the single-process run and keep the replicas bit-identical."""
    assert torch.cuda.is_available()
    world = 8
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "owner", True, True, grid_size, halves), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == [f"ok{r}" for r in range(world)]


def _trainer_worker(rank, world, port, result_dir):
    """the whole trainer (two stages, a checkpoint every 7 iterations but a synchronising summary only every 100) under data parallelism:
    checkpoints are written while parameter all-gathers of the owner-computes step may still be in flight -- state_dict() has to wait
    for them -- and must hold exactly what every rank holds"""
    import copy
    from pathlib import Path

    from thr3ed_atom_amd.trainers import train_sh_vox_grid_vol_mod_with_posed_images

    GRID["G"] = 32
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        data, model = _setup(dev, jitter=True)
        torch.manual_seed(3)  # (global_batch: all ranks draw the same batches and take their own slices)
        out = Path(result_dir) / f"rank{rank}"
        trained = train_sh_vox_grid_vol_mod_with_posed_images(
            model, data, output_dir=out, image_batch_cache_size=4, ray_batch_size=R, num_stages=2, num_iterations_per_stage=15, learning_rate=0.03,
            lr_decay_steps_per_stage=10, save_freq=7, test_freq=10**6, summary_freq=100, log=lambda *_: None, storage="split", global_batch=True)
        grid = trained.thre3d_repr
        copy.deepcopy(grid)  # (nothing un-copyable may be left on the module by the data-parallel step)
        final = torch.cat([grid.densities.detach().reshape(-1), grid.features.detach().reshape(-1)])
        gathered = [torch.empty_like(final) for _ in range(world)]
        dist.all_gather(gathered, final)
        assert all(torch.equal(gathered[0], t) for t in gathered), "replicas diverged in the trainer"
        if rank == 0:
            names = sorted(p.name for p in (out / "saved_models").iterdir())
            assert "model_final.pth" in names and any(n.startswith("model_stage_2_iter_28") for n in names), names
            # the last checkpoint of stage 1 == the grid the stage ended with == what an un-distributed reader would load
            ck = torch.load(out / "saved_models" / "model_final.pth", weights_only=False)
            sd = ck["thre3d_repr"]["state_dict"]
            assert torch.equal(sd["_densities"].to(dev), grid.densities.detach()) and torch.equal(sd["_features"].to(dev), grid.features.detach())
            mid = torch.load(out / "saved_models" / "model_stage_1_iter_14.pth", weights_only=False)["thre3d_repr"]["state_dict"]
            assert tuple(mid["_densities"].shape) == (16, 16, 16, 1) and bool(torch.isfinite(mid["_features"]).all())
        open(os.path.join(result_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_trainer_under_data_parallelism_writes_consistent_checkpoints(tmp_path):
    assert torch.cuda.is_available()
    world = 2
    mp.spawn(_trainer_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert sorted(n for n in os.listdir(tmp_path) if n.startswith("ok")) == [f"ok{r}" for r in range(world)]


# ---- the multi-rank FORWARD render: VolumetricModel.render(data_parallel=True) (counterpart of modules/volumetric_model.py:143-172) -----
def _frame_worker(rank, world, port, result_dir):
    """Every rank renders its ``shard_range`` of an odd frame (47 x 61 pixels: the cuts fall in the middle of a pixel row AND in the
    middle of an 8 x 8 tile row) against its replica of the grid and the [n, 6] results are all-gathered: the frame every rank ends up
    with must equal the single-process frame BIT FOR BIT -- a pixel does not depend on how a frame is cut into calls, in either
    frame kernel -- on both branches of ``render``: the one-launch branch (rays and keyed jitter generated in-kernel; ray packets and
    the per-ray kernel) and the chunk loop (ragged chunks inside the shard), device and host outputs, with the occupancy mask."""
    from thr3ed_atom_amd.voxels import VoxelGrid
    from tests.helpers import sparse_scene_grid

    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cam = hotdog_like_camera()
    dims = (20, 16, 24)
    dens, feat = sparse_scene_grid(dims, 27, 5)
    grid = VoxelGrid(dens.to(dev), feat.to(dev), rf.VoxelSize(*(3.0 / d for d in dims)), density_preactivation=torch.nn.Identity(),
                     density_postactivation=torch.nn.ReLU(), expected_density_scale=100.0 / 3.0, tunable=False, storage="split")
    intr = rf.CameraIntrinsics(47, 61, 70.0)
    pose = rf.pose_spherical(35.0, -25.0, cam["radius"])
    bounds = rf.CameraBounds(cam["near"], cam["far"])
    cfg = rf.SHVoxGridRenderConfig(40, bounds, perturb_sampled_points=True, white_bkgd=True)
    model = rf.VolumetricModel(grid, rf.render_sh_voxel_grid, cfg, device=dev)

    def same(a, b):  # bit for bit, every output (disparity is NaN on rays that hit nothing, accumulate.py:85-88: NaN == NaN here)
        pairs = [(a.colour, b.colour), (a.depth, b.depth)] + [(a.extra[k], b.extra[k]) for k in sorted(a.extra)]
        return sorted(a.extra) == sorted(b.extra) and all(torch.equal(torch.nan_to_num(x.to(dev), nan=-7.0), torch.nan_to_num(y.to(dev), nan=-7.0)) for x, y in pairs)

    # the single-process frames first (no process group yet: data_parallel=True is then the plain render)
    singles = {}
    for tiles in ("1", "0"):
        os.environ["RF_FRAME_TILES"] = tiles
        for mask in (False, True):
            torch.manual_seed(4)  # (the jitter key comes from torch's CPU generator: the same state on every rank)
            singles[tiles, mask] = model.render(pose, intr, data_parallel=True, use_occupancy_mask=mask)
    assert same(singles["1", False], singles["1", True]) and same(singles["0", False], singles["0", True])  # the mask is exact
    torch.manual_seed(4)
    single_chunked = model.render(pose, intr, parallel_rays_chunk_size=300, perturb_sampled_points=False, consume_reference_rng=True)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        lo, hi = rfdist.shard_range(47 * 61)
        assert rank == 0 or (lo % 61 != 0 and (lo // 61) % 8 != 0)  # the cut: inside a pixel row, inside a tile row
        for tiles in ("1", "0"):
            os.environ["RF_FRAME_TILES"] = tiles
            for mask in (False, True):
                torch.manual_seed(4)
                frame = model.render(pose, intr, data_parallel=True, use_occupancy_mask=mask)
                assert frame.colour.shape == (47, 61, 3) and frame.depth.shape == (47, 61, 1)
                assert same(frame, singles[tiles, mask]), f"sharded one-launch frame differs (tiles={tiles}, mask={mask})"
        # the chunk loop (asked for by consume_reference_rng): ragged chunks of 300 rays inside every shard; device and host outputs
        for gpu_render in (True, False):
            torch.manual_seed(4)
            frame = model.render(pose, intr, parallel_rays_chunk_size=300, data_parallel=True, gpu_render=gpu_render, perturb_sampled_points=False, consume_reference_rng=True)
            assert frame.colour.device.type == ("cuda" if gpu_render else "cpu")
            assert same(frame, single_chunked), "sharded chunk-loop frame differs"
        # every rank holds the same frame
        mine = frame.colour.to(dev).contiguous()
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        assert all(torch.equal(every[0], e) for e in every)
        open(os.path.join(result_dir, f"ok{rank}"), "w").write("ok")
    finally:
        os.environ.pop("RF_FRAME_TILES", None)
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_frame_render_equals_the_single_process_frame(tmp_path, world):
    assert torch.cuda.is_available()
    mp.spawn(_frame_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == [f"ok{r}" for r in range(world)]


# ---- bench.py --gpus N: the supervised multi-GPU run (top of bench.py) --------------------------------------------------------------
def _bench_two_ranks(extra_env, timeout=900, render_frames=0):
    """``python bench.py --gpus 2`` the way the driver launches N > 1 -- it re-executes itself as two ranks under torch.distributed.run,
    every rank a supervisor with the benchmark in a child process -- on ONE GPU over gloo (RF_SINGLE_DEVICE / RF_DIST_BACKEND), small
    workload.  Returns (completed process, the JSON lines it printed)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root, RF_SINGLE_DEVICE="1", RF_DIST_BACKEND="gloo", RF_BENCH_VALIDATE_TIMEOUT_S="20", **extra_env)
    for name in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "RF_BENCH_WORKER", "RF_BENCH_RUN_DIR"):
        env.pop(name, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "3", "--windows", "1", "--second-point-rays", "8192", "--grid", "64", "--rays", "4096",
           "--samples", "64", "--image-size", "200", "--cpu-rays", "0", "--render-frames", str(render_frames), "--highres-frames", "0", "--dropin-steps", "0"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=timeout)
    return r, [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_two_ranks_supervised(hip_device):
    """The undisturbed run: attempt 0 (owner-computes, interleaved halves, all-gathers in flight across the iteration boundary)
    validates itself on its first three iterations and is the configuration that gets timed; the world size in the line is counted
    by a collective."""
    r, lines = _bench_two_ranks({}, render_frames=2)
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    d = lines[0]["distributed"]
    assert lines[0]["n_gpus"] == 2 and d["world_size_seen_by_collectives"] == 2 and d["exchange"] == "owner" and d["owner_halves"] == 2
    assert d["exchange_fallback_reason"] is None and d["replicas_bit_identical"] is True
    assert lines[0]["ms_per_step_windows"]["windows"] == 1 and lines[0]["second_weak_scaling_point"]["rays_per_gpu_per_step"] == 8192
    # the forward legs on BOTH ranks (north_star: forward throughput "at 1/2/4/8 GPUs"): frame-parallel throughput and the sharded
    # frame's latency, the latter checked against the single-GPU frame inside the run
    m = lines[0]["fwd_render"]["multi_gpu"]
    assert m["n_gpus"] == 2 and m["frame_parallel"]["ray_samples_per_s"] > 0 and m["sharded_frame"]["ms_per_frame"] > 0
    assert m["sharded_frame"]["equals_single_gpu_frame_bit_for_bit"] is True


def test_bench_two_ranks_forward_legs_cannot_cost_the_line(hip_device):
    """A rank that never comes back from the multi-GPU forward legs (they run behind the committed training result, under a guard of their
    own): the line is printed without them and says so."""
    r, lines = _bench_two_ranks({"RF_BENCH_INJECT_FORWARD_LEGS_HANG": "1", "RF_BENCH_FORWARD_LEGS_TIMEOUT_S": "20"}, render_frames=1)
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    assert lines[0]["n_gpus"] == 2 and lines[0]["distributed"]["replicas_bit_identical"] is True and lines[0]["value"] > 0
    assert "not finished" in lines[0]["fwd_render"]["multi_gpu"]["error"]


@pytest.mark.parametrize("inject", ["hang", "failure", "death"])
def test_bench_two_ranks_watchdog_falls_back(hip_device, inject):
    """A rank that HANGS in its second iteration (the first that consumes all-gathers left in flight), one that RAISES in its first,
    one whose process DIES: the watchdog / the agreed failure flag / the exit code end attempt 0 on every rank, and the next attempt --
    owner-computes in its conservative configuration, fresh processes, a rendezvous of its own -- yields the line, labelled with why."""
    hook = {"hang": {"RF_OWNER_INJECT_HANG": "1:2"}, "failure": {"RF_OWNER_INJECT_FAILURE": "1"}, "death": {"RF_OWNER_INJECT_HANG": "1:2", "RF_OWNER_INJECT_DEATH": "1"}}[inject]
    r, lines = _bench_two_ranks(hook)
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    d = lines[0]["distributed"]
    assert lines[0]["n_gpus"] == 2 and d["exchange"] == "owner" and d["owner_halves"] == 1 and d["replicas_bit_identical"] is True
    why = d["exchange_fallback_reason"]
    assert why and "attempt 0" in why, why
    assert {"hang": "a hang", "failure": "injected failure", "death": "exited with code"}[inject] in why, why
