#!/usr/bin/env python
"""Headline benchmark of the MI355X ReLU-Fields render path.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json: "ray-samples/sec (fwd+bwd) ... 800x800 @ 128^3 grid", configs[2]/[3]):
one STEP = one training iteration of the posed-image trainer on a 128^3 SH-degree-2 ReLU field --
random 16384-ray batch out of 8 synthetic 800x800 images (randperm + ray generation), specular render
fwd, diffuse render fwd, L1 + L1, backward of both, [gradient all-reduce over RCCL when N > 1], Adam --
with 256 stratified (jittered) samples per ray.  ``value`` = nominal ray-samples/s over the whole job
= N * 2 renders * 16384 rays * 256 samples * K / time, inputs resident in HBM, nothing skipped.
Weak scaling: every rank draws its own 16384-ray batch.

Also reported in the same JSON line: the forward-only full-frame render of configs[1]
(``fwd_render``), the HBM roofline of the dominant kernel (``roofline``) and the oracle timed on the
host cores (``cpu_baseline``).  The oracle is used ONLY in that last leg.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import thr3ed_atom_amd as rf  # noqa: E402
from thr3ed_atom_amd import distributed as rfdist  # noqa: E402
from thr3ed_atom_amd import ops  # noqa: E402
from thr3ed_atom_amd.trainers import PosedImagesInMemory, TrainStepper  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
NEAR = float(np.float32(2.0) * 0.9)  # hotdog-like bounds, SURVEY.md 8d
FAR = float(np.float32(6.0) * 1.1)
RADIUS = 4.0311
WORLD = 3.0


def make_grid(dev, G, sh_degree, seed, sparse=False, storage="reference"):
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
        expected_density_scale=rf.compute_expected_density_scale_for_relu_field_grid((WORLD,) * 3),
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


def cpu_baseline(grid, rays_cpu, pixels_cpu, num_samples, n_rays, threads=0):
    """The oracle's float32 CPU path (same ATen ops the reference calls) on a bounded sample of the same
    workload: fwd+bwd of the specular and the diffuse render of ``n_rays`` rays, all host cores."""
    from oracle import relu_field_oracle as orc  # checker / baseline only

    # measured on the MI355X host (256 logical cores): 16 threads 4.2e5, 64 threads 3.8e5, 256 threads 0.5e5
    # ray-samples/s -- the ATen ops of this path do not scale past a few tens of threads, so the default is
    # capped at 16 (the count actually used is reported as "cores")
    cores = threads if threads > 0 else min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    dens = grid.densities.detach().cpu().clone().requires_grad_(True)
    feat = grid.features.detach().cpu().clone().requires_grad_(True)
    aabb = tuple(tuple(r) for r in grid.aabb)
    rho = grid.expected_density_scale

    def step(n):
        o, d, px = rays_cpu[0][:n], rays_cpu[1][:n], pixels_cpu[:n]
        dens.grad = feat.grad = None
        total = 0.0
        for diffuse in (False, True):
            t_rand = torch.rand(n, num_samples)
            out = orc.render(
                dens, feat, o, d, aabb, NEAR, FAR, num_samples, rho, "relu", white_bkgd=True,
                render_diffuse=diffuse, t_rand=t_rand, interp="aten",
            )
            total = total + torch.nn.functional.l1_loss(out["colour"], px)
        total.backward()

    step(min(256, n_rays))  # warm-up (thread pool, page-in)
    reps = 1
    t0 = time.perf_counter()
    for _ in range(reps):
        step(n_rays)
    dt = (time.perf_counter() - t0) / reps
    return {
        "value": 2 * n_rays * num_samples / dt,
        "unit": "ray-samples/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{reps} training-step cores (specular+diffuse fwd+bwd, no optimiser) of {n_rays} rays x {num_samples} samples "
        f"on the same 128^3 SH-2 grid, oracle with interp='aten' (F.grid_sample), torch {torch.__version__} CPU fp32; {dt:.2f} s/step",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50, help="untimed steps first (clocks and caches take ~50 steps to settle)")
    ap.add_argument("--grid", type=int, default=128)
    ap.add_argument("--sh-degree", type=int, default=2)
    ap.add_argument("--rays", type=int, default=16384, help="ray batch per GPU (reference CLI default)")
    ap.add_argument("--samples", type=int, default=256)
    ap.add_argument("--image-size", type=int, default=800)
    ap.add_argument("--images", type=int, default=8)
    ap.add_argument("--render-frames", type=int, default=5, help="full-frame forward renders timed for fwd_render (0 = skip)")
    ap.add_argument("--highres-frames", type=int, default=5, help="frames timed for the 256^3 / 512-sample configs[4] render (0 = skip)")
    ap.add_argument("--cpu-rays", type=int, default=2048, help="rays of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--storage", choices=["split", "bricked", "reference"], default="split",
                    help="HBM layout of the grid: split = MI355X-native (what the trainer uses), reference = the reference's two tensors")
    ap.add_argument("--ray-selection", choices=["keyed", "randperm"], default="keyed",
                    help="how a step picks its 16384 random pixels: keyed = fused keyed-permutation kernel (trainer default), "
                    "randperm = torch.randperm over all 5.12 M pixels like the reference")
    ap.add_argument("--backward", choices=["auto", "atomic", "binned"], default="auto",
                    help="specular gradient scatter of the train step: float32 atomics, or records binned by brick + atomic-free "
                    "LDS accumulation (auto = binned where supported and measured faster)")
    ap.add_argument("--deterministic", action="store_true", help="binned backward: stable radix sort instead of the counting sort")
    ap.add_argument("--timed-steps", type=int, default=5, help="how many of the --steps record per-kernel HIP events")
    ap.add_argument("--no-kernel-timer", action="store_true", help="do not record per-kernel HIP events in the timed region (no roofline object)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="torch CPU threads for cpu_baseline (0 = all host cores)")
    args = ap.parse_args()

    rank, local_rank, world = rfdist.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the render path has no CPU fallback)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    H = W = args.image_size
    focal = 1111.111 * (W / 800.0)
    intr = rf.CameraIntrinsics(H, W, focal)
    S, R, G = args.samples, args.rays, args.grid
    bounds = rf.CameraBounds(NEAR, FAR)

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
        model.render(pose, intr)  # warm-up
        timer = ops.KernelTimer()
        # every frame is timed on its own (sync before and after) and the MEDIAN is reported: the ROCm runtime now and then
        # stalls for tens of ms (observed with many outstanding timing events), which would swamp a 4 ms frame
        frame_s = []
        for i in range(args.render_frames):
            ops.KERNEL_TIMER = timer if i == 0 else None  # per-kernel events on one frame only
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.render(pose, intr)
            torch.cuda.synchronize()
            frame_s.append(time.perf_counter() - t0)
        dt = float(np.median(frame_s))
        ops.KERNEL_TIMER = None
        rays = rf.flatten_rays(rf.cast_rays(intr, pose, dev))
        n_in = count_inside(rays.origins, rays.directions, S, grid.aabb)
        ksum = timer.summary()
        kname = f"render_forward[sh{args.sh_degree}]"
        kms = ksum[kname]["total_ms"]  # the one frame with events
        del timer
        alg = n_in * 8 * (3 * (args.sh_degree + 1) ** 2 + 1) * 4 + H * W * 48
        fwd_render = {
            "workload": f"{G}^3 SH-{args.sh_degree} ReLU field, {H}x{W}, {S} samples/ray, jittered, VolumetricModel.render (32768-ray chunks)",
            "ms_per_frame": dt * 1e3,
            "ray_samples_per_s": H * W * S / dt,
            "rays_per_s": H * W / dt,
            "kernel_ms_per_frame": kms,
            "inside_fraction": n_in / (H * W * S),
            "algorithmic_GB_per_frame": alg / 1e9,
            "effective_GBps_kernel": alg / 1e9 / (kms / 1e3),
            "frac_of_hbm_peak": alg / 1e9 / (kms / 1e3) / HBM_PEAK_GBS,
        }

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
        times = {}
        for use in (False, True):
            hmodel.render(pose, intr, use_occupancy_mask=use)  # warm-up
            torch.cuda.synchronize()
            frame_s = []
            for _ in range(args.highres_frames):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                hmodel.render(pose, intr, use_occupancy_mask=use)
                torch.cuda.synchronize()
                frame_s.append(time.perf_counter() - t0)
            times[use] = float(np.median(frame_s))  # median of per-frame times (see fwd_render)
        a = hmodel.render(pose, intr, use_occupancy_mask=False, perturb_sampled_points=False)
        b = hmodel.render(pose, intr, use_occupancy_mask=True, perturb_sampled_points=False)
        occ_bits = int(sum(bin(w & 0xFFFFFFFF).count("1") for w in hg.occupancy.cpu().tolist()))
        highres = {
            "workload": f"configs[4]: 256^3 SH-{args.sh_degree} sparse ReLU field, {H}x{W}, 512 jittered samples/ray, VolumetricModel.render",
            "ms_per_frame_no_mask": times[False] * 1e3,
            "ms_per_frame_occupancy_mask": times[True] * 1e3,
            "ray_samples_per_s_occupancy_mask": H * W * 512 / times[True],
            "occupied_cell_fraction": occ_bits / float(257**3),
            "mask_build_ms": t_build * 1e3,
            "mask_is_exact": bool(torch.equal(a.colour, b.colour) and torch.equal(a.depth, b.depth)),
        }
        del hmodel, hg, a, b
        torch.cuda.empty_cache()

    # ---- training steps: the headline ---------------------------------------------------------------
    stepper = TrainStepper(model, R, learning_rate=0.03, apply_diffuse_render_regularization=True, ray_selection=args.ray_selection, backward=args.backward, deterministic=args.deterministic)
    torch.manual_seed(1234 + rank)  # every rank draws its own rays
    batches = dataset.image_batches(args.images)
    for _ in range(args.warmup):
        stepper.step(dataset, next(batches))
    # HIP events are recorded on every `timer_stride`-th timed step only: hundreds of outstanding timing events
    # make the ROCm runtime stall for tens of ms now and then (measured: 2.0 -> 2.9 ms/step in some runs)
    timer_stride = max(1, args.steps // max(1, args.timed_steps))
    n_timed = 0 if args.no_kernel_timer else len(range(0, args.steps, timer_stride))
    timer = ops.KernelTimer(preallocate=12 * n_timed)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ops.KERNEL_TIMER = timer if (not args.no_kernel_timer and i % timer_stride == 0) else None
        stats = stepper.step(dataset, next(batches))
    host_issue = time.perf_counter() - t0  # time the host needed to enqueue all steps (GPU runs asynchronously)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    ops.KERNEL_TIMER = None
    if world > 1:
        te = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(te, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(te.item())
        # every rank leaves the group together, BEFORE rank 0 starts its single-process reporting legs
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()

    if rank != 0:
        return
    ms_per_step = elapsed / args.steps * 1e3
    value = world * 2 * R * S * args.steps / elapsed

    # ---- roofline of the dominant kernel (HIP events recorded inside the timed region) ------------
    ksum = timer.summary()
    rays, pixels = stepper.select(dataset, next(batches))
    n_in = count_inside(rays.origins, rays.directions, S, grid.aabb)
    C = 3 * (args.sh_degree + 1) ** 2 + 1
    alg_bytes = {
        f"render_forward[sh{args.sh_degree},save]": n_in * 8 * C * 4 + R * 48,
        "render_forward[diffuse,save]": n_in * 8 * 4 * 4 + R * 48,
        f"render_backward[sh{args.sh_degree}]": n_in * 8 * C * 4 + R * 48,
        "render_backward[diffuse]": n_in * 8 * 4 * 4 + R * 48,
        "adam_step": G**3 * C * 4 * 7,
        # binned specular backward: the brick pass is the kernel that moves the scatter payload (8 corners x C
        # x 4 B per in-AABB sample) into the gradient tensor; emit / bin / scatter-expand are its front end
        f"brick_accumulate[sh{args.sh_degree}]": n_in * 8 * C * 4,
        "brick_accumulate[diffuse]": n_in * 8 * 4 * 4,
    }
    kernels = {}
    for name, rec in ksum.items():
        b = alg_bytes.get(name)
        kernels[name] = {"avg_ms": rec["avg_ms"], "launches": rec["launches"]}
        if b:
            kernels[name]["algorithmic_GB"] = b / 1e9
            kernels[name]["effective_GBps"] = b / 1e9 / (rec["avg_ms"] / 1e3)
    render_kernels = {k: v for k, v in ksum.items() if k in alg_bytes}  # every kernel of the step with a byte model
    if not render_kernels:
        print(json.dumps({"ms_per_step": ms_per_step, "value": value, "host_issue_ms_per_step": host_issue / args.steps * 1e3, "note": "kernel timer off"}))
        return
    dom = max(render_kernels, key=lambda k: render_kernels[k]["total_ms"])
    achieved = alg_bytes[dom] / 1e9 / (ksum[dom]["avg_ms"] / 1e3)
    roofline = {
        "kernel": dom,
        "bound": "hbm",
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "traffic": None,
        "algorithmic_bytes_per_launch": alg_bytes[dom],
        "avg_launch_ms": ksum[dom]["avg_ms"],
        "note": (
            "streaming kernel: 4 reads + 3 writes of 4 B per parameter (param, grad, exp_avg, exp_avg_sq); algorithmic = real traffic"
            if dom == "adam_step"
            else "effective bandwidth: algorithmic gather/scatter bytes (8 corners x C x 4 B per in-AABB sample, SURVEY 8d), "
            "not credited for cache reuse or skipped zero-weight samples, so it can exceed DRAM traffic"
        ),
        "by_kernel": {
            k: {"avg_launch_ms": ksum[k]["avg_ms"], "achieved": alg_bytes[k] / 1e9 / (ksum[k]["avg_ms"] / 1e3), "frac": alg_bytes[k] / 1e9 / (ksum[k]["avg_ms"] / 1e3) / HBM_PEAK_GBS}
            for k in render_kernels
        },
    }
    spec = f"sh{args.sh_degree}"
    pipeline = [k for k in ksum if k in (f"render_backward_emit[{spec}]", f"render_backward_emit_direct[{spec}]", "sort_keys", "expand_records", f"scatter_records[{spec}]", f"brick_accumulate[{spec}]")]
    if f"brick_accumulate[{spec}]" in ksum:
        # the whole specular backward (emit -> bin -> scatter-expand -> brick pass) against the same scatter payload
        # (bin_offsets, 10 us, is shared by both passes and counted once)
        total_ms = sum(ksum[k]["avg_ms"] for k in pipeline) + ksum.get("bin_offsets", {"avg_ms": 0.0})["avg_ms"]
        pbytes = alg_bytes[f"brick_accumulate[{spec}]"] + R * 48
        roofline["pipeline"] = {
            "kernels": pipeline,
            "total_ms": total_ms,
            "algorithmic_bytes": pbytes,
            "achieved": pbytes / 1e9 / (total_ms / 1e3),
            "frac": pbytes / 1e9 / (total_ms / 1e3) / HBM_PEAK_GBS,
        }
        roofline["pipeline"]["note"] = "the binned specular backward as a whole (scan -> emit at final positions -> brick pass; its counting runs inside the forward pass) against the scatter payload of SURVEY 8d"
    prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(prof):
        try:
            table = json.load(open(prof))
            roofline["traffic"] = table.get(dom, {}).get("hbm_bytes_per_launch")
            roofline["traffic_source"] = table.get("_source")
        except Exception:
            pass

    baseline = None
    if args.cpu_rays > 0:
        baseline = cpu_baseline(grid, (rays.origins.cpu(), rays.directions.cpu()), pixels.cpu(), S, args.cpu_rays, args.cpu_threads)

    line = {
        "metric": "ray-samples/sec (fwd+bwd) training step, 800x800 images @ 128^3 SH-2 ReLU field",
        "value": value,
        "unit": "ray-samples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "host_issue_ms_per_step": host_issue / args.steps * 1e3,
        "kernel_timer_steps": n_timed,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"configs[2]: train step on {G}^3 SH-degree-{args.sh_degree} ReLU field (U(-1,1) init), {args.images} synthetic {H}x{W} images, "
            f"{R} random distinct pixels/GPU/step out of all {args.images}x{H}x{W} ({args.ray_selection} selection), {S} jittered samples/ray, specular+diffuse fwd+bwd, L1+L1, fused Adam"
            + (", gradient exchange over RCCL: reduce-scatter -> Adam on 1/N of the grid per rank -> all-gather" if world > 1 else ""),
            "rays_per_gpu_per_step": R,
            "samples_per_ray": S,
            "renders_per_step": 2,
            "parallelism": f"dp{world}" + ("+zero1" if world > 1 and stepper.shard_optimizer else ""),
            "grid_storage": args.storage,
            "ray_selection": args.ray_selection,
            "backward": stepper.backward,
        },
        "rays_per_s": world * 2 * R * args.steps / elapsed,
        "final_specular_psnr": stats.psnr()["specular_psnr"],
        "inside_fraction": n_in / (R * S),
        "kernels": kernels,
        "roofline": roofline,
        "cpu_baseline": baseline,
        "fwd_render": fwd_render,
        "highres_render": highres,
    }
    print(json.dumps(line))


if __name__ == "__main__":
    main()
