#!/usr/bin/env python
"""PSNR after equal training steps for the backward variants ("PSNR within 0.05 dB after equal steps").

Trains the bench.py workload (128^3 SH-2 ReLU field, 8 synthetic 800x800 images, 16384 rays x 256 samples) for
--steps iterations with identical seeds, once per variant and repeat, and evaluates the mean PSNR of full renders of
the 8 training views + 2 held-out views (no jitter).  Atomic runs differ from each other (float atomics reorder sums), so
their spread is the yardstick for the other variants.

    python tools/psnr_equal_steps.py --steps 1000 --repeats 2
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import thr3ed_atom_amd as rf  # noqa: E402
from thr3ed_atom_amd.trainers import PosedImagesInMemory, TrainStepper  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--repeats", type=int, default=2)
    ap.add_argument("--grid", type=int, default=128)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    H = W = 800
    intr = rf.CameraIntrinsics(H, W, 1111.111)
    bounds = rf.CameraBounds(bench.NEAR, bench.FAR)
    gt = bench.make_grid(dev, args.grid, 2, seed=7, sparse=True)
    gt_cfg = rf.SHVoxGridRenderConfig(256, bounds, perturb_sampled_points=False, white_bkgd=True)
    gt_model = rf.VolumetricModel(gt, rf.render_sh_voxel_grid, gt_cfg, device=dev)
    train_poses = [rf.pose_spherical(45.0 * k, -30.0, bench.RADIUS) for k in range(8)]
    test_poses = [rf.pose_spherical(22.5, -30.0, bench.RADIUS), rf.pose_spherical(200.0, -40.0, bench.RADIUS)]
    images = torch.stack([gt_model.render(p, intr).colour.permute(2, 0, 1) for p in train_poses])
    test_images = [gt_model.render(p, intr).colour for p in test_poses]
    pose_mat = torch.stack([torch.cat([p.rotation, p.translation], dim=1) for p in train_poses]).to(dev)
    dataset = PosedImagesInMemory(images, pose_mat, intr, bounds)

    def psnr(a, b):
        return float(-10.0 * torch.log10(torch.mean((a - b) ** 2)))

    for variant in ("atomic", "binned", "binned-deterministic"):
        for rep in range(args.repeats):
            grid = bench.make_grid(dev, args.grid, 2, seed=42, storage="split")
            cfg = rf.SHVoxGridRenderConfig(256, bounds, perturb_sampled_points=True, white_bkgd=True)
            model = rf.VolumetricModel(grid, rf.render_sh_voxel_grid, cfg, device=dev)
            stepper = TrainStepper(model, 16384, learning_rate=0.03, backward=variant.split("-")[0], deterministic=variant.endswith("deterministic"))
            torch.manual_seed(1234)
            batches = dataset.image_batches(8)
            for _ in range(args.steps):
                stepper.step(dataset, next(batches))
            tr = sum(psnr(model.render(p, intr, perturb_sampled_points=False).colour, images[k].permute(1, 2, 0)) for k, p in enumerate(train_poses)) / 8
            te = sum(psnr(model.render(p, intr, perturb_sampled_points=False).colour, t) for p, t in zip(test_poses, test_images)) / 2
            print(f"{variant:22s} run {rep}: train-view PSNR {tr:.3f} dB, held-out PSNR {te:.3f} dB after {args.steps} steps", flush=True)


if __name__ == "__main__":
    main()
