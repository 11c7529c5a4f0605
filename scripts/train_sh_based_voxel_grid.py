#!/usr/bin/env python
"""Train a ReLU field from posed images on MI355X -- counterpart of the reference CLI
thre3d_elements/relu_fields/train_sh_based_voxel_grid_with_posed_images.py (same flag names for everything on the
render/training path; the disk dataset loader is replaced by an .npz file or a synthetic scene).

    python scripts/train_sh_based_voxel_grid.py -o out --synthetic            # procedural scene, no data needed
    python scripts/train_sh_based_voxel_grid.py -d scene.npz -o out           # images [M,3,H,W], poses [M,3,4], focal, near, far
    python -m torch.distributed.run --nproc-per-node 8 scripts/train_sh_based_voxel_grid.py ...   # data parallel
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import thr3ed_atom_amd as rf  # noqa: E402
from thr3ed_atom_amd import distributed as rfdist  # noqa: E402
from thr3ed_atom_amd.trainers import PosedImagesInMemory, train_sh_vox_grid_vol_mod_with_posed_images  # noqa: E402


def synthetic_dataset(dev, size, n_views, samples):
    """images of a procedural blob rendered by the HIP renderer itself (hotdog-like camera constants)"""
    near, far, radius, G = float(np.float32(2.0) * 0.9), float(np.float32(6.0) * 1.1), 4.0311, 64
    ax = ((torch.arange(G, device=dev, dtype=torch.float32) + 0.5) / G * 3.0 - 1.5) / 1.5
    r = torch.sqrt(ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2)
    dens = (3.0 * (0.5 - r))[..., None].contiguous()
    feat = torch.stack([torch.sin(3 * ax)[:, None, None].expand(G, G, G), torch.cos(2 * ax)[None, :, None].expand(G, G, G),
                        ax[None, None, :].expand(G, G, G)], dim=-1).contiguous() * 4.0
    gt = rf.VoxelGrid(dens, feat, rf.VoxelSize(3.0 / G, 3.0 / G, 3.0 / G), density_preactivation=torch.nn.Identity(),
                      density_postactivation=torch.nn.ReLU(), expected_density_scale=100.0 / 3.0)
    bounds = rf.CameraBounds(near, far)
    cfg = rf.SHVoxGridRenderConfig(samples, bounds, perturb_sampled_points=False, white_bkgd=True)
    model = rf.VolumetricModel(gt, rf.render_sh_voxel_grid, cfg, device=dev)
    intr = rf.CameraIntrinsics(size, size, size * 1.39)
    poses = [rf.pose_spherical(360.0 * k / n_views, -30.0 + 20.0 * np.sin(k), radius) for k in range(n_views)]
    images = torch.stack([model.render(p, intr).colour.permute(2, 0, 1) for p in poses])
    pose_mat = torch.stack([torch.cat([p.rotation, p.translation], dim=1) for p in poses]).to(dev)
    return PosedImagesInMemory(images, pose_mat, intr, bounds)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-d", "--data_path", default=None, help=".npz with images [M,3,H,W] in [0,1], poses [M,3,4], focal, near, far")
    ap.add_argument("-o", "--output_path", required=True)
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--synthetic_size", type=int, default=200)
    ap.add_argument("--grid_dims", type=int, nargs=3, default=(128, 128, 128))
    ap.add_argument("--grid_world_size", type=float, nargs=3, default=(3.0, 3.0, 3.0))
    ap.add_argument("--sh_degree", type=int, default=2)
    ap.add_argument("--use_relu_field", type=lambda s: s.lower() != "false", default=True)
    ap.add_argument("--use_softplus_field", type=lambda s: s.lower() == "true", default=False)
    ap.add_argument("--render_num_samples_per_ray", type=int, default=512)
    ap.add_argument("--white_bkgd", type=lambda s: s.lower() != "false", default=True)
    ap.add_argument("--ray_batch_size", type=int, default=16384)
    ap.add_argument("--train_num_samples_per_ray", type=int, default=256)
    ap.add_argument("--num_stages", type=int, default=3)
    ap.add_argument("--num_iterations_per_stage", type=int, default=500)
    ap.add_argument("--scale_factor", type=float, default=2.0)
    ap.add_argument("--learning_rate", type=float, default=0.03)
    ap.add_argument("--lr_decay_steps_per_stage", type=int, default=3000)
    ap.add_argument("--lr_decay_gamma_per_stage", type=float, default=0.1)
    ap.add_argument("--stagewise_lr_decay_gamma", type=float, default=1.0)
    ap.add_argument("--apply_diffuse_render_regularization", type=lambda s: s.lower() != "false", default=True)
    ap.add_argument("--save_frequency", type=int, default=1000)
    ap.add_argument("--summary_frequency", type=int, default=50)
    ap.add_argument("--global_batch", action="store_true",
                    help="data parallel: --ray_batch_size is the GLOBAL batch, split over the ranks (strong scaling: N GPUs "
                    "reproduce the single-GPU run); default: every rank draws its own --ray_batch_size rays (weak scaling)")
    ap.add_argument("--seed", type=int, default=42)
    args = ap.parse_args()

    rank, local_rank, world = rfdist.init_from_env()
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    # ray selection draws from the CPU generator: equal seeds on all ranks for one global batch, distinct ones otherwise
    torch.manual_seed(args.seed if args.global_batch else args.seed + rank)
    if args.synthetic or args.data_path is None:
        data = synthetic_dataset(dev, args.synthetic_size, 24, args.train_num_samples_per_ray)
    else:
        z = np.load(args.data_path)
        images = torch.from_numpy(z["images"]).float().to(dev)
        intr = rf.CameraIntrinsics(images.shape[2], images.shape[3], float(z["focal"]))
        data = PosedImagesInMemory(images, torch.from_numpy(z["poses"]).float().to(dev), intr, rf.CameraBounds(float(z["near"]), float(z["far"])))
    test = PosedImagesInMemory(data.images[-2:], data.poses[-2:], data.camera_intrinsics, data.camera_bounds)
    train = PosedImagesInMemory(data.images[:-2], data.poses[:-2], data.camera_intrinsics, data.camera_bounds)

    if args.use_relu_field:
        acts = dict(density_preactivation=torch.nn.Identity(), density_postactivation=torch.nn.ReLU(),
                    expected_density_scale=rf.compute_expected_density_scale_for_relu_field_grid(args.grid_world_size))
    elif args.use_softplus_field:
        acts = dict(density_preactivation=torch.nn.Identity(), density_postactivation=torch.nn.Softplus(),
                    expected_density_scale=rf.compute_expected_density_scale_for_relu_field_grid(args.grid_world_size))
    else:
        acts = dict(density_preactivation=torch.abs, density_postactivation=torch.nn.Identity(), expected_density_scale=1.0)
    F = 3 * (args.sh_degree + 1) ** 2
    dims = tuple(args.grid_dims)
    grid = rf.VoxelGrid(
        torch.empty((*dims, 1), device=dev).uniform_(-1, 1), torch.empty((*dims, F), device=dev).uniform_(-1, 1),
        rf.VoxelSize(*[w / d for w, d in zip(args.grid_world_size, dims)]), tunable=True, **acts,
    )
    cfg = rf.SHVoxGridRenderConfig(args.train_num_samples_per_ray, data.camera_bounds, white_bkgd=args.white_bkgd,
                                   render_num_samples_per_ray=args.render_num_samples_per_ray)
    model = rf.VolumetricModel(grid, rf.render_sh_voxel_grid, cfg, device=dev)
    train_sh_vox_grid_vol_mod_with_posed_images(
        model, train, args.output_path, test_dataset=test, ray_batch_size=args.ray_batch_size, num_stages=args.num_stages,
        num_iterations_per_stage=args.num_iterations_per_stage, scale_factor=args.scale_factor, learning_rate=args.learning_rate,
        lr_decay_gamma_per_stage=args.lr_decay_gamma_per_stage, lr_decay_steps_per_stage=args.lr_decay_steps_per_stage,
        stagewise_lr_decay_gamma=args.stagewise_lr_decay_gamma, save_freq=args.save_frequency, test_freq=args.num_iterations_per_stage,
        summary_freq=args.summary_frequency, apply_diffuse_render_regularization=args.apply_diffuse_render_regularization,
        global_batch=args.global_batch,
    )


if __name__ == "__main__":
    main()
