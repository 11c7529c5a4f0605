#!/usr/bin/env python
"""Train a ReLU field from posed images on MI355X -- counterpart of the reference CLI
thre3d_elements/relu_fields/train_sh_based_voxel_grid_with_posed_images.py: a click command with the reference's option names and
defaults (:38-134).  The reference's disk dataset loader (thre3d_atom/data, out of scope of this build) is replaced by an .npz file
or a synthetic scene; options that only steer out-of-scope parts (TensorBoard feedback, data workers, scene normalisation) are
accepted so that existing command lines keep working, and say so.

    python scripts/train_sh_based_voxel_grid.py -o out --synthetic True            # procedural scene, no data needed
    python scripts/train_sh_based_voxel_grid.py -d scene.npz -o out                # images [M,3,H,W], poses [M,3,4], focal, near, far
    python -m torch.distributed.run --nproc-per-node 8 scripts/train_sh_based_voxel_grid.py ...   # data parallel (extension)
"""
import os
import sys

import click
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


# -------------------------------------------------------------------------------------
#  Command line configuration for the script (option names / defaults of the reference) |
# -------------------------------------------------------------------------------------
# fmt: off
@click.command()
# Required arguments:
@click.option("-d", "--data_path", type=click.Path(), required=False, default=None,
              help=".npz with images [M,3,H,W] in [0,1], poses [M,3,4] (camera-to-world), focal, near, far (reference: a dataset directory)")
@click.option("-o", "--output_path", type=click.Path(file_okay=False, dir_okay=True), required=True, help="path for training output")
# Input dataset related arguments:
@click.option("--separate_train_test_folders", type=click.BOOL, required=False, default=True, help="(disk loader option of the reference: accepted, unused)")
@click.option("--data_downsample_factor", type=click.FloatRange(min=1.0), required=False, default=1.0, help="downscale factor for the input images")
# Voxel-grid related arguments:
@click.option("--grid_dims", type=click.INT, nargs=3, required=False, default=(256, 256, 256), help="dimensions (#voxels) of the grid along x, y and z axes")
@click.option("--grid_location", type=click.FLOAT, nargs=3, required=False, default=(0.0, 0.0, 0.0), help="dimensions (#voxels) of the grid along x, y and z axes")
@click.option("--normalize_scene_scale", type=click.BOOL, required=False, default=False, help="(disk loader option of the reference: accepted, must stay False)")
@click.option("--grid_world_size", type=click.FLOAT, nargs=3, required=False, default=(3.0, 3.0, 3.0), help="size (extent) of the grid in world coordinate system")
@click.option("--sh_degree", type=click.INT, required=False, default=2, help="degree of the spherical harmonics coefficients to be used")
@click.option("--use_relu_field", type=click.BOOL, required=False, default=True, help="whether to use relu_fields or revert to traditional grids")
@click.option("--use_softplus_field", type=click.BOOL, required=False, default=False, help="whether to use softplus_field or relu_field")
# Rendering related arguments:
@click.option("--render_num_samples_per_ray", type=click.INT, required=False, default=1024, help="number of samples taken per ray during rendering")
@click.option("--parallel_rays_chunk_size", type=click.INT, required=False, default=32768, help="number of parallel rays processed on the GPU (honoured by the chunked render path)")
@click.option("--white_bkgd", type=click.BOOL, required=False, default=True, help="whether to use white background for training with synthetic (background-less) scenes")
# Training related arguments:
@click.option("--ray_batch_size", type=click.INT, required=False, default=16384, help="number of randomly sampled rays used per training iteration")
@click.option("--train_num_samples_per_ray", type=click.INT, required=False, default=512, help="number of samples taken per ray during training")
@click.option("--num_stages", type=click.INT, required=False, default=4, help="number of progressive growing stages used in training")
@click.option("--num_iterations_per_stage", type=click.INT, required=False, default=7000, help="number of training iterations performed per stage")
@click.option("--scale_factor", type=click.FLOAT, required=False, default=2.0, help="factor by which the grid is up-scaled after each stage")
@click.option("--learning_rate", type=click.FLOAT, required=False, default=0.03, help="learning rate used at the beginning (ADAM OPTIMIZER)")
@click.option("--lr_decay_steps_per_stage", type=click.INT, required=False, default=3000, help="number of iterations after which lr is exponentially decayed per stage")
@click.option("--lr_decay_gamma_per_stage", type=click.FLOAT, required=False, default=0.1, help="value of gamma for exponential lr_decay (happens per stage)")
@click.option("--stagewise_lr_decay_gamma", type=click.FLOAT, required=False, default=1.0, help="value of gamma used for reducing the learning rate after each stage")
@click.option("--apply_diffuse_render_regularization", type=click.BOOL, required=False, default=True, help="whether to apply the diffuse render regularization")
@click.option("--num_workers", type=click.INT, required=False, default=4, help="(data loader option of the reference: accepted, unused -- the dataset is resident in HBM)")
# Various frequencies:
@click.option("--save_frequency", type=click.INT, required=False, default=250, help="number of iterations after which a model is saved")
@click.option("--test_frequency", type=click.INT, required=False, default=250, help="number of iterations after which test metrics are computed")
@click.option("--feedback_frequency", type=click.INT, required=False, default=100, help="(TensorBoard feedback of the reference: accepted, unused)")
@click.option("--summary_frequency", type=click.INT, required=False, default=50, help="number of iterations after which current loss is logged to console")
# Miscellaneous modes
@click.option("--verbose_rendering", type=click.BOOL, required=False, default=False, help="(accepted, unused)")
@click.option("--fast_debug_mode", type=click.BOOL, required=False, default=False, help="(accepted, unused)")
# Extensions of this build:
@click.option("--synthetic", type=click.BOOL, required=False, default=False, help="train on a procedural scene rendered on the fly (no data needed)")
@click.option("--synthetic_size", type=click.INT, required=False, default=200, help="image size of the synthetic scene")
@click.option("--global_batch", type=click.BOOL, required=False, default=False,
              help="data parallel: --ray_batch_size is the GLOBAL batch split over the ranks (N GPUs reproduce the single-GPU run); default: per rank")
@click.option("--seed", type=click.INT, required=False, default=42, help="seed of torch's generators")
# fmt: on
# -------------------------------------------------------------------------------------
def main(**kwargs) -> None:
    config = dict(kwargs)
    if config["normalize_scene_scale"]:
        raise click.UsageError("--normalize_scene_scale belongs to the reference's disk dataset loader, which this build does not have")
    rank, local_rank, world = rfdist.init_from_env()
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    # ray selection draws from the CPU generator: equal seeds on all ranks for one global batch, distinct ones otherwise
    torch.manual_seed(config["seed"] if config["global_batch"] else config["seed"] + rank)
    if config["synthetic"] or config["data_path"] is None:
        data = synthetic_dataset(dev, config["synthetic_size"], 24, min(config["train_num_samples_per_ray"], 256))
    else:
        z = np.load(config["data_path"])
        images = torch.from_numpy(z["images"]).float().to(dev)
        intr = rf.CameraIntrinsics(images.shape[2], images.shape[3], float(z["focal"]))
        data = PosedImagesInMemory(images, torch.from_numpy(z["poses"]).float().to(dev), intr, rf.CameraBounds(float(z["near"]), float(z["far"])))
    if config["data_downsample_factor"] != 1.0:
        data = data.downsampled(config["data_downsample_factor"])
    test = PosedImagesInMemory(data.images[-2:], data.poses[-2:], data.camera_intrinsics, data.camera_bounds)
    train = PosedImagesInMemory(data.images[:-2], data.poses[:-2], data.camera_intrinsics, data.camera_bounds)

    world_size = tuple(config["grid_world_size"])
    if config["use_relu_field"] and not config["use_softplus_field"]:  # the three configurations of the reference (:169-192)
        acts = dict(density_preactivation=torch.nn.Identity(), density_postactivation=torch.nn.ReLU(),
                    expected_density_scale=rf.compute_expected_density_scale_for_relu_field_grid(world_size))
    elif config["use_softplus_field"]:
        acts = dict(density_preactivation=torch.nn.Identity(), density_postactivation=torch.nn.Softplus(),
                    expected_density_scale=rf.compute_expected_density_scale_for_relu_field_grid(world_size))
    else:
        acts = dict(density_preactivation=torch.abs, density_postactivation=torch.nn.Identity(), expected_density_scale=1.0)
    F = 3 * (config["sh_degree"] + 1) ** 2
    dims = tuple(config["grid_dims"])
    grid = rf.VoxelGrid(
        torch.empty((*dims, 1), device=dev).uniform_(-1, 1), torch.empty((*dims, F), device=dev).uniform_(-1, 1),
        rf.VoxelSize(*[w / d for w, d in zip(world_size, dims)]), rf.VoxelGridLocation(*config["grid_location"]), tunable=True, **acts,
    )
    cfg = rf.SHVoxGridRenderConfig(config["train_num_samples_per_ray"], data.camera_bounds, white_bkgd=config["white_bkgd"],
                                   render_num_samples_per_ray=config["render_num_samples_per_ray"],
                                   parallel_rays_chunk_size=config["parallel_rays_chunk_size"])
    model = rf.VolumetricModel(grid, rf.render_sh_voxel_grid, cfg, device=dev)
    train_sh_vox_grid_vol_mod_with_posed_images(
        model, train, config["output_path"], test_dataset=test, ray_batch_size=config["ray_batch_size"], num_stages=config["num_stages"],
        num_iterations_per_stage=config["num_iterations_per_stage"], scale_factor=config["scale_factor"], learning_rate=config["learning_rate"],
        lr_decay_gamma_per_stage=config["lr_decay_gamma_per_stage"], lr_decay_steps_per_stage=config["lr_decay_steps_per_stage"],
        stagewise_lr_decay_gamma=config["stagewise_lr_decay_gamma"], save_freq=config["save_frequency"], test_freq=config["test_frequency"],
        summary_freq=config["summary_frequency"], apply_diffuse_render_regularization=config["apply_diffuse_render_regularization"],
        global_batch=config["global_batch"],
    )


if __name__ == "__main__":
    main()
