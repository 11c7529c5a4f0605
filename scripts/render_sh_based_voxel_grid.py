#!/usr/bin/env python
"""Render a camera path of a trained model -- counterpart of the reference CLI
thre3d_elements/relu_fields/render_sh_based_voxel_grid.py (a click command with the same option names and defaults, :28-61).
Checkpoints written by this build OR by the reference load (create_volumetric_model_from_saved_model maps the reference's pickled
names).  Video encoding (imageio) is out of scope: the colour frames are written as PNG files (PIL) or, without PIL, as .npy.

    python scripts/render_sh_based_voxel_grid.py -i out/saved_models/model_final.pth -o frames --num_frames 42
"""
import os
import sys
import time

import click
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import thr3ed_atom_amd as rf  # noqa: E402


# fmt: off
@click.command()
# Required arguments:
@click.option("-i", "--model_path", type=click.Path(file_okay=True, dir_okay=False), required=True, help="path to the trained (reconstructed) model")
@click.option("-o", "--output_path", type=click.Path(file_okay=False, dir_okay=True), required=True, help="path for saving rendered output")
# Non-required Render configuration options:
@click.option("--overridden_num_samples_per_ray", type=click.IntRange(min=1), default=512, required=False, help="overridden (increased) num_samples_per_ray for beautiful renders :)")
@click.option("--render_scale_factor", type=click.FLOAT, default=2.0, required=False, help="overridden (increased) resolution (again :D) for beautiful renders :)")
@click.option("--camera_path", type=click.Choice(["thre360", "spiral"]), default="thre360", required=False, help="which camera path to use for rendering the animation")
# thre360_path options
@click.option("--camera_pitch", type=click.FLOAT, default=60.0, required=False, help="pitch-angle value for the camera for 360 path animation")
@click.option("--num_frames", type=click.IntRange(min=1), default=180, required=False, help="number of frames in the video")
# spiral path options
@click.option("--vertical_camera_height", type=click.FLOAT, default=3.0, required=False, help="height at which the camera spiralling will happen")
@click.option("--num_spiral_rounds", type=click.IntRange(min=1), default=2, required=False, help="number of rounds made while transitioning between spiral radii")
# Non-required video options:
@click.option("--fps", type=click.IntRange(min=1), default=60, required=False, help="(video option of the reference: accepted, unused -- frames are written)")
# fmt: on
def main(**kwargs) -> None:
    config = dict(kwargs)
    dev = torch.device("cuda:0")
    creator = lambda info: rf.create_voxel_grid_from_saved_info_dict(info, storage="split")  # noqa: E731
    model, extra = rf.create_volumetric_model_from_saved_model(config["model_path"], creator, device=dev)
    radius, intr = extra["hemispherical_radius"], extra["camera_intrinsics"]
    if config["camera_path"] == "thre360":
        poses = rf.get_thre360_animation_poses(radius, config["camera_pitch"], config["num_frames"])
    else:
        poses = rf.get_thre360_spiral_animation_poses((radius / 8.0, radius), config["vertical_camera_height"], config["num_spiral_rounds"], config["num_frames"])
    intr = rf.scale_camera_intrinsics(intr, config["render_scale_factor"])
    os.makedirs(config["output_path"], exist_ok=True)
    try:
        from PIL import Image
    except ImportError:
        Image = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i, pose in enumerate(poses):
        out = model.render(pose, intr, num_samples_per_ray=config["overridden_num_samples_per_ray"])
        frame = (out.colour.clamp(0, 1) * 255).byte().cpu().numpy()
        if Image is not None:
            Image.fromarray(frame).save(os.path.join(config["output_path"], f"frame_{i:04d}.png"))
        else:
            np.save(os.path.join(config["output_path"], f"frame_{i:04d}.npy"), frame)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{len(poses)} frames of {intr.height}x{intr.width} x {config['overridden_num_samples_per_ray']} samples in {dt:.2f} s "
          f"({len(poses) / dt:.1f} fps incl. host copies and image encoding)")


if __name__ == "__main__":
    main()
