#!/usr/bin/env python
"""Render a turn-table of a trained model -- counterpart of the reference CLI
thre3d_elements/relu_fields/render_sh_based_voxel_grid.py (frames are written as .npy; video encoding is out of scope).

    python scripts/render_sh_based_voxel_grid.py -i out/saved_models/model_final.pth -o frames --num_frames 42
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import thr3ed_atom_amd as rf  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-i", "--model_path", required=True)
    ap.add_argument("-o", "--output_path", required=True)
    ap.add_argument("--overridden_num_samples_per_ray", type=int, default=512)
    ap.add_argument("--render_scale_factor", type=float, default=2.0)
    ap.add_argument("--camera_path", choices=["thre360", "spiral"], default="thre360")
    ap.add_argument("--camera_pitch", type=float, default=60.0)
    ap.add_argument("--num_frames", type=int, default=42)
    ap.add_argument("--diffuse", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    creator = lambda info: rf.create_voxel_grid_from_saved_info_dict(info, storage="split")  # noqa: E731
    model, extra = rf.create_volumetric_model_from_saved_model(args.model_path, creator, device=dev)
    intr = rf.scale_camera_intrinsics(extra["camera_intrinsics"], args.render_scale_factor)
    radius = extra["hemispherical_radius"]
    poses = rf.get_thre360_animation_poses(radius, args.camera_pitch - 90.0, args.num_frames + 1)
    os.makedirs(args.output_path, exist_ok=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i, pose in enumerate(poses):
        out = model.render(pose, intr, num_samples_per_ray=args.overridden_num_samples_per_ray, render_diffuse=args.diffuse,
                           perturb_sampled_points=False)
        np.save(os.path.join(args.output_path, f"frame_{i:04d}.npy"), (out.colour.clamp(0, 1) * 255).byte().cpu().numpy())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{len(poses)} frames of {intr.height}x{intr.width} x {args.overridden_num_samples_per_ray} samples in {dt:.2f} s "
          f"({len(poses) / dt:.1f} fps incl. host copies)")


if __name__ == "__main__":
    main()
