#!/usr/bin/env python
"""Nothing but frames: N renders of configs[1] (128^3, 256 samples) or configs[4] (256^3 sparse, 512 samples, occupancy mask) through
VolumetricModel.render -- the process tools/frame_counters.sh runs under rocprofv3 (kernel trace / one --pmc pass per counter set).

    python tools/frame_only.py [cfg1|cfg4] [frames]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402  (make_grid and the camera constants of the benchmark)
import thr3ed_atom_amd as rf  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
bounds = rf.CameraBounds(bench.NEAR, bench.FAR)
intr = rf.CameraIntrinsics(800, 800, 1111.111)
pose = rf.pose_spherical(30.0, -30.0, bench.RADIUS)
if which == "cfg1":
    grid = bench.make_grid(dev, 128, 2, seed=42, storage="split")
    cfg = rf.SHVoxGridRenderConfig(256, bounds, perturb_sampled_points=True, white_bkgd=True)
    over = {}
else:
    grid = bench.make_grid(dev, 256, 2, seed=11, sparse=True, storage="split")
    grid.build_occupancy()
    cfg = rf.SHVoxGridRenderConfig(512, bounds, perturb_sampled_points=True, white_bkgd=True)
    over = {"use_occupancy_mask": True}
model = rf.VolumetricModel(grid, rf.render_sh_voxel_grid, cfg, device=dev)
torch.manual_seed(1)
for _ in range(frames):
    out = model.render(pose, intr, **over)
torch.cuda.synchronize()
print(which, frames, "frames; mean colour", float(out.colour.mean()))
