#!/usr/bin/env python
"""Kernel time of the frame render (HIP events around the launch, median of N frames) for configs[1] and configs[4] (with the mask) --
one line per configuration; the tile-scheduling switches ($RF_TILE_WPB, $RF_TILE_XCD_ROWS, $RF_FRAME_TILES) are read by the library
once per process, so an A/B is one process per variant (tools/exp_tile_sched.sh).

    python tools/frame_time.py [frames]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import thr3ed_atom_amd as rf  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 9
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
bounds = rf.CameraBounds(bench.NEAR, bench.FAR)
intr = rf.CameraIntrinsics(800, 800, 1111.111)
pose = rf.pose_spherical(30.0, -30.0, bench.RADIUS)
out = {"RF_TILE_WPB": os.environ.get("RF_TILE_WPB"), "RF_TILE_XCD_ROWS": os.environ.get("RF_TILE_XCD_ROWS"), "RF_FRAME_TILES": os.environ.get("RF_FRAME_TILES")}
for name, G, S, sparse, over, deg in (("cfg1", 128, 256, False, {}, 2), ("cfg4_mask", 256, 512, True, {"use_occupancy_mask": True}, 2), ("cfg4_nomask", 256, 512, True, {}, 2),
                                      ("cfg1_sh0", 128, 256, False, {}, 0)):
    grid = bench.make_grid(dev, G, deg, seed=11 if sparse else 42, sparse=sparse, storage="split")
    if over:
        grid.build_occupancy()
    cfg = rf.SHVoxGridRenderConfig(S, bounds, perturb_sampled_points=True, white_bkgd=True)
    model = rf.VolumetricModel(grid, rf.render_sh_voxel_grid, cfg, device=dev)
    wall, kms, launches = bench.time_frames(lambda: model.render(pose, intr, **over), frames, kernel=f"render_forward[sh{deg},frame]")
    out[name] = {"kernel_ms": round(kms, 4), "wall_ms": round(wall * 1e3, 4)}
    del model, grid
    torch.cuda.empty_cache()
# SH degree 1 / 3 (round 6: the packet kernel's generic rest path) against the per-ray kernel ($RF_FRAME_TILES is read per call)
for deg in (1, 3):
    grid = bench.make_grid(dev, 128, deg, seed=42, storage="split")
    cfg = rf.SHVoxGridRenderConfig(256, bounds, perturb_sampled_points=True, white_bkgd=True)
    model = rf.VolumetricModel(grid, rf.render_sh_voxel_grid, cfg, device=dev)
    for tiles in ("1", "0"):
        os.environ["RF_FRAME_TILES"] = tiles
        wall, kms, launches = bench.time_frames(lambda: model.render(pose, intr), frames, kernel=f"render_forward[sh{deg},frame]")
        out[f"cfg1_sh{deg}_{'packets' if tiles == '1' else 'per_ray'}"] = {"kernel_ms": round(kms, 4), "wall_ms": round(wall * 1e3, 4)}
    os.environ.pop("RF_FRAME_TILES", None)
    del model, grid
    torch.cuda.empty_cache()
print(json.dumps(out))
