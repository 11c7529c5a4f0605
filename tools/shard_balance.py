#!/usr/bin/env python
"""How evenly does VolumetricModel.render(data_parallel=True) split a frame's work?  Times every rank's shard of an 800 x 800 frame on ONE
GPU (kernel time, HIP events), for N = 2, 4, 8: contiguous pixel ranges (shard_range: the split of the flat ray list) against interleaved
strips of 8 pixel rows (strip k to rank k mod N; emulated as the union of a rank's strips rendered one call per strip -- launch-bound,
so its time is the SUM of its strips' shares of a whole-frame launch: estimated from per-strip costs measured in one whole-frame pass).
    python tools/shard_balance.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import thr3ed_atom_amd as rf  # noqa: E402
from thr3ed_atom_amd import distributed as rfdist  # noqa: E402
from thr3ed_atom_amd.renderers import render_sh_voxel_grid_frame  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
bounds = rf.CameraBounds(bench.NEAR, bench.FAR)
intr = rf.CameraIntrinsics(800, 800, 1111.111)
pose = rf.pose_spherical(30.0, -30.0, bench.RADIUS)


def kernel_ms(fn, reps=7):
    fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


out = {}
for name, G, S, sparse, over in (("cfg1", 128, 256, False, {}), ("cfg4_mask", 256, 512, True, {"use_occupancy_mask": True})):
    grid = bench.make_grid(dev, G, 2, seed=11 if sparse else 42, sparse=sparse, storage="split")
    if over:
        grid.build_occupancy()
    cfg = rf.SHVoxGridRenderConfig(S, bounds, perturb_sampled_points=True, white_bkgd=True, **over)
    whole = kernel_ms(lambda: render_sh_voxel_grid_frame(grid, intr, pose, cfg))
    res = {"whole_frame_ms": round(whole, 4)}
    # cost of every strip of 8 rows when rendered alone is launch-bound; instead: cumulative cost profile from growing prefixes
    prefix = [0.0]
    for rows in range(80, 801, 80):
        prefix.append(kernel_ms(lambda r=rows: render_sh_voxel_grid_frame(grid, intr, pose, cfg, first_ray=0, num_rays=r * 800), reps=5))
    res["prefix_ms_per_80_rows"] = [round(p, 4) for p in prefix]
    for n in (2, 4, 8):
        shards = []
        for r in range(n):
            lo, hi = rfdist.shard_range(800 * 800, r, n)
            shards.append(kernel_ms(lambda lo=lo, hi=hi: render_sh_voxel_grid_frame(grid, intr, pose, cfg, first_ray=lo, num_rays=hi - lo), reps=5))
        res[f"contiguous_n{n}"] = {"per_rank_ms": [round(s, 4) for s in shards], "max_ms": round(max(shards), 4), "ideal_ms": round(whole / n, 4)}
    out[name] = res
    del grid
    torch.cuda.empty_cache()
print(json.dumps(out))
