"""Distribution of the records a brick workgroup processes per list on the bench step (development tool; run on the GPU box)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import thr3ed_atom_amd as rf  # noqa: E402
from thr3ed_atom_amd.trainers import PosedImagesInMemory, TrainStepper  # noqa: E402

dev = torch.device("cuda:0")
bounds = rf.CameraBounds(bench.NEAR, bench.FAR)
intr = rf.CameraIntrinsics(800, 800, 1111.111)
gt = bench.make_grid(dev, 128, 2, seed=7, sparse=True)
gt_model = rf.VolumetricModel(gt, rf.render_sh_voxel_grid, rf.SHVoxGridRenderConfig(256, bounds, perturb_sampled_points=False, white_bkgd=True), device=dev)
poses = [rf.pose_spherical(45.0 * k, -30.0, bench.RADIUS) for k in range(8)]
images = torch.stack([gt_model.render(p, intr).colour.permute(2, 0, 1) for p in poses])
pose_mat = torch.stack([torch.cat([p.rotation, p.translation], dim=1) for p in poses]).to(dev)
data = PosedImagesInMemory(images, pose_mat, intr, bounds)
grid = bench.make_grid(dev, 128, 2, seed=42, storage="split")
model = rf.VolumetricModel(grid, rf.render_sh_voxel_grid, rf.SHVoxGridRenderConfig(256, bounds, perturb_sampled_points=True, white_bkgd=True), device=dev)
stepper = TrainStepper(model, 16384, 0.03)
batches = data.image_batches(8)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 15):
    stepper.step(data, next(batches))
torch.cuda.synchronize()
off = stepper._exec["tensors"]["offsets2"].cpu().numpy()  # [2, nkeys + 1]
nb = 16
for li, name in enumerate(("specular (wide)", "diffuse (narrow)")):
    o = off[li]
    cnt = (o[1:] - o[:-1]).reshape(nb, 2, nb, nb, 4)  # [bx, fx, by, bz, fy | fz << 1]
    per = np.zeros((nb, nb, nb), dtype=np.int64)
    # a brick (bx, by, bz) processes: its own records (all flags) + the flagged records of the lower neighbours that reach into it
    for ox in (0, 1):
        for oy in (0, 1):
            for oz in (0, 1):
                sel = np.zeros((nb, nb, nb), dtype=np.int64)
                for fx in (0, 1):
                    for f in range(4):
                        fy, fz = f & 1, f >> 1
                        if (fx >= ox) and (fy >= oy) and (fz >= oz):
                            sel += cnt[:, fx, :, :, f]
                # shift: neighbour (bx - ox, ...) contributes to (bx, by, bz)
                shifted = np.zeros_like(sel)
                shifted[ox:, oy:, oz:] = sel[: nb - ox, : nb - oy, : nb - oz]
                per += shifted
    p = per.reshape(-1)
    print(f"{name}: records/brick processed mean {p.mean():.0f} median {np.median(p):.0f} max {p.max()} zero-bricks {(p == 0).sum()}")
    for bs in (128, 192, 256, 320, 384, 512):
        print(f"   batch {bs}: batches/brick mean {np.ceil(p / bs).mean():.2f}")
    print("   histogram (<=64, <=128, <=192, <=256, <=320, <=384, <=512, <=768, >768):", [int(((p > a) & (p <= b)).sum()) for a, b in ((-1, 64), (64, 128), (128, 192), (192, 256), (256, 320), (320, 384), (384, 512), (512, 768), (768, 1 << 30))])
