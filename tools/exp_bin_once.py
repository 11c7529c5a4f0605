"""Go / no-go measurement of "bin once, use twice" (DESIGN.md): would a forward pass made of
   (a) a density-only march of both renders (no feature gather in the specular render) and
   (b) a feature gather fed by the specular samples SORTED BY BRICK (each brick's `rest` channels fetched once)
beat today's forward pair?  (a) is timed by bench.py on a -DRF_EXP_NO_P1 build (tools/exp_bin_once.sh); this script times (b): it runs
the bench step for a few iterations on a -DRF_EXP_GATHER build, takes the sorted specular record list of the last one (same samples, same
index quads, same directions as the sample records would have) and launches exp_gather_sorted_kernel over it.  GPU box only."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import thr3ed_atom_amd as rf  # noqa: E402
from thr3ed_atom_amd import _lib  # noqa: E402
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
torch.manual_seed(1234)
batches = data.image_batches(8)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 15):
    stepper.step(data, next(batches))
torch.cuda.synchronize()
t = stepper._exec["tensors"]
srt, off = t["pass0"]["sorted"], t["pass0"]["offsets"]
nrec = int(off[-1].item())
out = torch.zeros((srt.shape[0], 3), dtype=torch.float32, device=dev)
lib = _lib.load()
rf_grid = grid.to_rf_grid()
stream = torch.cuda.current_stream(dev).cuda_stream
fn = lib.rf_exp_gather_sorted
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]


def launch():
    rc = fn(C.byref(rf_grid), srt.data_ptr(), off.data_ptr(), int(stepper.brick_size), out.data_ptr(), stream)
    assert rc == 0, rc


for _ in range(3):
    launch()
torch.cuda.synchronize()
times = []
for rep in range(10):
    # between two launches: a streaming pass over a buffer larger than L2 + Infinity Cache, like the rest of an iteration would be
    torch.empty(96 * 1024 * 1024, dtype=torch.float32, device=dev).fill_(1.0)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    launch()
    b.record()
    torch.cuda.synchronize()
    times.append(a.elapsed_time(b))
warm = []
for rep in range(10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    launch()
    b.record()
    torch.cuda.synchronize()
    warm.append(a.elapsed_time(b))
used = int((out.abs().sum(dim=1) > 0).sum().item())
print(f"sorted specular samples: {nrec}; brick_size {stepper.brick_size}; outputs written {used}")
print(f"exp_gather_sorted: {np.median(times):.4f} ms after a 384 MB streaming write (cold caches), {np.median(warm):.4f} ms back to back (grid in the Infinity Cache); "
      f"min {min(times + warm):.4f}")
print(f"bytes: rest tensor {grid.kernel_tensors()[1].numel() * 4 / 1e6:.1f} MB once, records read {nrec * 48 / 1e6:.1f} MB (a sample record would be 32 B), rgb written {nrec * 12 / 1e6:.1f} MB")
