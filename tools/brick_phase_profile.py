"""Phase timing of the brick pass (development tool).  Builds a -DRF_BRICK_PROFILE copy of the library, runs training steps of
the bench configuration and prints the s_memtime spans that thread 0 of every brick workgroup spent in each phase.

    python tools/brick_phase_profile.py build          (here, no GPU)
    RF_LIB_PATH=tools/librelu_field_hip_prof.so python tools/brick_phase_profile.py run   (on the GPU box)
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROF = os.path.join(ROOT, "tools", "librelu_field_hip_prof.so")

if sys.argv[1] == "build":
    from thr3ed_atom_amd import _lib

    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + _lib.HIPCC_FLAGS + ["-DRF_BRICK_PROFILE", "-I", _lib.INCLUDE_DIR,
          os.path.join(_lib.CSRC_DIR, "relu_field_kernels.hip"), "-o", PROF]
    subprocess.run(cmd, check=True)
    print("built", PROF)
    sys.exit(0)

import torch  # noqa: E402

import bench  # noqa: E402
import thr3ed_atom_amd as rf  # noqa: E402
from thr3ed_atom_amd import _lib  # noqa: E402
from thr3ed_atom_amd.trainers import PosedImagesInMemory, TrainStepper  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
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
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for _ in range(5):
    stepper.step(data, next(batches))
out = (C.c_ulonglong * 8)()
lib.rf_debug_brick_profile(out, 1)
from thr3ed_atom_amd import ops  # noqa: E402

events = [ops.StepEvents() for _ in range(steps)]
for k in range(steps):
    stepper.step_events = events[k]
    stepper.step(data, next(batches))
stepper.step_events = None
torch.cuda.synchronize()
lib.rf_debug_brick_profile(out, 0)
brick_ms = sum(e.elapsed_ms()["brick_accumulate"] for e in events) / steps
print(f"brick pass of this instrumented build: {brick_ms:.4f} ms per launch ($RF_BRICK_STAGGER = {os.environ.get('RF_BRICK_STAGGER', '0')})")
names = ["range set-up", "batch: wait for loads + LDS stores + barrier", "batch: barriers after record pass / tiles", "batch: lists + tiles (MFMA), wave 0", "flush / optimizer", "accumulator image", "batch: issue of the next loads", "batch: record pass, wave 0"]
from thr3ed_atom_amd.ops import brick_counts  # noqa: E402

_nb = brick_counts(grid, stepper.brick_size)
nb = _nb[0] * _nb[1] * _nb[2] * steps
print('bricks per launch', nb // steps, 'brick_size', stepper.brick_size)
tot = sum(out[i] for i in range(len(names)))
for i, nm in enumerate(names):
    print(f"{nm:34s} {out[i] / nb:10.0f} ticks per brick   {100.0 * out[i] / max(tot, 1):5.1f} %")
print(f"{'sum':34s} {tot / nb:10.0f} ticks per brick (s_memtime ticks = 100 MHz constant clock or shader cycles, see DESIGN)")
