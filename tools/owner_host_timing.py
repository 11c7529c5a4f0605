"""Host-side time line of the owner-computes data-parallel step on a 1-rank RCCL group with every collective call made (development tool; GPU box)."""
import os, sys, socket
sys.path.insert(0, "/root/repo")
os.environ["RF_OWNER_FORCE_COLLECTIVES"] = "1"
import numpy as np, torch
import bench, thr3ed_atom_amd as rf
from thr3ed_atom_amd import distributed as rfdist
from thr3ed_atom_amd.trainers import PosedImagesInMemory, TrainStepper
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
torch.distributed.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
rfdist.FORCE_COLLECTIVES = True
dev = torch.device("cuda:0")
bounds = rf.CameraBounds(bench.NEAR, bench.FAR)
intr = rf.CameraIntrinsics(800, 800, 1111.111)
gt = bench.make_grid(dev, 128, 2, seed=7, sparse=True)
gm = rf.VolumetricModel(gt, rf.render_sh_voxel_grid, rf.SHVoxGridRenderConfig(256, bounds, perturb_sampled_points=False, white_bkgd=True), device=dev)
poses = [rf.pose_spherical(45.0 * k, -30.0, bench.RADIUS) for k in range(8)]
images = torch.stack([gm.render(p, intr).colour.permute(2, 0, 1) for p in poses])
pose_mat = torch.stack([torch.cat([p.rotation, p.translation], dim=1) for p in poses]).to(dev)
data = PosedImagesInMemory(images, pose_mat, intr, bounds)
grid = bench.make_grid(dev, 128, 2, seed=42, storage="split")
model = rf.VolumetricModel(grid, rf.render_sh_voxel_grid, rf.SHVoxGridRenderConfig(256, bounds, perturb_sampled_points=True, white_bkgd=True), device=dev)
st = TrainStepper(model, 16384, 0.03, exchange="owner")
batches = data.image_batches(8)
for _ in range(10): st.step(data, next(batches))
torch.cuda.synchronize()
st.host_timing = []
import time
t0 = time.perf_counter()
for _ in range(40): st.step(data, next(batches))
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / 40
ht = np.array(st.host_timing) * 1e3
print("ms/step", el * 1e3)
print("host sections ms: issue fwd+emit, wait bounds, exchanges, brick launch, all-gathers:", ht.mean(0).round(4), "sum", ht.mean(0).sum().round(4))
