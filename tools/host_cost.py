import sys, time
sys.path.insert(0, "/root/repo")
import torch, bench, thr3ed_atom_amd as rf
from thr3ed_atom_amd.trainers import PosedImagesInMemory, TrainStepper
dev = torch.device("cuda:0")
bounds = rf.CameraBounds(bench.NEAR, bench.FAR)
intr = rf.CameraIntrinsics(100, 100, 138.0)
images = torch.rand(8, 3, 100, 100, device=dev)
poses = [rf.pose_spherical(45.0 * k, -30.0, bench.RADIUS) for k in range(8)]
pose_mat = torch.stack([torch.cat([p.rotation, p.translation], dim=1) for p in poses]).to(dev)
data = PosedImagesInMemory(images, pose_mat, intr, bounds)
grid = bench.make_grid(dev, 16, 2, seed=42, storage="split")
model = rf.VolumetricModel(grid, rf.render_sh_voxel_grid, rf.SHVoxGridRenderConfig(32, bounds, perturb_sampled_points=True, white_bkgd=True), device=dev)
st = TrainStepper(model, 64, 0.03)
batches = data.image_batches(8)
for _ in range(20): st.step(data, next(batches))
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(200): st.step(data, next(batches))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host ms/step (tiny workload)", (t1 - t0) / 200 * 1e3, "incl. final sync", (t2 - t0) / 200 * 1e3)
