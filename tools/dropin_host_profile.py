import sys, cProfile, pstats
sys.argv = [sys.argv[0]] + sys.argv[1:]
sys.path.insert(0, "/root/repo")
import torch, bench, thr3ed_atom_amd as rf
from thr3ed_atom_amd.trainers import PosedImagesInMemory, TrainStepper
dev = torch.device("cuda:0")
bounds = rf.CameraBounds(bench.NEAR, bench.FAR)
intr = rf.CameraIntrinsics(800, 800, 1111.111)
images = torch.rand(8, 3, 800, 800, device=dev)
poses = [rf.pose_spherical(45.0 * k, -30.0, bench.RADIUS) for k in range(8)]
pose_mat = torch.stack([torch.cat([p.rotation, p.translation], dim=1) for p in poses]).to(dev)
data = PosedImagesInMemory(images, pose_mat, intr, bounds)
sel, jit = sys.argv[1], sys.argv[2]
grid = bench.make_grid(dev, 128, 2, seed=42, storage="reference")
cfg = rf.SHVoxGridRenderConfig(256, bounds, perturb_sampled_points=True, white_bkgd=True, jitter=jit)
model = rf.VolumetricModel(grid, rf.render_sh_voxel_grid, cfg, device=dev)
st = TrainStepper(model, 16384, 0.03, fused=False, ray_selection=sel, data_parallel=False)
batches = data.image_batches(8)
for _ in range(5): st.step(data, next(batches))
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): st.step(data, next(batches))
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
