"""Which part of the grid ever receives gradient on the bench workload (bricks / x-slabs with a non-zero Adam second moment)? (GPU box)"""
import sys
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
grid = bench.make_grid(dev, 128, 2, seed=42, storage="split")
model = rf.VolumetricModel(grid, rf.render_sh_voxel_grid, rf.SHVoxGridRenderConfig(256, bounds, perturb_sampled_points=True, white_bkgd=True), device=dev)
st = TrainStepper(model, 16384, 0.03)
batches = data.image_batches(8)
for n in (1, 10, 100):
    while st.optimizer.step_count < n: st.step(data, next(batches))
    torch.cuda.synchronize()
    nd = st.flat.flat_gradient_parts()[0].numel()
    v = st.optimizer.exp_avg_sq[:nd].view(128, 128, 128, 4)
    node = (v != 0).any(-1)
    brick = node.view(16, 8, 16, 8, 16, 8).permute(0, 2, 4, 1, 3, 5).reshape(16, 16, 16, -1).any(-1)
    print(f"after {n} steps: nodes touched {float(node.float().mean()):.3f}, bricks touched {float(brick.float().mean()):.3f}, per x-slab {[round(float(b.float().mean()), 2) for b in brick]}")
