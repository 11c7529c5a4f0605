"""Step time of the fused trainer on the small grids of the reference's progressive schedule (16^3 .. 128^3), binned against atomic adjoint
(development tool, GPU box)."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch, bench, thr3ed_atom_amd as rf
from thr3ed_atom_amd.trainers import PosedImagesInMemory, TrainStepper
dev = torch.device("cuda:0")
bounds = rf.CameraBounds(bench.NEAR, bench.FAR)
intr = rf.CameraIntrinsics(400, 400, 555.555)
gt = bench.make_grid(dev, 128, 2, seed=7, sparse=True)
gm = rf.VolumetricModel(gt, rf.render_sh_voxel_grid, rf.SHVoxGridRenderConfig(256, bounds, perturb_sampled_points=False, white_bkgd=True), device=dev)
poses = [rf.pose_spherical(45.0 * k, -30.0, bench.RADIUS) for k in range(8)]
images = torch.stack([gm.render(p, intr).colour.permute(2, 0, 1) for p in poses])
pose_mat = torch.stack([torch.cat([p.rotation, p.translation], dim=1) for p in poses]).to(dev)
data = PosedImagesInMemory(images, pose_mat, intr, bounds)
for G in (16, 32, 64, 128):
    for backward in ("binned", "atomic"):
        grid = bench.make_grid(dev, G, 2, seed=42, storage="split")
        model = rf.VolumetricModel(grid, rf.render_sh_voxel_grid, rf.SHVoxGridRenderConfig(256, bounds, perturb_sampled_points=True, white_bkgd=True), device=dev)
        st = TrainStepper(model, 16384, 0.03, backward=backward, data_parallel=False)
        batches = data.image_batches(8)
        for _ in range(30): st.step(data, next(batches))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): st.step(data, next(batches))
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
        print(f"G = {G:3d} backward = {backward:7s} fuse_optimizer = {st.fuse_optimizer} brick = {st.brick_size}: {dt * 1e3:.3f} ms / step", flush=True)
        st.flat.detach(); del st, model, grid; torch.cuda.empty_cache()
