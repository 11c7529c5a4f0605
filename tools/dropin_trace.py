"""The strict drop-in iteration (reference storage, autograd op, torch.rand jitter) alone, for rocprofv3 --kernel-trace --stats (GPU box)."""
import sys, time
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
sel = sys.argv[1] if len(sys.argv) > 1 else "keyed"
jit = sys.argv[2] if len(sys.argv) > 2 else "torch"
grid = bench.make_grid(dev, 128, 2, seed=42, storage="reference")
cfg = rf.SHVoxGridRenderConfig(256, bounds, perturb_sampled_points=True, white_bkgd=True, jitter=jit)
model = rf.VolumetricModel(grid, rf.render_sh_voxel_grid, cfg, device=dev)
st = TrainStepper(model, 16384, 0.03, fused=False, ray_selection=sel, data_parallel=False)
batches = data.image_batches(8)
for _ in range(5): st.step(data, next(batches))
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 30
per = []
for _ in range(N):
    a = time.perf_counter(); st.step(data, next(batches)); per.append(time.perf_counter() - a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
import os
print("per-step host ms:", " ".join(f"{x*1e3:.2f}" for x in per), "| threads", torch.get_num_threads())
print(f"dropin[{sel},{jit}] ms/step {(t2 - t0) / N * 1e3:.4f} host issue {(t1 - t0) / N * 1e3:.4f}")
from thr3ed_atom_amd import voxels
sh = voxels._RF_SHADOW_CACHE.get(grid)
if sh is not None:
    a, b = sh["base"].data_ptr(), sh["rest"].data_ptr()
    d, f = grid._densities.data_ptr(), grid._features.data_ptr()
    print("shadow base/rest distance GB", abs(a - b) / 2**30, "span incl sizes", (max(a + sh['base'].numel()*4, b + sh['rest'].numel()*4) - min(a, b)) / 2**30, "| parameters distance", abs(d - f) / 2**30)
    m = st.optimizer._split_moments
    if m is not None: print("moments distances GB", [abs(x[0].data_ptr() - x[1].data_ptr()) / 2**30 for x in m])
