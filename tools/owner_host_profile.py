"""Host cost of the owner-computes data-parallel step: a TINY workload (the GPU side is negligible) on a 1-rank RCCL group with every
collective call made; cProfile of 300 steps (development tool; GPU box)."""
import os, sys, socket, time, cProfile, pstats
sys.path.insert(0, "/root/repo")
os.environ["RF_OWNER_FORCE_COLLECTIVES"] = "1"
import torch, bench, thr3ed_atom_amd as rf
from thr3ed_atom_amd import distributed as rfdist
from thr3ed_atom_amd.trainers import PosedImagesInMemory, TrainStepper
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
torch.distributed.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
rfdist.FORCE_COLLECTIVES = True
dev = torch.device("cuda:0")
bounds = rf.CameraBounds(bench.NEAR, bench.FAR)
intr = rf.CameraIntrinsics(100, 100, 138.0)
images = torch.rand(8, 3, 100, 100, device=dev)
poses = [rf.pose_spherical(45.0 * k, -30.0, bench.RADIUS) for k in range(8)]
pose_mat = torch.stack([torch.cat([p.rotation, p.translation], dim=1) for p in poses]).to(dev)
data = PosedImagesInMemory(images, pose_mat, intr, bounds)
grid = bench.make_grid(dev, 16, 2, seed=42, storage="split")
model = rf.VolumetricModel(grid, rf.render_sh_voxel_grid, rf.SHVoxGridRenderConfig(32, bounds, perturb_sampled_points=True, white_bkgd=True), device=dev)
st = TrainStepper(model, 64, 0.03, exchange="owner")
batches = data.image_batches(8)
for _ in range(20): st.step(data, next(batches))
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    for _ in range(200): st.step(data, next(batches))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("host ms/step (tiny workload, owner step, 1-rank RCCL group, collectives forced)", (t1 - t0) / 200 * 1e3, "halves", st._owner["H"])
st.host_timing = []
for _ in range(100): st.step(data, next(batches))
import numpy as np
print("host sections ms [issue fwd+emit, wait bounds, exchanges issued, bricks+all-gathers issued, end]:", (np.array(st.host_timing) * 1e3).mean(0).round(4))
st.host_timing = None
pr = cProfile.Profile()
pr.enable()
for _ in range(300): st.step(data, next(batches))
pr.disable()
torch.cuda.synchronize()
ps = pstats.Stats(pr); ps.sort_stats("tottime").print_stats(28)
