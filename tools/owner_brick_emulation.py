"""What a data-parallel rank's brick pass costs at N ranks, measured on ONE GPU: the merged brick pass + Adam over 1/(N H) of the bricks
(one piece of the interleaved ownership) with N copies of this GPU's own record lists standing in for the N ranks' lists -- the same
number of records per owned brick, the same number of lists per kind, the same flush.  (Development tool, GPU box)"""
import sys, time
sys.path.insert(0, "/root/repo")
import torch, bench, thr3ed_atom_amd as rf
from thr3ed_atom_amd import ops
from thr3ed_atom_amd.trainers import PosedImagesInMemory, TrainStepper
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
st = TrainStepper(model, 16384, 0.03, brick_size=8)  # (owners sum 8^3 bricks)
batches = data.image_batches(8)
for _ in range(12): st.step(data, next(batches))
torch.cuda.synchronize()
t = st._exec["tensors"]
opt = st.optimizer
nd = st.flat.flat_gradient_parts()[0].numel()
m, v = (opt.exp_avg[:nd], opt.exp_avg[nd:]), (opt.exp_avg_sq[:nd], opt.exp_avg_sq[nd:])
nbyz = 16 * 16
SCRATCH = {}
def run(N, H, piece, reps=20, parts=1):
    q = 16 // (N * H)
    lists = [(t["pass0"]["sorted"], t["offsets2"][0], False)] * N + [(t["pass1"]["sorted"], t["offsets2"][1], True)] * N
    rng = (piece * q * nbyz, q * nbyz)
    split = None
    if parts > 1:
        key = (rng[1], parts)
        if key not in SCRATCH: SCRATCH[key] = ops.brick_split_scratch(grid, rng[1], parts)
        split = (parts, SCRATCH[key])
    for _ in range(3): ops.brick_accumulate_adam_raw(grid, 8, lists, m, v, 0.03, 0.9, 0.999, 1e-8, 20, brick_range=rng, split=split)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): ops.brick_accumulate_adam_raw(grid, 8, lists, m, v, 0.03, 0.9, 0.999, 1e-8, 20, brick_range=rng, split=split)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
print("records per list", int(t["offsets2"][0][-1]), int(t["offsets2"][1][-1]))
for N, H in ((1, 1), (1, 2), (2, 2), (4, 2), (8, 2), (8, 1)):
    times = [run(N, H, p) for p in ((0, N * H // 2, N * H - 1) if N * H > 2 else range(N * H))]
    print(f"N = {N}, H = {H}: brick pass + Adam of one piece ({16 // (N * H)} x-slab(s) of bricks, {N} list pair(s)): " + ", ".join(f"{x:.4f}" for x in times) + f" ms (pieces first / middle / last); per rank and step ~ {H * sum(times) / len(times):.4f} ms")

# ---- several workgroups per brick (rf_brick_accumulate_adam_split) ----
for N, H, parts in ((8, 2, 2), (8, 2, 4), (8, 2, 8), (4, 2, 2), (4, 2, 4), (2, 2, 2), (8, 1, 4)):
    times = [run(N, H, p, parts=parts) for p in (0, N * H // 2, N * H - 1)]
    print(f"N = {N}, H = {H}, {parts} workgroups per brick: " + ", ".join(f"{x:.4f}" for x in times) + f" ms (pieces first / middle / last); per rank and step ~ {H * sum(times) / len(times):.4f} ms")
# parity: one launch of the split pass == one launch of the plain pass on identical state (N = 8, middle piece)
def one(parts):
    p0 = st.flat.flat_param.clone(); m0 = opt.exp_avg.clone(); v0 = opt.exp_avg_sq.clone()
    N, H, piece = 8, 2, 8
    lists = [(t["pass0"]["sorted"], t["offsets2"][0], False)] * N + [(t["pass1"]["sorted"], t["offsets2"][1], True)] * N
    rng = (piece * nbyz, nbyz)
    split = (parts, ops.brick_split_scratch(grid, rng[1], parts)) if parts > 1 else None
    ops.brick_accumulate_adam_raw(grid, 8, lists, m, v, 0.03, 0.9, 0.999, 1e-8, 20, brick_range=rng, split=split)
    torch.cuda.synchronize()
    out = (st.flat.flat_param.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone())
    st.flat.flat_param.copy_(p0); opt.exp_avg.copy_(m0); opt.exp_avg_sq.copy_(v0)
    if split is not None: assert int(split[1].view(torch.int32)[: (nbyz * (1 + parts))].abs().sum()) == 0, "the split pass left its counters dirty"
    return out
ref = one(1)
for parts in (2, 4, 8):
    got = one(parts)
    print(f"parity, {parts} workgroups per brick vs 1: max |d param| {float((got[0]-ref[0]).abs().max()):.3e}, max |d exp_avg| {float((got[1]-ref[1]).abs().max()):.3e} (scale {float(ref[1].abs().max()):.3e}), "
          f"max |d exp_avg_sq| {float((got[2]-ref[2]).abs().max()):.3e} (scale {float(ref[2].abs().max()):.3e}); changed params {int((ref[0] != p0).sum()) if False else ''}")
