"""An owner's brick pass at N ranks with cubic and with 4 x 8 x 8 bricks (development tool, GPU box; see tools/owner_brick_emulation.py): the
owner pass is bound by its tile loops (N lists per kind and brick), and what helps there is dealing the LISTS of a brick out to several
workgroups (rf_brick_accumulate_adam_split: 0.198 ms for the heaviest piece at N = 8), not smaller bricks (0.293 ms) -- the data-parallel
step keeps 8^3 bricks."""
import sys
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
for bs, xs in ((8, 16), (ops.BRICK_4X8X8, 32)):
    grid = bench.make_grid(dev, 128, 2, seed=42, storage="split")
    model = rf.VolumetricModel(grid, rf.render_sh_voxel_grid, rf.SHVoxGridRenderConfig(256, bounds, perturb_sampled_points=True, white_bkgd=True), device=dev)
    st = TrainStepper(model, 16384, 0.03, brick_size=bs)
    torch.manual_seed(3)
    batches = data.image_batches(8)
    for _ in range(12): st.step(data, next(batches))
    torch.cuda.synchronize()
    t = st._exec["tensors"]; opt = st.optimizer
    nd = st.flat.flat_gradient_parts()[0].numel()
    m, v = (opt.exp_avg[:nd], opt.exp_avg[nd:]), (opt.exp_avg_sq[:nd], opt.exp_avg_sq[nd:])
    nbyz = 256
    def run(N, H, piece, reps=20, parts=1):
        q = xs // (N * H)
        lists = [(t["pass0"]["sorted"], t["offsets2"][0], False)] * N + [(t["pass1"]["sorted"], t["offsets2"][1], True)] * N
        rng = (piece * q * nbyz, q * nbyz)
        split = None
        if parts > 1: split = (parts, ops.brick_split_scratch(grid, rng[1], parts))
        for _ in range(3): ops.brick_accumulate_adam_raw(grid, bs, lists, m, v, 0.03, 0.9, 0.999, 1e-8, 20, brick_range=rng, split=split)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): ops.brick_accumulate_adam_raw(grid, bs, lists, m, v, 0.03, 0.9, 0.999, 1e-8, 20, brick_range=rng, split=split)
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps
    for N, H, parts in ((1, 1, 1), (8, 2, 1), (8, 1, 1), (4, 2, 1), (2, 2, 1)) + (((8, 2, 2), (4, 2, 2)) if bs == 8 else ()):
        times = [run(N, H, p, parts=parts) for p in ((0, N * H // 2, N * H - 1) if N * H > 2 else range(N * H))]
        print(f"brick {bs}: N = {N}, H = {H}, parts {parts}: " + ", ".join(f"{x:.4f}" for x in times) + f" ms; per rank and step ~ {H * sum(times) / len(times):.4f} ms", flush=True)
    st.flat.detach(); del st, model, grid; torch.cuda.empty_cache()
