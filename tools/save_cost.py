"""What does writing the per-sample cache cost the forward passes?  (GPU box, repo root)"""
import sys, time
import torch
sys.path.insert(0, ".")
import bench
import thr3ed_atom_amd as rf
from thr3ed_atom_amd import ops

dev = torch.device("cuda:0")
grid = bench.make_grid(dev, 128, 2, seed=42, storage="split")
intr = rf.CameraIntrinsics(800, 800, 1111.111)
gen = torch.Generator(device=dev); gen.manual_seed(3)
rays_all = []
for k in range(8):
    r = rf.flatten_rays(rf.cast_rays(intr, rf.pose_spherical(45.0 * k, -30.0, bench.RADIUS), dev))
    idx = torch.randint(0, len(r), (2048,), device=dev, generator=gen)
    rays_all.append(r[idx])
o = torch.cat([r.origins for r in rays_all]).contiguous(); d = torch.cat([r.directions for r in rays_all]).contiguous()
nb = ops.brick_counts(grid, 8); hist = torch.zeros(nb[0] * nb[1] * nb[2] * 8, dtype=torch.int32, device=dev)
for diffuse in (False, True):
    flags = ops.render_flags(True, diffuse, False, False)
    for save, h in ((False, None), (True, None), (True, hist)):
        for _ in range(5):
            ops.render_forward_raw(grid, o, d, ops.KeyedJitter(7, 0), 256, bench.NEAR, bench.FAR, flags, save=save, key_hist=h)
        torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            ops.render_forward_raw(grid, o, d, ops.KeyedJitter(7, 0), 256, bench.NEAR, bench.FAR, flags, save=save, key_hist=h)
        b.record(); torch.cuda.synchronize()
        print("diffuse" if diffuse else "specular", "save" if save else "no save", "hist" if h is not None else "", "%.4f ms" % (a.elapsed_time(b) / 20))
