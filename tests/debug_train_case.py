"""debug aid (test infrastructure: it calls the oracle; run on a GPU box as `python tests/debug_train_case.py <run seed> <case>`): gradients of the first iteration of a tests/parity_fuzz.py train case -- HIP vs the float32 oracle vs the float64 oracle"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (repo root)
import thr3ed_atom_amd as rf
from thr3ed_atom_amd import ops
from oracle import relu_field_oracle as orc
from tests import parity_fuzz as pf
from tests.helpers import hash_uniform, procedural_grid

run_seed, i = int(sys.argv[1]), int(sys.argv[2])
seed = pf.case_seed(run_seed, i)
rng = np.random.default_rng(seed)
dims = tuple(int(rng.integers(4, 21)) for _ in range(3)); deg = int(rng.integers(0, 4)); mode = str(rng.choice(["relu", "relu", "softplus", "abs"]))
storage = str(rng.choice(["reference", "split", "bricked"])); voxel = tuple(3.0 / d for d in dims)
rho = 1.0 if mode == "abs" else float(rng.choice([5.0, 100.0 / 3.0])); F = 3 * (deg + 1) ** 2
dens, feat = procedural_grid(dims, F, seed % 100000)
n = int(rng.choice([1, 3, 37, 64, 130, 257])); S = int(rng.choice([2, 17, 40, 64, 70]))
fused = bool(rng.integers(2)); backward = str(rng.choice(["atomic", "binned"]))
print(dims, deg, mode, storage, n, S, fused, backward, rho)
dev = torch.device("cuda:0")
o = torch.from_numpy(hash_uniform((n, 3), seed + 1)); o = o / o.norm(dim=-1, keepdim=True).clamp_min(1e-3) * 4.0
d = torch.from_numpy(hash_uniform((n, 3), seed + 3)) * 1.2 - o; d = d / d.norm(dim=-1, keepdim=True)
pixels = torch.from_numpy(hash_uniform((n, 3), seed + 6, 0.0, 1.0))
white = False
t = torch.from_numpy(hash_uniform((n, S), seed + 10, 0.0, 1.0)).clamp_(0.0, 1.0 - 2.0**-24)
grid = pf.make_grid(dev, dens, feat, voxel, (0, 0, 0), mode, rho, storage, True)
cfg = rf.SHVoxGridRenderConfig(S, rf.CameraBounds(1.8, 6.6), perturb_sampled_points=True, white_bkgd=white)
res = {}
for bw in ("atomic", "binned"):
    ops.AUTOGRAD_BACKWARD = bw
    grid.zero_grad() if hasattr(grid, "zero_grad") else None
    for p_ in grid.parameters():
        p_.grad = None
    out = rf.render_sh_voxel_grid(grid, rf.Rays(o.to(dev), d.to(dev)), cfg, None, t_rand=t.to(dev))
    torch.nn.functional.l1_loss(out.colour, pixels.to(dev)).backward()
    gd, gf = grid.reference_gradients()
    res[bw] = (gd.detach().cpu().double().clone(), gf.detach().cpu().double().clone())
aabb = orc.make_aabb(dims, voxel)
ref = {}
for dt in (torch.float32, torch.float64):
    cd, cf = dens.detach().clone().to(dt).requires_grad_(True), feat.detach().clone().to(dt).requires_grad_(True)
    r = orc.render(cd, cf, o.to(dt), d.to(dt), aabb, 1.8, 6.6, S, rho, mode, white_bkgd=white, t_rand=t.to(dt))
    torch.nn.functional.l1_loss(r["colour"], pixels.to(dt)).backward()
    ref[dt] = (cd.grad.double(), cf.grad.double())
g64 = ref[torch.float64]
for name, g in (("oracle32", ref[torch.float32]), ("hip atomic", res["atomic"]), ("hip binned", res["binned"])):
    for k, lab in ((0, "dens"), (1, "feat")):
        e = (g[k] - g64[k]).abs(); m = float(g64[k].abs().max())
        big = g64[k].abs() > 1e-3 * m
        rel = (e[big] / g64[k].abs()[big])
        print(f"{name:11s} {lab}: max|g64| {m:.3e}  max abs err {float(e.max()):.3e} ({float(e.max())/m:.2e} of max)  max rel err on |g|>1e-3 max: {float(rel.max()) if rel.numel() else 0:.3e}")
