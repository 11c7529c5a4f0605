"""debug aid (test infrastructure: it calls the oracle; run on a GPU box as `python tests/debug_train_case2.py <run seed> <case>`): replay a tests/parity_fuzz.py train case step by step (parameters after each iteration vs the oracle + torch Adam)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (repo root)
import thr3ed_atom_amd as rf
from oracle import relu_field_oracle as orc
from tests import parity_fuzz as pf
from tests.helpers import hash_uniform, procedural_grid
from thr3ed_atom_amd.trainers import TrainStepper

run_seed, i = int(sys.argv[1]), int(sys.argv[2])
over = dict(a.split("=") for a in sys.argv[3:])
seed = pf.case_seed(run_seed, i)
rng = np.random.default_rng(seed)
dims = tuple(int(rng.integers(4, 21)) for _ in range(3)); deg = int(rng.integers(0, 4)); mode = str(rng.choice(["relu", "relu", "softplus", "abs"]))
storage = str(rng.choice(["reference", "split", "bricked"])); voxel = tuple(3.0 / d for d in dims)
rho = 1.0 if mode == "abs" else float(rng.choice([5.0, 100.0 / 3.0])); F = 3 * (deg + 1) ** 2
dens, feat = procedural_grid(dims, F, seed % 100000)
n = int(rng.choice([1, 3, 37, 64, 130, 257])); S = int(rng.choice([17, 33, 40, 64, 70]))
fused = bool(rng.integers(2)); backward = str(rng.choice(["atomic", "binned"]))
fuse_opt = bool(rng.integers(2)) if (fused and backward == "binned") else None
brick = None if not (fused and backward == "binned") else [None, 4, 8][int(rng.integers(3))]
white, diffuse_reg = bool(rng.integers(2)), bool(rng.integers(2))
storage = over.get("storage", storage); backward = over.get("backward", backward)
if "fused" in over: fused = over["fused"] == "1"
print(dims, deg, mode, storage, n, S, "fused", fused, backward, fuse_opt, brick, white, diffuse_reg, rho)
dev = torch.device("cuda:0"); lr = 0.03
o = torch.from_numpy(hash_uniform((n, 3), seed + 1)); o = o / o.norm(dim=-1, keepdim=True).clamp_min(1e-3) * 4.0
d = torch.from_numpy(hash_uniform((n, 3), seed + 3)) * 1.2 - o; d = d / d.norm(dim=-1, keepdim=True)
pixels = torch.from_numpy(hash_uniform((n, 3), seed + 6, 0.0, 1.0))
grid = pf.make_grid(dev, dens, feat, voxel, (0, 0, 0), mode, rho, storage, True)
perturb = fused
cfg = rf.SHVoxGridRenderConfig(S, rf.CameraBounds(1.8, 6.6), perturb_sampled_points=perturb, white_bkgd=white)
model = rf.VolumetricModel(grid, rf.render_sh_voxel_grid, cfg, device=dev)
kw = dict(fused=fused, backward=backward, apply_diffuse_render_regularization=diffuse_reg, data_parallel=False)
stepper = TrainStepper(model, n, learning_rate=lr, **kw)
print("stepper: backward", stepper.backward, "fuse_optimizer", stepper.fuse_optimizer, "merged", getattr(stepper, "merged_bricks", None), "brick", stepper.brick_size)
cd, cf = dens.clone().requires_grad_(True), feat.clone().requires_grad_(True)
opt = torch.optim.Adam([{"params": [cd, cf], "lr": lr}], betas=(0.9, 0.999))
aabb = orc.make_aabb(dims, voxel)
for it in range(2):
    t_rands = [torch.from_numpy(hash_uniform((n, S), seed + 10 + 2 * it + k, 0.0, 1.0)).clamp_(0.0, 1.0 - 2.0**-24) for k in range(2)]
    stats = stepper.step_on(rf.Rays(o.to(dev), d.to(dev)), pixels.to(dev), t_rand=[t.to(dev) for t in t_rands] if perturb else None)
    opt.zero_grad(); losses = []
    for k, diffuse in enumerate((False, True) if diffuse_reg else (False,)):
        out = orc.render(cd, cf, o, d, aabb, 1.8, 6.6, S, rho, mode, white_bkgd=white, render_diffuse=diffuse, t_rand=t_rands[k] if perturb else None)
        losses.append(torch.nn.functional.l1_loss(out["colour"], pixels))
    sum(losses).backward()
    g = torch.cat([cd.grad.reshape(-1), cf.grad.reshape(-1)])
    opt.step()
    torch.cuda.synchronize()
    ours = torch.cat([grid.densities.detach().reshape(-1), grid.features.detach().reshape(-1)]).cpu()
    ref = torch.cat([cd.detach().reshape(-1), cf.detach().reshape(-1)])
    err = (ours - ref).abs()
    nd = cd.numel()
    print(f"it {it}: loss hip {float(stats.specular_loss):.7f} oracle {losses[0].item():.7f}; nonzero grads {int((g != 0).sum())} of {g.numel()}; param err max {float(err.max()):.3e}"
          f" (dens {float(err[:nd].max()):.3e} feat {float(err[nd:].max()):.3e}); frac within 5e-5: {float((err <= 5e-5).float().mean()):.4f}; # err > 1e-3: {int((err > 1e-3).sum())}"
          f" of which grad==0 in oracle: {int(((err > 1e-3) & (g == 0)).sum())}")
    bad = (err > 1e-3).nonzero().reshape(-1)[:5]
    for b in bad.tolist():
        print("   idx", b, "ours", float(ours[b]), "ref", float(ref[b]), "oracle grad", float(g[b]))
