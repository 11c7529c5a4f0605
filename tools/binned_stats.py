"""Record statistics of the binned backward on the bench workload: records per list, brick visits, balance."""
import os, runpy, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from thr3ed_atom_amd import trainers

seen = {}
orig = trainers.TrainStepper._fused_step_on
def wrapped(self, *a, **k):
    seen["stepper"] = self
    return orig(self, *a, **k)
trainers.TrainStepper._fused_step_on = wrapped
sys.argv = ["bench.py", "--steps", os.environ.get("STEPS", "10"), "--warmup", "5", "--cpu-rays", "0", "--render-frames", "0", "--highres-frames", "0", "--backward", "binned"]
runpy.run_path(os.path.join(sys.path[0], "bench.py"), run_name="__main__")
st = seen["stepper"]
b = st._bins
nb = b["num_bricks"]
grid = st.vol_mod.thre3d_repr
from thr3ed_atom_amd.ops import brick_counts
nbx, nby, nbz = brick_counts(grid, st.brick_size)
tot_vis = torch.zeros(nb, dtype=torch.int64, device="cuda")
for i, off in enumerate([b["offsets"]]):
    cnt = (off[1:] - off[:-1]).view(nbx, nby, nbz, 8)
    print(f"list {i}: slots {off[-1].item()}, records {int(off[-1] - off[0])}, per-flag-class", cnt.sum((0, 1, 2)).tolist())
    vis = torch.zeros(nbx, nby, nbz, dtype=torch.int64, device="cuda")
    for o in range(8):
        ox, oy, oz = o & 1, (o >> 1) & 1, o >> 2
        sel = [f for f in range(8) if (f & o) == o]
        c = cnt[..., sel].sum(-1)
        vis[ox:, oy:, oz:] += c[: nbx - ox, : nby - oy, : nbz - oz]
    v = vis.flatten()
    tot_vis += v
    print(f"   visits {int(v.sum())}, non-empty bricks {(v > 0).sum().item()}, max {v.max().item()}, mean(non-empty) {v[v > 0].float().mean().item():.0f}")
v = tot_vis
print(f"both lists: visits {int(v.sum())}, non-empty {(v > 0).sum().item()}, max {v.max().item()}, p99 {v.float().quantile(0.99).item():.0f}, p50(non-empty) {v[v > 0].float().median().item():.0f}")
