"""The brick pass with EMPTY record lists = its optimizer flush alone (development tool; GPU box): what the flush's access pattern
reaches when nothing else runs, against the streaming rf_adam_step over the same tensors."""
import sys
import time

sys.path.insert(0, "/root/repo")
import torch

import bench
import thr3ed_atom_amd as rf
from thr3ed_atom_amd import ops
from thr3ed_atom_amd.optim import FlatGrid, FusedAdam

dev = torch.device("cuda:0")
for storage in ("split", "bricked"):
    grid = bench.make_grid(dev, 128, 2, seed=42, storage=storage)
    first, second = grid.kernel_tensors()
    m = (torch.zeros_like(first), torch.zeros_like(second))
    v = (torch.zeros_like(first), torch.zeros_like(second))
    nb = ops.brick_counts(grid, 8)
    nkeys = nb[0] * nb[1] * nb[2] * 8
    off = torch.zeros(nkeys + 1, dtype=torch.int64, device=dev)
    rec_w = torch.zeros((16, ops.expanded_record_floats(grid)), dtype=torch.float32, device=dev)
    rec_n = torch.zeros((16, ops.expanded_record_floats(grid, True)), dtype=torch.float32, device=dev)
    lists = [(rec_w, off, False), (rec_n, off, True)]

    def run(step):
        ops.brick_accumulate_adam_raw(grid, 8, lists, m, v, 0.03, 0.9, 0.999, 1e-8, step)

    for i in range(5):
        run(i + 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(50):
        run(i + 6)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    nbytes = (first.numel() + second.numel()) * 24
    print(f"{storage}: flush-only brick pass {ms:.4f} ms = {nbytes / ms / 1e9:.2f} TB/s ({nbytes / 1e9:.3f} GB)")
# streaming Adam over one flat tensor of the same size (reads p, g, m, v; writes p, m, v: 28 B per parameter)
n = 128**3 * 28
p, g_, m1, v1 = (torch.zeros(n, device=dev) for _ in range(4))
for i in range(5):
    ops.adam_step_hip(p, g_, m1, v1, 0.03, 0.9, 0.999, 1e-8, i + 1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(50):
    ops.adam_step_hip(p, g_, m1, v1, 0.03, 0.9, 0.999, 1e-8, i + 6)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
print(f"streaming rf_adam_step {ms:.4f} ms = {n * 28 / ms / 1e9:.2f} TB/s")
