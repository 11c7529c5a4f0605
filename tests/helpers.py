"""Deterministic synthetic inputs shared by the golden-vector generator (oracle/gen_golden.py) and
the tests.  Everything here is integer-hash based so that it is reproducible bit for bit on any
machine and any torch version (no dependence on an RNG implementation)."""
import os
import sys

import numpy as np
import torch

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO_ROOT not in sys.path:
    sys.path.insert(0, REPO_ROOT)

GOLDEN_DIR = os.path.join(REPO_ROOT, "tests", "golden")


def hash_uniform(shape, seed: int, lo: float = -1.0, hi: float = 1.0) -> np.ndarray:
    """float32 array of the given shape, values on a 2^-23 lattice in [lo, hi) from a
    splitmix64-style hash of (seed, linear index)."""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        x = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    u = (x >> np.uint64(40)).astype(np.float64) / float(1 << 24)  # [0, 1) on a 2^-24 lattice
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def procedural_grid(dims, num_features: int, seed: int):
    """densities [X,Y,Z,1], features [X,Y,Z,F] ~ U(-1,1) -- the reference's initialisation
    distribution (train_sh_based_voxel_grid_with_posed_images.py:202-206)."""
    X, Y, Z = dims
    dens = hash_uniform((X, Y, Z, 1), seed)
    feat = hash_uniform((X, Y, Z, num_features), seed + 1)
    return torch.from_numpy(dens), torch.from_numpy(feat)


def sparse_scene_grid(dims, num_features: int, seed: int):
    """cfg5-style sparse scene: raw density = 0.5 - |p/1.5| + 0.05 U(-1,1) on the voxel centres of a
    [-1.5, 1.5]^3 world (positive inside radius ~0.75), features U(-1,1)."""
    X, Y, Z = dims
    ax = [((np.arange(n, dtype=np.float32) + 0.5) / n * 3.0 - 1.5) / 1.5 for n in (X, Y, Z)]
    r = np.sqrt(ax[0][:, None, None] ** 2 + ax[1][None, :, None] ** 2 + ax[2][None, None, :] ** 2)
    dens = (0.5 - r + 0.05 * hash_uniform((X, Y, Z), seed)).astype(np.float32)[..., None]
    feat = hash_uniform((X, Y, Z, num_features), seed + 1)
    return torch.from_numpy(dens), torch.from_numpy(feat)


def hotdog_like_camera():
    """Synthetic camera constants taken from the reference: radius 4.0311
    (data/tests/test_datasets.py:50), near/far 2.0/6.0 (tools/convert_from_nerf_blender_dataset.py:15)
    x 0.9 / 1.1 in float32 (data/datasets.py:243-244)."""
    near = float(np.float32(2.0) * 0.9)
    far = float(np.float32(6.0) * 1.1)
    return {"radius": 4.0311, "near": near, "far": far}


def load_golden(name: str):
    path = os.path.join(GOLDEN_DIR, name)
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def g9b_batch(g, step: int):
    """The batch the REAL reference trainer drew at global iteration ``step`` (0-based) of golden G9b
    (oracle/gen_golden_trainer.py --stages): (origins, directions, pixels, t_rand of the specular render, t_rand of the diffuse render).
    The fixture stores, per stage, every ray and pixel of every training image (cast / loaded by the reference) and, per step, the
    image batch + the trainer's torch.randperm prefix; the stratified jitter of the run was procedural (hash_uniform)."""
    G, deg, hw, n_img, n_rays, iters, S, stages, eval_S, seed0 = (int(v) for v in g["config"])
    stage = 1 + step // iters
    pix, dirs = g[f"pixels_stage{stage}"], g[f"directions_stage{stage}"]
    per_image = pix.shape[1]
    sel = g["selection"][step].astype(np.int64)
    img = g["image_ids"][step].astype(np.int64)[sel // per_image]
    within = sel % per_image
    t_spec = hash_uniform((n_rays, S), seed0 + 2 * step, 0.0, 1.0)
    t_diff = hash_uniform((n_rays, S), seed0 + 2 * step + 1, 0.0, 1.0)
    return g["camera_origins"][img], dirs[img, within], pix[img, within], t_spec, t_diff


def g9b_learning_rate(g, step: int) -> float:
    """lr of global iteration ``step`` (0-based): stage lr = lr0 * stage_gamma^(stage-1), ExponentialLR(gamma) stepped every
    ``decay_steps`` iterations of the stage (modules/trainers.py:227-250, 389-390)."""
    iters = int(g["config"][5])
    lr0, gamma, decay_steps, stage_gamma = (float(v) for v in g["schedule"])
    stage, it = step // iters, step % iters
    return lr0 * stage_gamma**stage * gamma ** (it // int(decay_steps))
